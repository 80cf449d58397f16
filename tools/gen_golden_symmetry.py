#!/usr/bin/env python
"""tests/golden/symmetry_anymal.npz: outputs of the REFERENCE's `compute_symmetric_states`
(/root/reference/source/robot_lab/.../velocity/mdp/symmetry/anymal.py, imported unchanged) on random batches.
`tensordict` is not installed here: a dict with the two members the function touches (`batch_size`, `repeat`) stands in.

Run in the build container:  python tools/gen_golden_symmetry.py"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
REF = "/root/reference/source/robot_lab/robot_lab/tasks/manager_based/locomotion/velocity/mdp/symmetry/anymal.py"


class TensorDict(dict):
    @property
    def batch_size(self):
        return next(iter(self.values())).shape[:1]

    def repeat(self, n):
        return TensorDict({k: v.repeat(n, *([1] * (v.ndim - 1))) for k, v in self.items()})


sys.modules["tensordict"] = types.SimpleNamespace(TensorDict=TensorDict)
spec = importlib.util.spec_from_file_location("ref_anymal_symmetry", REF)
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

g = torch.Generator().manual_seed(11)
B = 37
obs = TensorDict(policy=torch.randn(B, 45, generator=g), critic=torch.randn(B, 48, generator=g))
actions = torch.randn(B, 12, generator=g)
env = types.SimpleNamespace(unwrapped=None)
obs_aug, act_aug = mod.compute_symmetric_states(env, obs, actions)
assert torch.equal(obs_aug["critic"], obs["critic"].repeat(4, 1))  # other groups are only replicated (anymal.py:52)
np.savez_compressed(os.path.join(os.environ.get("RL_GOLDEN_DIR", os.path.join(ROOT, "tests", "golden")), "symmetry_anymal.npz"), obs=obs["policy"].numpy(), actions=actions.numpy(),
                    obs_aug=obs_aug["policy"].numpy(), actions_aug=act_aug.numpy())
print("wrote symmetry_anymal.npz", tuple(obs_aug["policy"].shape), tuple(act_aug.shape))
