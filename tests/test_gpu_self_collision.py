"""-m gpu: the self-collision pass of the trunk + limbs instance (csrc/env_step.h self_place / self_apply) through the C-ABI,
against the oracle, from a state IN contact (tests/test_self_collision.py builds it: an upper arm pressed against the torso)."""
import numpy as np
import pytest

from helpers import assert_close
from test_self_collision import G1, _drive_arm_into_torso, capsule_gaps

pytestmark = pytest.mark.gpu


def _native_after(state, a, N, steps, monkeypatch=None, self_off=False):
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    if self_off:
        monkeypatch.setenv("RL_ENV_SELF", "0")
    env = ManagerBasedRLEnv(G1, num_envs=N, seed=7, device="cuda:0")
    env.reset()
    s = env.read_state()
    for k, v in state.items():  # env 0's state of the oracle run in every env of the batch
        if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == 1 and k in s and s[k].shape[0] == N:
            s[k] = np.repeat(v, N, axis=0)
    s["episode_sums"][:] = 0.0
    s["step_count"] = state["step_count"]
    env.load_state(s)
    act = torch.from_numpy(np.repeat(a, N, axis=0)).to("cuda")
    for _ in range(steps):
        env.step(act)
    return env.read_state()


@pytest.mark.parametrize("sub", ["4", "8"])
def test_hip_matches_oracle_in_self_contact(sub, monkeypatch):
    monkeypatch.setenv("RL_ENV_SUB", sub)  # both lane mappings of the trunk + limbs instance
    _, src, a = _drive_arm_into_torso(True, steps=25)
    assert min(g for g, _, _ in capsule_gaps(src)) < -0.005
    state = src.read_state()
    for _ in range(2):
        src.step(a)
    N = 16  # env 0 is the oracle's env (the others carry their own startup randomisation: masses, friction, COM): compare that one
    got = _native_after(state, a, N, 2)
    want = src.read_state()
    for k, rtol, atol in (("root_state", 1e-3, 1e-4), ("joint_pos", 1e-3, 1e-4), ("joint_vel", 3e-3, 3e-3)):
        assert_close(k, got[k][:1], want[k], rtol, atol)
    assert np.isfinite(got["joint_vel"]).all()
    # and the pass acted: RL_ENV_SELF=0 ends elsewhere
    off = _native_after(state, a, N, 2, monkeypatch, self_off=True)
    assert np.abs(off["joint_vel"][:1] - got["joint_vel"][:1]).max() > 0.05
