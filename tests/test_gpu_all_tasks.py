"""`-m gpu`: every committed descriptor bundle (robot_lab_amd/data/*.json - 39 task ids) steps for a while under random actions and
stays finite, and a diverged environment cannot take the launch down.  Short-horizon parity (5 steps) did not notice that Agibot D1's
URDF declares velocity="0" for every joint (a CAD exporter's "not specified"), which froze its joints against the solver and blew the
state up within ~100 steps - and the non-finite root position then indexed the heightfield out of bounds (a GPU memory fault).  Both
ends are pinned here: no bundle diverges, and NaN / huge root positions written into the state on purpose are survived."""
import glob
import os

import numpy as np
import pytest

from robot_lab_amd.scene import DATA_DIR

pytestmark = pytest.mark.gpu
TASKS = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(DATA_DIR, "*.json")))


@pytest.mark.parametrize("task", TASKS)
def test_every_bundle_stays_finite(task):
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    N = 256
    env = ManagerBasedRLEnv(task, num_envs=N, seed=1, device="cuda:0")
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(200):
        obs, rew, term, tout, _ = env.step(torch.rand(N, env.num_actions, device="cuda", generator=g) * 2 - 1)
    st = env.read_state()
    assert torch.isfinite(obs["policy"]).all() and torch.isfinite(obs["critic"]).all() and torch.isfinite(rew).all()
    assert np.isfinite(st["root_state"]).all() and np.isfinite(st["joint_pos"]).all() and np.isfinite(st["joint_vel"]).all()
    assert np.abs(st["joint_vel"]).max() < 1e3 and np.abs(st["root_state"][:, 7:13]).max() < 1e3, "velocities of a physical robot"
    env.close()


@pytest.mark.parametrize("task", ["RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0", "RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0"])
def test_non_finite_state_cannot_fault_the_launch(task):
    """Root positions of NaN, +-inf and 1e30 in a few envs: the heightfield lookups (contacts, height scan) clamp their cell indices as
    integers, so the launch completes, the other envs are untouched, and a reset brings the poisoned ones back."""
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    N = 128
    env = ManagerBasedRLEnv(task, num_envs=N, seed=2, device="cuda:0")
    twin = ManagerBasedRLEnv(task, num_envs=N, seed=2, device="cuda:0")
    env.reset()
    twin.reset()
    st = env.read_state()
    root = st["root_state"].copy()
    bad = [3, 17, 64, 99]
    root[3, 0] = np.nan
    root[17, 1] = np.inf
    root[64, 0] = -np.inf
    root[99, :2] = 1e30
    env.write_state(root_state=root)
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(5):
        a = torch.rand(N, env.num_actions, device="cuda", generator=g) * 2 - 1
        obs, rew, term, tout, _ = env.step(a)
        obs2, rew2, _, _, _ = twin.step(a)
    torch.cuda.synchronize()  # the launches completed: no memory fault
    good = np.setdiff1d(np.arange(N), bad)
    assert torch.equal(obs["critic"][good], obs2["critic"][good]) and torch.equal(rew[good], rew2[good])
    env.reset(env_ids=torch.tensor(bad, device="cuda"))
    obs, rew, _, _, _ = env.step(torch.zeros(N, env.num_actions, device="cuda"))
    assert torch.isfinite(obs["critic"]).all() and torch.isfinite(rew).all()
    env.close()
    twin.close()
