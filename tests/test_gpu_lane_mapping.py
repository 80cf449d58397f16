"""`-m gpu`: the lane mapping `rl_env_create` picks from the launch size (csrc/rl_env.hip envs_per_wave): three mappings of the one lane
program - 16, 8 or 4 lanes per env (4, 8 or 16 envs per wavefront) - chosen by rounds x cost; `RL_ENV_SUB` forces one; the trunk + limbs
instances have a 16- and a 32-lane mapping (4 or 2 envs per wavefront) and take the 32-lane one at every size (same envs per round,
shorter round: profiles/r04b_g1_sweep.txt).  The mappings themselves are held to the oracle at full size by tests/test_gpu_teacher_forced.py
("sub1" / "sub2" configs) and to bit-equality across workgroup shapes by tests/test_gpu_canary.py; here: the choice, and that the three
mappings agree with each other to fp32 round-off on the first steps (they differ by the order of the cross-limb sums only)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
A1 = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"


@pytest.mark.parametrize("task,N,want", [(A1, 4096, 4), (A1, 8192, 8), (A1, 16384, 16), (A1, 24576, 8), (A1, 65536, 16),
                                         ("RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0", 8192, 2), ("RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0", 256, 2)])
def test_mapping_chosen_from_launch_size(task, N, want, monkeypatch):
    from robot_lab_amd.env import ManagerBasedRLEnv

    monkeypatch.delenv("RL_ENV_SUB", raising=False)
    env = ManagerBasedRLEnv(task, num_envs=N, seed=1, device="cuda:0")
    assert env._native.envs_per_wavefront() == want
    env.close()


def test_forced_mappings_agree(monkeypatch):
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    N, outs = 256, {}
    for sub in ("4", "2", "1"):
        monkeypatch.setenv("RL_ENV_SUB", sub)
        env = ManagerBasedRLEnv(A1, num_envs=N, seed=9, device="cuda:0")
        assert env._native.envs_per_wavefront() == 16 // int(sub)
        obs, _ = env.reset()
        g = torch.Generator(device="cuda").manual_seed(4)
        for _ in range(2):
            obs, rew, term, tout, _ = env.step(torch.rand(N, env.num_actions, device="cuda", generator=g) * 2 - 1)
        outs[sub] = (obs["critic"].cpu().numpy(), rew.cpu().numpy(), env.reward_terms().cpu().numpy(), (term | tout).cpu().numpy())
        env.close()
    for sub in ("2", "1"):
        np.testing.assert_allclose(outs[sub][0], outs["4"][0], rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(outs[sub][1], outs["4"][1], rtol=1e-3, atol=2e-5)
        np.testing.assert_allclose(outs[sub][2], outs["4"][2], rtol=2e-3, atol=2e-5)
        assert np.array_equal(outs[sub][3], outs["4"][3])
