"""`-m gpu`: the HIP path, called through the C-ABI (robot_lab_amd.env -> librl_env_hip.so), against the
fp64 oracle on identical seeds and actions, over FREE-RUNNING 5-step trajectories (20 physics substeps with contacts) of
21 task ids.  A free run diverges wherever one side crosses a contact switch the other does not, so the comparison is
explicit about that: the oracle records how close every env came to a discontinuity of the model on every step
(Physics.margins; helpers.switch_mask with 10x the one-step margins, because the two trajectories drift apart by
round-off amplification before they reach the switch), those envs are excluded and counted, and 100 % of the entries of
the remaining envs must agree: state rtol 2e-3 (atol 2e-4), rewards atol 2e-5, dones exact.  The tight, single-step form
of this comparison at the BASELINE sizes is tests/test_gpu_teacher_forced.py."""
import numpy as np
import pytest

from helpers import SWITCH_EPS, assert_close, oracle_root_state, switch_mask
from oracle.env import OracleEnv
from robot_lab_amd.scene import build_world, load_bundle

pytestmark = pytest.mark.gpu

TASKS = [
    "RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0",
    "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0",
    "RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0",
    "RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0",
    "RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0",
    "RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0",
    "RobotLab-Isaac-Velocity-Rough-Unitree-B2-v0",
    "RobotLab-Isaac-Velocity-Rough-Deeprobotics-Lite3-v0",
    "RobotLab-Isaac-Velocity-Rough-Deeprobotics-M20-v0",
    "RobotLab-Isaac-Velocity-Rough-Zsibot-ZSL1-v0",
    "RobotLab-Isaac-Velocity-Rough-Zsibot-ZSL1W-v0",
    "RobotLab-Isaac-Velocity-Rough-RoboParty-ATOM01-v0",
    "RobotLab-Isaac-Velocity-Rough-RobotEra-Xbot-v0",
    "RobotLab-Isaac-Velocity-Rough-MagicLab-Bot-Gen1-v0",
    "RobotLab-Isaac-Velocity-Flat-Openloong-Loong-v0",
    "RobotLab-Isaac-Velocity-Flat-DDTRobot-Tita-v0",
    "RobotLab-Isaac-Velocity-Rough-MagicLab-Bot-Z1-v0",
    "RobotLab-Isaac-Velocity-Flat-HandStand-Unitree-A1-v0",
    "RobotLab-Isaac-Velocity-Rough-Unitree-B2W-v0",
    "RobotLab-Isaac-Velocity-Flat-MagicLab-Dog-W-v0",
    "RobotLab-Isaac-Velocity-Flat-MagicLab-Dog-v0",
]


def _pair(task, N, seed):
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    env = ManagerBasedRLEnv(task, num_envs=N, seed=seed, device="cuda:0")
    desc, extra = load_bundle(task)
    h, to, eo = build_world(desc, extra, N, 0)
    return env, OracleEnv(desc, h, to, N, seed, eo), torch


FREE_RUN_EPS = {k: 10.0 * v for k, v in SWITCH_EPS.items()}


@pytest.mark.parametrize("task", TASKS)
def test_short_horizon_parity(task):
    N = 32 if any(r in task for r in ("G1", "ATOM01", "Xbot", "Gen1", "Loong", "Tita", "Z1")) else 64
    env, ora, torch = _pair(task, N, 11)
    obs, _ = env.reset()
    o = ora.reset()
    assert_close("obs0", obs["policy"].cpu().numpy(), o[0], 1e-4, 1e-5)
    assert_close("critic0", obs["critic"].cpu().numpy(), o[1], 1e-3, 1e-4)
    rng = np.random.default_rng(3)
    ora.phys.margins = {}  # minima over the whole trajectory: an env that touched a switch at step s stays excluded afterwards
    on_switch = np.zeros(N, dtype=bool)
    for s in range(5):
        a = rng.uniform(-1, 1, (N, env.num_actions)).astype(np.float32)
        obs, rew, term, tout, _ = env.step(torch.from_numpy(a).cuda())
        o = ora.step(a)
        on_switch |= switch_mask(ora.phys.margins, FREE_RUN_EPS)
        ok = ~on_switch
        assert_close(f"reward[{s}]", rew.cpu().numpy()[ok], ora.reward[ok], 1e-3, 2e-5)
        assert np.array_equal((term | tout).cpu().numpy()[ok], (ora.terminated | ora.time_outs)[ok])
    ok = ~on_switch
    assert on_switch.mean() <= 0.25, f"{on_switch.sum()} of {N} envs came within the switch margins over 5 steps"
    d = env.scene["robot"].data
    assert_close("root", d.root_state_w.cpu().numpy()[ok], oracle_root_state(ora)[ok], 2e-3, 2e-4)
    assert_close("q", d.joint_pos.cpu().numpy()[ok], ora.st["q"][ok], 2e-3, 2e-4)
    assert_close("qd", d.joint_vel.cpu().numpy()[ok], ora.st["qd"][ok], 5e-3, 5e-3)
    assert_close("rew_terms", env.reward_terms().cpu().numpy()[:, ok], ora.reward_terms[:, ok], 2e-3, 2e-5)
    assert_close("policy", obs["policy"].cpu().numpy()[ok], o[0][ok], 5e-3, 5e-3)
    assert_close("critic", obs["critic"].cpu().numpy()[ok], o[1][ok], 5e-3, 5e-3)
    env.close()


@pytest.mark.parametrize("task,N", [(TASKS[1], 64), (TASKS[5], 32)])
def test_time_out_reset_parity(task, N):
    """Force time-outs through the settable episode_length_buf (rsl_rl init_at_random_ep_len path)."""
    env, ora, torch = _pair(task, N, 5)
    env.reset()
    ora.reset()
    ep = np.zeros(N, dtype=np.int64)
    ep[::4] = env.max_episode_length - 2
    env.episode_length_buf = torch.from_numpy(ep)
    ora.episode_length_buf[:] = ep
    rng = np.random.default_rng(0)
    n_reset = 0
    ora.phys.margins = {}
    on_switch = np.zeros(N, dtype=bool)
    for s in range(3):
        a = rng.uniform(-1, 1, (N, env.num_actions)).astype(np.float32)
        obs, rew, term, tout, extras = env.step(torch.from_numpy(a).cuda())
        o = ora.step(a)
        on_switch |= switch_mask(ora.phys.margins, FREE_RUN_EPS)
        done = (term | tout).cpu().numpy()
        assert np.array_equal(done, ora.terminated | ora.time_outs)
        n_reset += int(done.sum())
        if done.any():
            assert float(extras["log"]["Episode_Termination/time_out"]) == float(ora.time_outs_terms[0].sum())
            for name in ("Episode_Reward/track_lin_vel_xy_exp", "Metrics/base_velocity/error_vel_xy"):
                np.testing.assert_allclose(float(extras["log"][name]), ora.log[name], rtol=2e-3, atol=1e-6)
        assert np.array_equal(env.episode_length_buf.cpu().numpy(), ora.episode_length_buf)
    assert n_reset >= N // 4  # the forced time-outs (+ any illegal-contact terminations on G1)
    d = env.scene["robot"].data
    ok = ~on_switch
    assert_close("root", d.root_state_w.cpu().numpy()[ok], oracle_root_state(ora)[ok], 2e-3, 2e-4)
    assert_close("critic", obs["critic"].cpu().numpy()[ok], o[1][ok], 5e-3, 5e-3)
    env.close()


@pytest.mark.parametrize("task,N", [(TASKS[1], 4096), (TASKS[2], 4096), (TASKS[5], 2048), (TASKS[3], 4096)])
def test_full_size_properties(task, N):
    """BASELINE configs 2 (A1 Rough, 4096 envs), 3 (Go2 Rough, 4096 per GPU), 4 (G1 Rough, 2048) and 5 (Go2W Rough, 4096) at
    full size: size-independent invariants of step() over 60 steps (oracle parity at these sizes: test_gpu_teacher_forced.py)."""
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    env = ManagerBasedRLEnv(task, num_envs=N, seed=42, device="cuda:0")
    env2 = ManagerBasedRLEnv(task, num_envs=N, seed=42, device="cuda:0")
    A, scan0 = env.num_actions, env.get_observations()["critic"].shape[1] - 187
    env.reset()
    env2.reset()
    env.episode_length_buf = torch.randint(0, env.max_episode_length, (N,))
    env2.episode_length_buf = env.episode_length_buf.clone()
    g = torch.Generator(device="cuda").manual_seed(0)
    total_done = 0
    for s in range(60):
        a = torch.rand(N, A, device="cuda", generator=g) * 2 - 1
        ep_before = env.episode_length_buf.clone()
        obs, rew, term, tout, _ = env.step(a)
        obs2, rew2, _, _, _ = env2.step(a)
        done = term | tout
        total_done += int(done.sum())
        # determinism: same seed, same actions -> bit-identical
        assert torch.equal(obs["critic"], obs2["critic"]) and torch.equal(rew, rew2)
        assert torch.isfinite(obs["policy"]).all() and torch.isfinite(obs["critic"]).all() and torch.isfinite(rew).all()
        # reward is the sum of its weighted terms
        torch.testing.assert_close(env.reward_terms().sum(0), rew, rtol=1e-4, atol=1e-5)
        # episode counter: +1, or 0 after a reset
        ep = env.episode_length_buf
        assert torch.equal(ep[~done], ep_before[~done] + 1) and bool((ep[done] == 0).all())
        # height scan is clipped to [-1, 1] (velocity_env_cfg.py:236-241)
        assert float(obs["critic"][:, scan0:].abs().max()) <= 1.0
    assert total_done > 0
    quat = env.scene["robot"].data.root_quat_w
    torch.testing.assert_close(quat.norm(dim=1), torch.ones(N, device="cuda"), rtol=1e-4, atol=1e-4)
    env.close()
    env2.close()
