"""`-m gpu`: edge cases of the hot path through the C-ABI - ragged env counts (padding to the wavefront tile),
partial resets by env id, zero actions, the one-lane-per-leg mapping, and seeds."""
import os

import numpy as np
import pytest

from helpers import OracleWithTwin, assert_close, oracle_root_state
from oracle.env import OracleEnv
from robot_lab_amd.scene import build_world, load_bundle

pytestmark = pytest.mark.gpu
TASK = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"


def _pair(N, seed, task=TASK, oracle_only=False):
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    desc, extra = load_bundle(task)
    h, to, eo = build_world(desc, extra, N, 0)
    if oracle_only:
        return OracleEnv(desc, h, to, N, seed, eo)
    env = ManagerBasedRLEnv(task, num_envs=N, seed=seed, device="cuda:0")
    return env, OracleEnv(desc, h, to, N, seed, eo), torch


@pytest.mark.parametrize("N", [1, 7, 37])
def test_ragged_env_counts(N):
    """num_envs that is not a multiple of the 16-env allocation tile: padded lanes must not leak into results."""
    env, ora, torch = _pair(N, 3)
    obs, _ = env.reset()
    o = ora.reset()
    assert obs["policy"].shape == (N, 45) and obs["critic"].shape == (N, 235)
    rng = np.random.default_rng(N)
    for _ in range(3):
        a = rng.uniform(-1, 1, (N, 12)).astype(np.float32)
        obs, rew, term, tout, _ = env.step(torch.from_numpy(a).cuda())
        o = ora.step(a)
    assert_close("reward", rew.cpu().numpy(), ora.reward, 1e-3, 2e-5)
    assert_close("critic", obs["critic"].cpu().numpy(), o[1], 5e-3, 5e-3)
    assert_close("root", env.scene["robot"].data.root_state_w.cpu().numpy(), oracle_root_state(ora), 2e-3, 2e-4)
    env.close()


@pytest.mark.parametrize("task", [TASK, "RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0"])
def test_partial_reset_by_env_ids(task):
    N = 32
    env, ora, torch = _pair(N, 9, task)
    env.reset()
    ora.reset()
    rng = np.random.default_rng(0)
    for _ in range(2):
        a = rng.uniform(-1, 1, (N, env.num_actions)).astype(np.float32)
        env.step(torch.from_numpy(a).cuda())
        ora.step(a)
    ids = [3, 4, 17, 31]
    before = env.scene["robot"].data.root_state_w.clone()
    obs, _ = env.reset(env_ids=ids)
    o = ora.reset(env_ids=ids)
    after = env.scene["robot"].data.root_state_w
    keep = np.setdiff1d(np.arange(N), ids)
    assert torch.equal(before[keep], after[keep])                       # untouched envs keep their state bit-for-bit
    tol = (1e-4, 1e-5) if task == TASK else (2e-3, 2e-4)  # the untouched envs carry two steps of fp32 drift (29 DoF on G1)
    assert_close("root", after.cpu().numpy(), oracle_root_state(ora), *tol)
    assert_close("critic", obs["critic"].cpu().numpy(), o[1], 2e-3, 2e-3)
    assert np.array_equal(env.episode_length_buf.cpu().numpy(), ora.episode_length_buf)
    with pytest.raises(Exception):
        env.reset(env_ids=[N])                                          # out of range -> error, not silent corruption
    env.close()


def test_zero_actions_stand_regime():
    """zero_agent.py:68 regime: joints are held at the default pose by the PD loop."""
    N = 64
    env, ora, torch = _pair(N, 2, "RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0")
    env.reset()
    ora.reset()
    a = np.zeros((N, 12), dtype=np.float32)
    two = OracleWithTwin(lambda: ora)  # free run: per-entry tolerance from the disturbed twin (helpers.OracleWithTwin)
    two.twin = _pair(N, 2, "RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0", oracle_only=True)
    two.twin.phys.solve_dtype = np.float32
    two.twin.reset()
    two._perturb()
    for _ in range(8):
        obs, rew, term, tout, _ = env.step(torch.from_numpy(a).cuda())
        two.step(a)
    two.close("reward", rew.cpu().numpy(), lambda e: e.reward, 1e-3, 3e-5)
    two.close("q", env.scene["robot"].data.joint_pos.cpu().numpy(), lambda e: e.st["q"], 3e-3, 3e-4)
    env.close()


def test_one_lane_per_leg_mapping_matches_too(monkeypatch):
    """RL_ENV_SUB=1 selects the 4-lanes-per-env kernels; both mappings implement the same step."""
    monkeypatch.setenv("RL_ENV_SUB", "1")
    N = 48
    env, ora, torch = _pair(N, 4)
    env.reset()
    ora.reset()
    rng = np.random.default_rng(1)
    two = OracleWithTwin(lambda: ora)
    two.twin = _pair(N, 4, oracle_only=True)
    two.twin.phys.solve_dtype = np.float32
    two.twin.reset()
    two._perturb()
    for _ in range(4):
        a = rng.uniform(-1, 1, (N, 12)).astype(np.float32)
        obs, rew, term, tout, _ = env.step(torch.from_numpy(a).cuda())
        two.step(a)
    two.close("reward", rew.cpu().numpy(), lambda e: e.reward, 1e-3, 2e-5)
    two.close("root", env.scene["robot"].data.root_state_w.cpu().numpy(), oracle_root_state, 2e-3, 2e-4)
    env.close()


def test_seeds_give_different_but_reproducible_episodes():
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    obs = []
    for seed in (1, 1, 2):
        env = ManagerBasedRLEnv(TASK, num_envs=64, seed=seed, device="cuda:0")
        o, _ = env.reset()
        for _ in range(3):
            o, *_ = env.step(torch.zeros(64, 12, device="cuda"))
        obs.append(o["critic"].clone())
        env.close()
    assert torch.equal(obs[0], obs[1]) and not torch.equal(obs[0], obs[2])


def test_episode_log_stays_readable_past_the_ring():
    """extras["log"] is a view of the step's slot in the device-side log ring (include/rl_env.h RL_LOG_RING): it can
    be read many steps later (rsl_rl keeps the dicts of an iteration and reads them at its end) and holds that step's
    numbers only; a dict that is still held when its slot is about to be reused is materialised by the env (device-side
    copies), so a reader that logs every 100 steps gets the right numbers, too."""
    from robot_lab_amd.desc import RL_LOG_RING

    N = 64
    env, ora, torch = _pair(N, 11)
    env.reset()
    ora.reset()
    ep = np.zeros(N, dtype=np.int64)
    ep[::4] = ora.max_episode_length - 2  # 16 envs time out at the second step
    ora.episode_length_buf[:] = ep
    env.episode_length_buf = torch.as_tensor(ep)
    rng = np.random.default_rng(5)
    kept, want = [], []
    for s in range(RL_LOG_RING + 8):
        a = rng.uniform(-1, 1, (N, env.num_actions)).astype(np.float32)
        _, _, term, tout, extras = env.step(torch.from_numpy(a).cuda())
        if s < 3:
            ora.step(a)
            kept.append(extras["log"])
            want.append((int((ora.terminated | ora.time_outs).sum()), dict(ora.log) if (ora.terminated | ora.time_outs).any() else None))
        if s == 30:  # 28-30 steps later: still this step's numbers
            for lg, (ndone, olog) in zip(kept[:2], want[:2]):
                if olog is not None:
                    for name in ("Episode_Reward/track_lin_vel_xy_exp", "Episode_Reward/action_rate_l2"):
                        np.testing.assert_allclose(float(lg[name]), olog[name], rtol=2e-3, atol=1e-6)
            assert want[1][0] >= 16 and float(kept[1]["Episode_Termination/time_out"]) == 16.0
    # never read inside the window: the env materialised it before the ring wrapped.  Step 2 reset nobody, so its log is the log of
    # the most recent step that did (step 1), as in the reference
    if want[2][0] == 0:
        assert float(kept[2]["Episode_Termination/time_out"]) == 16.0
        assert float(kept[2]["Episode_Reward/action_rate_l2"]) == float(kept[1]["Episode_Reward/action_rate_l2"])
    else:  # somebody fell over in step 2: its own log
        np.testing.assert_allclose(float(kept[2]["Episode_Reward/action_rate_l2"]), want[2][1]["Episode_Reward/action_rate_l2"], rtol=2e-3, atol=1e-6)
    assert float(kept[1]["Episode_Termination/time_out"]) == 16.0  # materialised at step 30: stays
    env.close()
