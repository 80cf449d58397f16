"""`-m gpu`: the command-range curricula (`command_levels_lin_vel / _ang_vel`, VEL/mdp/curriculums.py:21-94) on the HIP path, in
the REFERENCE's order: the decision is taken first inside the reset of the deciding step, so the commands that step's own resets
draw, and its heading clip, already use the widened range.  The kernels get there by splitting such a step
into a head launch (to the rewards; collects the episode sums of the envs about to reset), the one-thread decision and a tail
launch (resets, commands, push, observations) - csrc/rl_env_host.h step(), csrc/env_terms.h step_head / step_tail.  Compared with
the fp64 oracle in its reference-exact mode on every step of three short episodes; the rule itself is pinned to the reference's
functions by tests/test_command_levels.py (golden traces)."""
import numpy as np
import pytest

from test_command_levels import A1_FLAT, _short_task

pytestmark = pytest.mark.gpu


def test_hip_matches_oracle_in_the_reference_order():
    import torch

    from oracle.env import OracleEnv
    from robot_lab_amd.env import ManagerBasedRLEnv
    from robot_lab_amd.scene import build_world, load_bundle

    N = 64
    desc, extra = load_bundle(A1_FLAT)
    _short_task(desc)
    env = ManagerBasedRLEnv(desc=desc, extra=extra, num_envs=N, seed=3, device="cuda:0")
    h, to, eo = build_world(desc, extra, N, 0)
    ora = OracleEnv(desc, h, to, N, 3, eo)
    assert ora.cmd_levels_immediate and env.max_episode_length == ora.max_episode_length == 20
    obs, _ = env.reset()
    o = ora.reset()
    rng = np.random.default_rng(0)
    L = env.max_episode_length
    widened, lv0 = 0, env.command_levels.cpu().numpy().copy()
    for k in range(3 * L + 2):
        a = rng.uniform(-0.3, 0.3, (N, env.num_actions)).astype(np.float32)
        obs, rew, term, tout, _ = env.step(torch.from_numpy(a).cuda())
        o = ora.step(a)
        lv = env.command_levels.cpu().numpy()
        np.testing.assert_allclose(lv[:6], ora.cmd_levels.reshape(-1), atol=1e-6, err_msg=f"live ranges after step {k}")
        # the commands of the deciding step itself (every env times out on it and resamples): drawn from the range decided IN that step
        np.testing.assert_allclose(env.command_manager.get_command("base_velocity").cpu().numpy(), ora.vel_command_b, atol=2e-5, err_msg=f"commands after step {k}")
        np.testing.assert_array_equal((term | tout).cpu().numpy(), ora.terminated | ora.time_outs)
        # the observation rows of that step show those commands (the tail launch computes them after the decision); A1 policy row:
        # base_ang_vel, projected_gravity, velocity_commands (columns 6:9, scale 1, no noise), ...
        np.testing.assert_allclose(obs["policy"].cpu().numpy()[:, 6:9], ora.vel_command_b, atol=2e-5, err_msg=f"command columns of the policy row, step {k}")
        widened += int(not np.allclose(lv[:6], lv0[:6]))
        lv0 = lv.copy()
    assert widened == 3
    np.testing.assert_allclose(lv[:6], [-0.35, 0.35, -0.35, 0.35, -0.5, 0.5], atol=1e-6)
    env.close()
