"""command_levels_lin_vel / command_levels_ang_vel (reference VEL/mdp/curriculums.py:21-94).

1. the widening rule of the oracle (`OracleEnv._apply_cmd_levels`) against the reference's own functions, imported from
   /root/reference when it is present (this container) and against the committed fixture otherwise
   (tests/golden/command_levels.npz, written by tools/gen_golden_command_levels.py);
2. the lane program (host emulator) against the oracle on a short-episode task with both curricula enabled: live range table,
   commands drawn from it, `Curriculum/command_levels_*` values.
"""
import os

import numpy as np
import pytest

from helpers import host_view, make_pair

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A1_FLAT = "RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0"

GOLD = os.path.join(ROOT, "tests", "golden", "command_levels.npz")


def _oracle_trace(case):
    """Replays one fixture case through the oracle's rule: -> [n_calls, 6] live ranges after each call."""
    from oracle.env import OracleEnv
    from robot_lab_amd.scene import build_world, load_bundle

    desc, extra = load_bundle(A1_FLAT)
    t = desc.task
    t.cur_cmd_lin = t.cur_cmd_ang = 1
    names = list(desc.reward_names)
    t.cur_cmd_lin_term, t.cur_cmd_ang_term = names.index("track_lin_vel_xy_exp"), names.index("track_ang_vel_z_exp")
    t.cur_cmd_lin_mult[0], t.cur_cmd_lin_mult[1] = case["mult_lin"]
    t.cur_cmd_ang_mult[0], t.cur_cmd_ang_mult[1] = case["mult_ang"]
    for i in range(3):
        t.cmd_range[i][0], t.cmd_range[i][1] = case["ranges"][i]
    t.rewards[t.cur_cmd_lin_term].weight, t.rewards[t.cur_cmd_ang_term].weight = case["weights"]
    t.episode_length_s = float(case["episode_length_s"])
    h, to, eo = build_world(desc, extra, 4, 0)
    env = OracleEnv(desc, h, to, 4, 0, eo)
    out = []
    for ml, ma in zip(case["mean_lin"], case["mean_ang"]):
        env._cmd_levels_pending = [float(ml), float(ma)]
        env._apply_cmd_levels()
        out.append(env.cmd_levels.reshape(-1).copy())
    return np.array(out)


def _cases():
    g = np.load(GOLD, allow_pickle=True)
    return g["cases"].tolist()


def test_rule_matches_golden():
    cases = _cases()
    assert len(cases) >= 4
    for c in cases:
        got = _oracle_trace(c)
        np.testing.assert_allclose(got, c["trace"], rtol=0, atol=1e-6, err_msg=str(c["name"]))
        assert c["trace"][-1].tolist() != c["trace"][0].tolist() or c["name"].startswith("never")


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present (GPU box)")
def test_golden_is_what_the_reference_does():
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_golden_command_levels as gen

    fresh = gen.make_cases()
    for a, b in zip(fresh, _cases()):
        assert a["name"] == b["name"]
        np.testing.assert_array_equal(a["trace"], b["trace"])


def _short_task(desc):
    """Both curricula on, 0.4 s episodes (20 steps).  lin is driven by the lin_vel_z_l2 penalty (weight -2: `sum / T > 0.8 w` holds
    unless the base moves vertically at ~0.9 m/s rms, so it widens at every decision); ang by track_lin_vel_xy_exp with a narrow
    kernel, which a robot dropped with +-0.5 m/s never satisfies in 0.4 s - so one range moves and the other must not."""
    t = desc.task
    names = list(desc.reward_names)
    t.cur_cmd_lin = t.cur_cmd_ang = 1
    t.cur_cmd_lin_term, t.cur_cmd_ang_term = names.index("lin_vel_z_l2"), names.index("track_lin_vel_xy_exp")
    t.cur_cmd_lin_mult[0], t.cur_cmd_lin_mult[1] = 0.1, 0.35  # clamps at the third widening
    t.cur_cmd_ang_mult[0], t.cur_cmd_ang_mult[1] = 0.5, 1.0
    t.rewards[t.cur_cmd_ang_term].p[0] = 0.01
    t.episode_length_s = 0.4
    return desc


def test_emu_matches_oracle_with_curricula(emu_lib):
    desc, ora, emu = make_pair(A1_FLAT, 24, 3, emu_lib, mutate=_short_task)
    L = emu.max_episode_length
    assert L == 20 and ora.max_episode_length == 20
    ora.reset()
    emu.reset()
    rng = np.random.default_rng(0)
    lv0 = host_view(emu, "CMD_LEVELS").copy()
    np.testing.assert_allclose(lv0[:6], ora.cmd_levels.reshape(-1), atol=1e-7)
    np.testing.assert_allclose(lv0[:2], np.array([desc.task.cmd_range[0][0], desc.task.cmd_range[0][1]]) * 0.1, atol=1e-7)
    widened = 0
    for k in range(3 * L + 2):
        a = rng.uniform(-0.3, 0.3, (24, emu.num_actions)).astype(np.float32)
        emu.step(a.ctypes.data)
        ora.step(a)
        lv = host_view(emu, "CMD_LEVELS")
        np.testing.assert_allclose(lv[:6], ora.cmd_levels.reshape(-1), atol=1e-6, err_msg=f"step {k}")
        assert np.all(lv[8:12] == 0), "decision accumulators are cleared behind every deciding step"
        np.testing.assert_allclose(host_view(emu, "COMMAND")[:24], ora.vel_command_b, atol=2e-5, err_msg=f"step {k}")
        widened += int(not np.allclose(lv[:6], lv0[:6]))
        lv0 = lv.copy()
    # every env times out at step 20, 40, 60 -> three decisions: 0.1 -> 0.2 -> 0.3 -> 0.35 (clamped)
    assert widened == 3 and ora.cmd_levels_immediate  # the reference order: the deciding step's own resets draw from the widened range
    np.testing.assert_allclose(lv[:6], [-0.35, 0.35, -0.35, 0.35, -0.5, 0.5], atol=1e-6)
    # commands are drawn inside the live range, not the table's
    assert np.abs(host_view(emu, "COMMAND")[:24, 0]).max() <= lv[1] + 1e-6


def test_deferred_decision_is_bounded_against_the_reference_order():
    """ADVICE r2 (medium), history: the reference runs curriculum_manager.compute() FIRST inside _reset_idx, so the envs reset in the
    deciding step already draw their commands from the widened range (oracle: cmd_levels_immediate = True, the default - the rule
    itself is pinned by test_golden_is_what_the_reference_does).  The round-2 kernels decided behind the launch (False): those envs -
    and the deciding step's heading clip - still used the old range.  Round 3 splits such a step around the decision (csrc/
    rl_env_host.h step(): head launch, decision, tail launch), and test_emu_matches_oracle_with_curricula now holds the lane
    program to the reference order on every step.  This test keeps the measure of what the deferred order cost, between the two
    modes of the oracle on the same seeds and actions (worst case by construction: every env times out on the deciding step):
      * the live range tables agree from the end of the deciding step on (the decision itself is the same);
      * commands differ only in envs reset on a deciding step: per component by at most the widening step 0.1 (same uniform sample,
        range ends moved by -0.1 / +0.1) - unless the small-command rule (VEL/mdp/commands.py:43-47: |v_xy| <= threshold -> 0) zeroes
        the command on one side only, which bounds the difference by threshold + 0.1 sqrt(2) -, and only until their next resample,
        after which the two runs draw from identical ranges again."""
    from oracle.env import OracleEnv
    from robot_lab_amd.scene import build_world, load_bundle

    def make(immediate):
        desc, extra = load_bundle(A1_FLAT)
        _short_task(desc)
        h, to, eo = build_world(desc, extra, 24, 0)
        env = OracleEnv(desc, h, to, 24, 3, eo)
        env.cmd_levels_immediate = immediate
        env.reset()
        return env

    ref, ker = make(True), make(False)
    L = ref.max_episode_length
    rng = np.random.default_rng(0)
    worst, differing_steps = 0.0, 0
    for k in range(3 * L + 2):
        a = rng.uniform(-0.3, 0.3, (24, ref.desc.model.num_dof)).astype(np.float32)
        ref.step(a)
        ker.step(a)
        np.testing.assert_allclose(ker.cmd_levels, ref.cmd_levels, atol=1e-7, err_msg=f"range table after step {k}")
        d = np.abs(ker.vel_command_b - ref.vel_command_b).max()
        worst = max(worst, float(d))
        differing_steps += int(d > 1e-7)
        deciding = (k + 1) % L == 0
        if not deciding and d > 1e-7:
            # a difference outside a deciding step is one inherited from it (the command is held for the resampling period)
            assert ref.cmd_time_left.min() > 0
    bound = float(ref.desc.task.cmd_small_threshold) + 0.1 * np.sqrt(2.0)
    assert 0.0 < worst <= bound + 1e-6, (worst, bound)  # the deviation exists (this test would notice its removal) and is bounded
    assert differing_steps > 0
