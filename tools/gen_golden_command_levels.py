"""Writes tests/golden/command_levels.npz: the reference's OWN command_levels_lin_vel / command_levels_ang_vel
(/root/reference .../velocity/mdp/curriculums.py:21-94), imported from the checkout and driven with a stand-in env object,
on a handful of episode-sum sequences.  tests/test_command_levels.py replays the cases through the oracle's restatement of the
rule.  Run in the build container (the reference checkout does not exist on the GPU box)."""
import importlib.util
import os
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/source/robot_lab/robot_lab/tasks/manager_based/locomotion/velocity/mdp/curriculums.py"


def _ref():
    spec = importlib.util.spec_from_file_location("_ref_curriculums", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _fake_env(ranges, weights, episode_length_s, step_dt=0.02):
    r = types.SimpleNamespace(lin_vel_x=tuple(ranges[0]), lin_vel_y=tuple(ranges[1]), ang_vel_z=tuple(ranges[2]))
    term = types.SimpleNamespace(cfg=types.SimpleNamespace(ranges=r))
    sums = {"lin": torch.zeros(8), "ang": torch.zeros(8)}
    cfgs = {"lin": types.SimpleNamespace(weight=weights[0]), "ang": types.SimpleNamespace(weight=weights[1])}
    env = types.SimpleNamespace(
        device="cpu", common_step_counter=0, max_episode_length=int(round(episode_length_s / step_dt)), max_episode_length_s=episode_length_s,
        command_manager=types.SimpleNamespace(get_term=lambda name: term),
        reward_manager=types.SimpleNamespace(_episode_sums=sums, get_term_cfg=lambda name: cfgs[name]))
    return env, r


def run_case(name, ranges, weights, mult_lin, mult_ang, episode_length_s, mean_lin, mean_ang):
    ref = _ref()
    env, r = _fake_env(ranges, weights, episode_length_s)
    ids = torch.arange(8)
    # the call at counter 0 (env.reset() before the first step) installs the initial ranges; zero episode sums never pass the test
    ref.command_levels_lin_vel(env, ids, "lin", mult_lin)
    ref.command_levels_ang_vel(env, ids, "ang", mult_ang)
    trace = []
    for k, (ml, ma) in enumerate(zip(mean_lin, mean_ang)):
        env.common_step_counter = (k + 1) * env.max_episode_length
        # eight envs whose mean is the prescribed value
        env.reward_manager._episode_sums["lin"] = torch.full((8,), float(ml)) + torch.linspace(-0.5, 0.5, 8)
        env.reward_manager._episode_sums["ang"] = torch.full((8,), float(ma)) + torch.linspace(-0.25, 0.25, 8)
        # a call off the episode-length grid must change nothing
        env.common_step_counter += 1
        before = (tuple(r.lin_vel_x), tuple(r.lin_vel_y), tuple(r.ang_vel_z))
        ref.command_levels_lin_vel(env, ids, "lin", mult_lin)
        ref.command_levels_ang_vel(env, ids, "ang", mult_ang)
        assert before == (tuple(r.lin_vel_x), tuple(r.lin_vel_y), tuple(r.ang_vel_z))
        env.common_step_counter -= 1
        up_l = ref.command_levels_lin_vel(env, ids, "lin", mult_lin)
        up_a = ref.command_levels_ang_vel(env, ids, "ang", mult_ang)
        row = np.array(list(r.lin_vel_x) + list(r.lin_vel_y) + list(r.ang_vel_z), dtype=np.float32)
        assert float(up_l) == row[1] and float(up_a) == row[5]  # what Curriculum/command_levels_* logs
        trace.append(row)
    return dict(name=name, ranges=np.array(ranges, dtype=np.float32), weights=np.array(weights, dtype=np.float32),
                mult_lin=np.array(mult_lin, dtype=np.float32), mult_ang=np.array(mult_ang, dtype=np.float32),
                episode_length_s=float(episode_length_s), mean_lin=np.array(mean_lin, dtype=np.float32),
                mean_ang=np.array(mean_ang, dtype=np.float32), trace=np.array(trace, dtype=np.float32))


def make_cases():
    a1 = [(-1.0, 1.0), (-1.0, 1.0), (-1.0, 1.0)]  # unitree_a1 ranges (velocity_env_cfg.py:115-119)
    w = (3.0, 1.5)
    hi_l, hi_a = 0.9 * 3.0 * 20.0, 0.9 * 1.5 * 20.0
    return [
        run_case("widen_until_clamped", a1, w, (0.1, 1.0), (0.1, 1.0), 20.0, [hi_l] * 12, [hi_a] * 12),
        run_case("never_widens", a1, w, (0.1, 1.0), (0.1, 1.0), 20.0, [0.5 * hi_l] * 4, [0.5 * hi_a] * 4),
        run_case("threshold_edge", a1, w, (0.2, 1.0), (0.5, 1.0), 20.0,
                 [0.8 * 3.0 * 20.0 - 0.01, 0.8 * 3.0 * 20.0 + 0.01, 0.0, hi_l], [hi_a, 0.8 * 1.5 * 20.0 - 0.01, 0.8 * 1.5 * 20.0 + 0.01, 0.0]),
        run_case("asymmetric_ranges", [(-0.5, 2.0), (-0.3, 0.3), (-1.5, 1.0)], (1.0, 0.5), (0.1, 0.6), (0.3, 1.0), 10.0,
                 [9.5] * 15, [4.9] * 15),
        run_case("lin_only_passes", a1, w, (0.1, 1.0), (0.1, 1.0), 20.0, [hi_l] * 3, [0.0] * 3),
    ]


if __name__ == "__main__":
    cases = make_cases()
    out = os.path.join(os.environ.get("RL_GOLDEN_DIR", os.path.join(ROOT, "tests", "golden")), "command_levels.npz")
    np.savez_compressed(out, cases=np.array(cases, dtype=object))
    for c in cases:
        print(c["name"], c["trace"][0], "->", c["trace"][-1])
