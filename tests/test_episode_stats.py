"""CPU tier of the distributional parity test (tests/test_gpu_episode_stats.py is the `-m gpu` form at 512 / 256 envs x 300 steps):
the lane program (CPU lane emulator, the source hipcc compiles) against the fp64 oracle and its fp32-disturbed twin, run live at a
size the oracle finishes in seconds; plus the committed fixtures' own sanity (they are what the GPU tier is judged against)."""
import os

import numpy as np
import pytest

from helpers import OracleWithTwin, compare_episode_stats, emu_read_state, host_view, run_episode_stats, staggered_episode_lengths
from oracle.env import OracleEnv
from robot_lab_amd.scene import build_world, load_bundle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("task,N,steps", [("RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0", 48, 100), ("RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0", 16, 60),
                                          ("RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0", 16, 60)])  # (GR1: the six-joint-spine instance)
def test_lane_program_episode_statistics(task, N, steps, emu_lib):
    from robot_lab_amd.capi import NativeEnv

    desc, extra = load_bundle(task)
    h, to, eo = build_world(desc, extra, N, 0)
    two = OracleWithTwin(lambda: OracleEnv(desc, h, to, N, 42, eo))
    two.reset()
    ep = staggered_episode_lengths(N, two.ora.max_episode_length)
    ep[::3] = two.ora.max_episode_length - 1 - (np.arange(len(ep[::3])) * 7) % steps  # short run: a third of the envs time out inside it
    for env in (two.ora, two.twin):
        env.episode_length_buf[:] = ep
    T, A = desc.task.n_rewards, desc.model.num_dof
    stats = {}
    for name, env in (("oracle", two.ora), ("twin", two.twin)):
        env.phys.margins = None

        def step(a, env=env):
            env.step(a)
            return env.reward.copy(), env.reward_terms.copy(), env.terminated.copy(), env.time_outs.copy()

        stats[name] = run_episode_stats(step, env.read_state, N, T, 0, A, steps, 7)
    nat = NativeEnv(desc, h, to, eo, N, 42, 0, emu_lib)
    nat.reset()
    host_view(nat, "EPISODE_LENGTH")[:] = ep

    def nstep(a):
        nat.step(np.ascontiguousarray(a).ctypes.data)
        return (host_view(nat, "REWARD").copy(), host_view(nat, "REWARD_TERMS")[:, :N].copy(), host_view(nat, "TERMINATED").astype(bool),
                host_view(nat, "TIME_OUT").astype(bool))

    got = run_episode_stats(nstep, lambda: emu_read_state(nat), N, T, 0, A, steps, 7)
    compare_episode_stats(got, stats["oracle"], stats["twin"])
    nat.close()


@pytest.mark.parametrize("key", ["A1", "G1"])
def test_fixture_is_eventful_and_self_consistent(key):
    """The committed oracle / twin statistics: finite, the run contains terminations, and the twin sits inside its own interval
    (compare_episode_stats(twin, oracle, twin) is a tautology for the mean difference but exercises every statistic's floor)."""
    fx = np.load(os.path.join(GOLDEN, f"episode_stats_{key}.npz"))
    ora = {k.split("/", 1)[1]: fx[k] for k in fx.files if k.startswith("oracle/")}
    twin = {k.split("/", 1)[1]: fx[k] for k in fx.files if k.startswith("twin/")}
    assert set(ora) == set(twin) and all(np.isfinite(v).all() for v in ora.values())
    # eventful: episodes end inside the run (A1: the staggered time-outs; G1 under random actions falls - most of its resets are terminations)
    assert ora["time_out"].mean() > 1e-4 and (ora["time_out"] + ora["terminated"]).mean() > 5e-4
    assert ora["contacts"].mean() > 0.5 and len(ora["reward"]) == int(fx["n_envs"])
    compare_episode_stats(twin, ora, twin)
    # reward = sum of its terms, also on average
    total = sum(ora[k] for k in ora if k.startswith("term_"))
    np.testing.assert_allclose(total, ora["reward"], rtol=1e-9, atol=1e-12)
