#!/usr/bin/env python
"""Episode statistics of the fp64 oracle and of its fp32-disturbed twin over a long free run - the fixture of the distributional
parity test (tests/test_gpu_episode_stats.py; SURVEY.md section 7 "Chaotic divergence": "...plus distributional agreement of
episode statistics").  The oracle needs minutes for 512 envs x 300 steps, which the GPU box's test tier should not spend, and it
is deterministic, so its per-env statistics are committed:

    python tools/gen_golden_episode_stats.py            # A1 Rough and G1 Rough -> tests/golden/episode_stats_{A1,G1}.npz

Same construction as the free-run parity tests (helpers.OracleWithTwin): the twin restarts from the oracle's reset state perturbed
by a relative 1e-6 and solves its linear systems in fp32; both take the action stream default_rng(ACTION_SEED).uniform(-1, 1)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import OracleWithTwin, run_episode_stats, staggered_episode_lengths  # noqa: E402
from oracle.env import OracleEnv  # noqa: E402
from robot_lab_amd.scene import build_world, load_bundle  # noqa: E402

CONFIGS = {"A1": ("RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0", 512, 300), "G1": ("RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0", 256, 300)}
SEED, ACTION_SEED = 42, 20260925


def side(env):
    def step(a):
        env.step(a)
        return env.reward.copy(), env.reward_terms.copy(), env.terminated.copy(), env.time_outs.copy()
    return step, env.read_state


def main():
    for key in sys.argv[1:] or list(CONFIGS):
        task, N, steps = CONFIGS[key]
        desc, extra = load_bundle(task)
        h, to, eo = build_world(desc, extra, N, 0)
        two = OracleWithTwin(lambda: OracleEnv(desc, h, to, N, SEED, eo))
        two.reset()
        for env in (two.ora, two.twin):
            env.episode_length_buf[:] = staggered_episode_lengths(N, env.max_episode_length)
        T, A, L = desc.task.n_rewards, desc.model.num_dof, desc.terrain.num_rows
        out = {}
        for name, env in (("oracle", two.ora), ("twin", two.twin)):
            t0 = time.time()
            env.phys.margins = None
            st = run_episode_stats(*side(env), N, T, L, A, steps, ACTION_SEED)
            for k, v in st.items():
                out[f"{name}/{k}"] = v.astype(np.float64)
            print(f"{key} {name}: {N} envs x {steps} steps in {time.time() - t0:.0f} s; reward/step {st['reward'].mean():+.5f}, "
                  f"terminated/step {st['terminated'].mean():.4f}, time-out/step {st['time_out'].mean():.4f}, contacts {st['contacts'].mean():.3f}, level {st['level'].mean():.3f}", flush=True)
        path = os.path.join(os.environ.get("RL_GOLDEN_DIR", os.path.join(ROOT, "tests", "golden")), f"episode_stats_{key}.npz")
        np.savez_compressed(path, task=task, n_envs=N, steps=steps, seed=SEED, action_seed=ACTION_SEED, **out)
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
