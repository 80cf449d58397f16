#!/usr/bin/env python
"""Collection loop of `play.py:244-248` / the rsl_rl rollout: actions = policy(obs); obs, ... = env.step(actions),
with both sides on the HIP kernels of this repo (random actor weights: the arithmetic does not depend on them).
    python tools/bench_rollout.py [task] [num_envs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robot_lab_amd.env import ManagerBasedRLEnv  # noqa: E402
from robot_lab_amd.policy import MlpPolicy  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = ManagerBasedRLEnv(task, num_envs=N, seed=42, device="cuda:0")
obs, _ = env.reset()
dims = [obs["policy"].shape[1], 512, 256, 128, env.num_actions]
rng = np.random.default_rng(0)
ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(4)]
bs = [np.zeros(dims[i + 1], dtype=np.float32) for i in range(4)]
policy = MlpPolicy(ws, bs, "elu", device="cuda:0")
with torch.inference_mode():
    for _ in range(50):
        obs, *_ = env.step(policy(obs))
    torch.cuda.synchronize()
    K = 1000
    t0 = time.perf_counter()
    for _ in range(K):
        obs, rew, term, tout, extras = env.step(policy(obs))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{task} N={N}: policy + env.step loop {1e6 * dt / K:.1f} us / iteration = {N * K / dt / 1e6:.1f} M env-steps/s (finite: {bool(torch.isfinite(rew).all())})")
