#!/usr/bin/env python
"""Throughput of the env-step kernel vs environments per GPU (A1 Rough): where the chip fills.
    python tools/sweep_envs.py [task] [N1,N2,...]      (RL_ENV_SUB=1 selects the one-lane-per-limb mapping for the whole sweep)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robot_lab_amd.env import ManagerBasedRLEnv  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
print(f"{task}\n{'envs':>8} {'us/step':>10} {'M env-steps/s':>14}")
SIZES = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (512, 1024, 2048, 4096, 8192, 16384, 32768, 65536)
for N in SIZES:
    env = ManagerBasedRLEnv(task, num_envs=N, seed=42, device="cuda:0")
    env.reset()
    A = env.num_actions
    g = torch.Generator(device="cuda:0").manual_seed(0)
    acts = [torch.rand(N, A, device="cuda:0", generator=g) * 2 - 1 for _ in range(8)]
    native, stream = env._native, env._stream()
    ptrs = [a.data_ptr() for a in acts]
    for i in range(30):
        native.step(ptrs[i % 8], stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    K = 200
    e0.record()
    for i in range(K):
        native.step(ptrs[i % 8], stream)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / K * 1e3
    print(f"{N:8d} {us:10.1f} {N / us:14.2f}")
    env.close()
