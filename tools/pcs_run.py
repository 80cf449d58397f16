#!/usr/bin/env python
"""Workload for a rocprofv3 PC-sampling run: step one task's env kernel N times with random actions (nothing else on the GPU).
    RL_ENV_LIB=<variant.so> python tools/pcs_run.py [task] [num_envs] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from robot_lab_amd.env import ManagerBasedRLEnv

task = sys.argv[1] if len(sys.argv) > 1 else "RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 600
env = ManagerBasedRLEnv(task, num_envs=N, seed=42, device="cuda:0")
g = torch.Generator(device="cuda").manual_seed(1234)
ring = [torch.rand(N, env.num_actions, device="cuda", generator=g) * 2 - 1 for _ in range(32)]
env.reset()
nat, st = env._native, env._stream()
for i in range(steps):
    nat.step(ring[i % 32].data_ptr(), st)
torch.cuda.synchronize()
print("stepped", steps)
