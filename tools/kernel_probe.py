#!/usr/bin/env python
"""Kernel-time breakdown by ablation (GPU): variants of the task descriptor -> mean kernel ms."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from robot_lab_amd.env import ManagerBasedRLEnv
from robot_lab_amd.scene import load_bundle


def run(task, N=4096, decim=None, zero=False, steps=200, mutate=None, lib_path=None):
    desc, extra = load_bundle(task)
    if decim is not None:
        desc.sim.decimation = decim
    if mutate:
        mutate(desc)
    env = ManagerBasedRLEnv(None, desc=desc, extra=extra, num_envs=N, seed=42, device="cuda:0", lib_path=lib_path)
    env.reset()
    A = env.num_actions
    ring = [(torch.zeros(N, A, device="cuda") if zero else torch.rand(N, A, device="cuda") * 2 - 1) for _ in range(8)]
    st = env._stream()
    for i in range(100):
        env._native.step(ring[i % 8].data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(steps):
        env._native.step(ring[i % 8].data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    env.close()
    return ms


def no_rewards(d):
    d.task.n_rewards = 0


def no_noise(d):
    d.task.policy_corrupt = 0


if __name__ == "__main__":
    R, F = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0", "RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0"
    for name, kw in [
        ("rough d4 random", dict(task=R)), ("rough d4 zero", dict(task=R, zero=True)), ("rough d1 random", dict(task=R, decim=1)),
        ("rough d2 random", dict(task=R, decim=2)), ("flat d4 random", dict(task=F)), ("flat d1 random", dict(task=F, decim=1)),
        ("rough d4 no rewards", dict(task=R, mutate=no_rewards)), ("rough d4 no noise", dict(task=R, mutate=no_noise)),
        ("rough d4 N=1024", dict(task=R, N=1024)), ("rough d4 N=16384", dict(task=R, N=16384)), ("rough d4 N=65536", dict(task=R, N=65536)),
    ]:
        print(f"{name:24s} {1e3 * run(**kw):8.1f} us", flush=True)
