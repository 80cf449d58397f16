#!/usr/bin/env python
"""A/B of env-kernel builds inside ONE gpurun call (boxes differ by a few %, so variants are only compared within a call):
    python tools/ab_bench.py [--task ID] [--num-envs N] [--rounds R] lib1.so lib2.so ...
Each library is a full C-ABI build (RL_ENV_LIB); kernel time by HIP events over 300 launches, `rounds` interleaved repeats."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
sys.path.insert(0, %r)
import torch
from robot_lab_amd.env import ManagerBasedRLEnv
task, N, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
env = ManagerBasedRLEnv(task, num_envs=N, seed=42, device="cuda:0")
g = torch.Generator(device="cuda").manual_seed(1234)
ring = [torch.rand(N, env.num_actions, device="cuda", generator=g) * 2 - 1 for _ in range(32)]
env.reset()
nat, st = env._native, env._stream()
ptrs = [r.data_ptr() for r in ring]
for i in range(150): nat.step(ptrs[i %% 32], st)
out = []
for r in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(300): nat.step(ptrs[i %% 32], st)
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 300 * 1e3)
print(json.dumps(out))
''' % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    res = {l: [] for l in a.libs}
    for r in range(a.rounds):
        for l in a.libs:
            env = dict(os.environ, RL_ENV_LIB=os.path.abspath(l))
            p = subprocess.run([sys.executable, "-c", CHILD, a.task, str(a.num_envs), "3"], env=env, capture_output=True, text=True)
            if p.returncode != 0:
                print(l, "FAILED", p.stderr[-400:])
                continue
            import json
            res[l] += json.loads(p.stdout.strip().splitlines()[-1])
    for l, v in res.items():
        if v:
            print(f"{os.path.basename(l):28s} {a.task.split('Velocity-')[1]:28s} N={a.num_envs}: min {min(v):7.2f} us  median {sorted(v)[len(v)//2]:7.2f} us  ({len(v)} samples)")


if __name__ == "__main__":
    main()
