#!/usr/bin/env python
"""A/B of env-kernel builds inside ONE gpurun call (boxes differ by a few %, so variants are only compared within a call):
    python tools/ab_bench.py [--task ID] [--num-envs N] [--rounds R] [--steady] lib1.so lib2.so ...
Each library is a full C-ABI build (RL_ENV_LIB); kernel time by HIP events over 300 launches, `rounds` interleaved repeats.
A variant `label:KEY=VAL,KEY=VAL[@lib.so]` runs the product library (or lib.so) under those environment variables instead
(RL_ENV_SPEC=0 / 1: the term-stack interpreter against the kernel specialised on the task).  --steady: episode clocks spread over
[0, max) and 300 untimed steps first, as bench.py - some env resets on every step of the window."""
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
task, N, reps, steady = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
env = ManagerBasedRLEnv(task, num_envs=N, seed=42, device="cuda:0")
g = torch.Generator(device="cuda").manual_seed(1234)
ring = [torch.rand(N, env.num_actions, device="cuda", generator=g) * 2 - 1 for _ in range(32)]
env.reset()
if steady:
    env.episode_length_buf = torch.randint(0, int(env.max_episode_length), (N,), device="cuda", generator=g)
    for i in range(300): env.step(ring[i %% 32])
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
print(json.dumps(dict(us=out, spec=nat.spec_id(), ept=nat.envs_per_wavefront())))
''' % ROOT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steady", action="store_true")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    import json

    res = {l: [] for l in a.libs}
    info = {}
    for r in range(a.rounds):
        for l in a.libs:
            env = dict(os.environ)
            if ":" in l and "=" in l:  # label:KEY=VAL,...[@lib.so]
                spec, _, lib = l.split(":", 1)[1].partition("@")
                env.update(kv.split("=", 1) for kv in spec.split(",") if kv)
                if lib:
                    env["RL_ENV_LIB"] = os.path.abspath(lib)
            else:
                env["RL_ENV_LIB"] = os.path.abspath(l)
            p = subprocess.run([sys.executable, "-c", CHILD, a.task, str(a.num_envs), "3", "1" if a.steady else "0"], env=env, capture_output=True, text=True)
            if p.returncode != 0:
                print(l, "FAILED", p.stderr[-400:])
                continue
            d = json.loads(p.stdout.strip().splitlines()[-1])
            res[l] += d["us"]
            info[l] = f"spec {d['spec']} ept {d['ept']}"
    for l, v in res.items():
        if v:
            name = l.split(":", 1)[0] if (":" in l and "=" in l) else os.path.basename(l)
            print(f"{name:28s} {a.task.split('Velocity-')[1]:28s} N={a.num_envs}{' steady' if a.steady else ''}: min {min(v):7.2f} us  median {sorted(v)[len(v)//2]:7.2f} us  ({len(v)} samples, {info.get(l, '')})")


if __name__ == "__main__":
    main()
