#!/usr/bin/env python
"""env.step() throughput of every compiled task id (robot_lab_amd/data/*.json): 4096 envs on the quadruped instances, 2048 on the trunk +
limbs instance - with the step kernel the library picks by itself (a built-in Spec for the eight BASELINE tasks, else the term-stack
interpreter) and, with --jit, with the step kernel specialised on the task at run time (robot_lab_amd/jit.py; the compile is not timed).
    python tools/bench_every_task.py [--jit] > profiles/<round>_all_tasks.txt"""
import glob
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robot_lab_amd.env import ManagerBasedRLEnv  # noqa: E402
from robot_lab_amd.scene import DATA_DIR, load_bundle  # noqa: E402

JIT = "--jit" in sys.argv


def measure(task, N, specialise):
    t0 = time.time()
    env = ManagerBasedRLEnv(task, num_envs=N, seed=1, device="cuda:0", specialise=specialise)
    t_create = time.time() - t0
    env.reset()
    g = torch.Generator(device="cuda:0").manual_seed(0)
    acts = [torch.rand(N, env.num_actions, device="cuda:0", generator=g) * 2 - 1 for _ in range(8)]
    for i in range(30):
        env.step(acts[i % 8])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    K = 300
    e0.record()
    for i in range(K):
        obs, rew, *_ = env.step(acts[i % 8])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / K * 1e3
    ok = bool(torch.isfinite(obs["critic"]).all() and torch.isfinite(rew).all())
    kern = env.step_kernel
    env.close()
    return us, ok, kern, t_create


print(f"{'task':62} {'envs':>5} {'dof':>4} {'chains':>7} {'us/step':>8} {'M env-steps/s':>14} finite  step kernel" + ("   | specialised at run time: us/step  M env-steps/s  gain  create s" if JIT else ""))
for path in sorted(glob.glob(os.path.join(DATA_DIR, "*.json"))):
    task = os.path.basename(path)[:-5]
    m = load_bundle(task)[0].model
    quad = m.num_trunk == 0 and len({m.chain_nj[k] for k in range(4)}) == 1 and m.chain_len <= 4
    N = 4096 if quad else 2048
    us, ok, kern, _ = measure(task, N, False)
    line = f"{task:62} {N:5d} {m.num_dof:4d} {m.num_chains}x{m.chain_len:<5d} {us:8.1f} {N / us:14.2f} {ok}  {kern}"
    if JIT and kern == "interpreter":
        us2, ok2, kern2, tc = measure(task, N, True)
        line += f"   | {us2:8.1f} {N / us2:10.2f} {100 * (us / us2 - 1):+6.1f} % {tc:6.1f}  {kern2} {ok2}"
    print(line, flush=True)
