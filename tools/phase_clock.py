#!/usr/bin/env python
"""Phase breakdown of the env-step kernel from in-kernel cycle-counter stamps.

Needs a probe build of the HIP library that writes s_memtime stamps (one per phase boundary, lane 0
of every wavefront, after s_waitcnt 0) into the unused rows 24.. of the REWARD_TERMS buffer; the recipe that patches a
copy of robot_lab_amd/csrc and builds tools/_probe/librl_env_stamp.so is in DESIGN.md ("Measurement").
Usage: RL_ENV_LIB=tools/_probe/librl_env_stamp.so python tools/phase_clock.py [task] [num_envs]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robot_lab_amd.env import ManagerBasedRLEnv, _DevView  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = ManagerBasedRLEnv(task, num_envs=N, seed=42, device="cuda:0")
env.reset()
A = env.num_actions
g = torch.Generator(device="cuda:0").manual_seed(1)
names = os.environ["RL_STAMP_NAMES"].split(",") if os.environ.get("RL_STAMP_NAMES") else ["tables->LDS", "load+actions", "substep1", "substep2", "substep3", "substep4", "task regs+terminations", "reward terms", "ep_sums epilogue",
         "reset+commands+push", "policy obs", "critic obs", "flush obs", "store"]
acc = np.zeros(14)
cnt = 0
nwave = N * 16 // 64 if os.environ.get("RL_ENV_SUB", "4") != "1" else N * 4 // 64
rt = env._bufs["REWARD_TERMS"]
Np = rt.shape[1]
buf = torch.as_tensor(_DevView(rt.data_ptr() + 24 * Np * 4, (nwave * 16,), np.int64, env), device="cuda:0")
for s in range(150):
    a = torch.rand(N, A, device="cuda:0", generator=g) * 2 - 1
    env.step(a)
    if s < 50:
        continue
    torch.cuda.synchronize()
    raw = buf.view(nwave, 16).cpu().numpy()
    d = np.diff(raw[:, :15].astype(np.float64), axis=1)
    acc += d.mean(axis=0)
    cnt += 1
acc /= cnt
tot = acc.sum()
print(f"{task} N={N}: mean cycles per wavefront (s_memtime ticks), total {tot:.0f}")
for n, v in zip(names, acc):
    print(f"  {n:28s} {v:10.0f}  {100 * v / tot:5.1f} %")
