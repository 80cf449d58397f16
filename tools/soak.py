#!/usr/bin/env python
"""Long random/zero-action rollout on the GPU: finiteness, episode statistics, curriculum sanity.
    python tools/soak.py [task] [steps] [random|zero] [num_envs]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from robot_lab_amd.env import ManagerBasedRLEnv

task = sys.argv[1] if len(sys.argv) > 1 else "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
mode = sys.argv[3] if len(sys.argv) > 3 else "random"
N = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
env = ManagerBasedRLEnv(task, num_envs=N, seed=1, device="cuda:0")
env.reset()
env.episode_length_buf = torch.randint(0, env.max_episode_length, (N,))
tot_done = 0
rsum = 0.0
for s in range(1, steps + 1):
    a = torch.zeros(N, env.num_actions, device="cuda") if mode == "zero" else torch.rand(N, env.num_actions, device="cuda") * 2 - 1
    obs, rew, term, tout, extras = env.step(a)
    tot_done += int((term | tout).sum())
    rsum += float(rew.mean())
    if s % 500 == 0 or s == steps:
        d = env.scene["robot"].data
        ok = bool(torch.isfinite(obs["critic"]).all() and torch.isfinite(d.root_state_w).all() and torch.isfinite(d.joint_vel).all())
        rel_z = (d.root_pos_w[:, 2] - env.scene.env_origins[:, 2])
        up = (obs["critic"][:, 8] < -0.7).float().mean()
        print(f"step {s}: finite={ok} dones={tot_done} mean_rew={rsum / s:+.4f} rel_z mean={float(rel_z.mean()):.3f} min={float(rel_z.min()):.3f} max={float(rel_z.max()):.3f} "
              f"upright={float(up):.2f} |qd|max={float(d.joint_vel.abs().max()):.1f} |v|max={float(d.root_lin_vel_w.abs().max()):.1f} "
              f"levels mean={float(env.terrain_levels.float().mean()):.2f} max={int(env.terrain_levels.max())}", flush=True)
        assert ok
env.close()
