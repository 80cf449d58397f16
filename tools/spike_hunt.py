import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from robot_lab_amd.env import ManagerBasedRLEnv
task = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
N = 4096
env = ManagerBasedRLEnv(task, num_envs=N, seed=1, device="cuda:0")
env.reset()
env.episode_length_buf = torch.randint(0, env.max_episode_length, (N,))
prev = None
types = None
hits = 0
for s in range(1, 3001):
    a = torch.rand(N, 12, device="cuda") * 2 - 1
    obs, rew, term, tout, _ = env.step(a)
    d = env.scene["robot"].data
    rs = d.root_state_w.clone()
    v = rs[:, 7:10].norm(dim=1)
    m = int(v.argmax())
    if float(v[m]) > 18 and hits < 8:
        hits += 1
        o = env.scene.env_origins[m]
        print(f"step {s} env {m} |v|={float(v[m]):.1f} v={rs[m,7:10].tolist()} w={rs[m,10:13].tolist()} rel={(rs[m,:3]-o).tolist()} level={int(env.terrain_levels[m])} type_col={m*20//N} ep={int(env.episode_length_buf[m])}")
        if prev is not None:
            print(f"    prev |v|={float(prev[m,7:10].norm()):.2f} rel={(prev[m,:3]-o).tolist()} w={prev[m,10:13].tolist()} qd_prev_max={float(prevqd[m].abs().max()):.1f}")
    prev = rs
    prevqd = d.joint_vel.clone()
env.close()
