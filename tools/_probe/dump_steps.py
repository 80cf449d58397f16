import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from robot_lab_amd.env import ManagerBasedRLEnv
task, N, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
env = ManagerBasedRLEnv(task, num_envs=N, seed=11, device="cuda:0")
obs, _ = env.reset()
rng = np.random.default_rng(3)
d = {"obs0": obs["critic"].cpu().numpy()}
for s in range(3):
    a = rng.uniform(-1, 1, (N, env.num_actions)).astype(np.float32)
    obs, rew, term, tout, _ = env.step(torch.from_numpy(a).cuda())
    r = env.scene["robot"].data
    d[f"rew{s}"] = rew.cpu().numpy(); d[f"terms{s}"] = env.reward_terms().cpu().numpy(); d[f"critic{s}"] = obs["critic"].cpu().numpy()
    d[f"policy{s}"] = obs["policy"].cpu().numpy()
    d[f"root{s}"] = r.root_state_w.cpu().numpy(); d[f"q{s}"] = r.joint_pos.cpu().numpy(); d[f"qd{s}"] = r.joint_vel.cpu().numpy()
    d[f"done{s}"] = (term | tout).cpu().numpy()
np.savez(out, **d)
