"""Side-by-side debugging of two builds of the env library: read_state() after reset() and after one step() with fixed actions, to
an .npz.   RL_ENV_LIB=<variant>.so python tools/dump_state.py <task> <num_envs> <out.npz>   (how the __launch_bounds__(64) miscompile
was located: profiles/r02_launch_bounds64_miscompile.txt)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from robot_lab_amd.env import ManagerBasedRLEnv
task, N, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
env = ManagerBasedRLEnv(task, num_envs=N, seed=11, device="cuda:0")
env.reset()
d = {"r_" + k: np.asarray(v) for k, v in env.read_state().items() if hasattr(v, "shape") or isinstance(v, (int, float))}
rng = np.random.default_rng(3)
a = rng.uniform(-1, 1, (N, env.num_actions)).astype(np.float32)
env.step(torch.from_numpy(a).cuda())
d.update({"s_" + k: np.asarray(v) for k, v in env.read_state().items() if hasattr(v, "shape") or isinstance(v, (int, float))})
np.savez(out, **d)
