import sys, os, copy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from kernel_probe import run
from robot_lab_amd.scene import load_bundle
R = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
desc, _ = load_bundle(R)
names = list(desc.reward_names)
base = run(R, mutate=lambda d: setattr(d.task, "n_rewards", 0))
print(f"none {1e3*base:.1f}")
import ctypes
for i, n in enumerate(names):
    def only(d, i=i):
        ctypes.memmove(ctypes.addressof(d.task.rewards[0]), ctypes.addressof(d.task.rewards[i]), ctypes.sizeof(d.task.rewards[0]))
        d.task.n_rewards = 1
    t = run(R, mutate=only)
    print(f"{n:28s} +{1e3*(t-base):6.1f} us")
