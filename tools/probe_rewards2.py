import sys, os, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from kernel_probe import run
from robot_lab_amd.scene import load_bundle
R = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
desc, _ = load_bundle(R)
names = list(desc.reward_names)
base = run(R, mutate=lambda d: setattr(d.task, "n_rewards", 0))
print(f"none {1e3*base:.1f}")
sz = ctypes.sizeof(desc.task.rewards[0])
for name in ("upward", "joint_torques_l2", "contact_forces", "feet_height_body"):
    i = names.index(name)
    for n in (1, 8, 17):
        def rep(d, i=i, n=n):
            src = ctypes.string_at(ctypes.addressof(d.task.rewards[i]), sz)
            for j in range(n):
                ctypes.memmove(ctypes.addressof(d.task.rewards[j]), src, sz)
            d.task.n_rewards = n
        t = run(R, mutate=rep)
        print(f"{name:20s} x{n:2d} +{1e3*(t-base):6.1f} us")
