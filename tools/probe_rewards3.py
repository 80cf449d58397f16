import sys, os, ctypes
R0 = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R0); sys.path.insert(0, os.path.join(R0, "tools"))
from kernel_probe import run
from robot_lab_amd.scene import load_bundle
R = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
desc, _ = load_bundle(R)
names = list(desc.reward_names)
sz = ctypes.sizeof(desc.task.rewards[0])
groups = {
    "vector(6)": ["lin_vel_z_l2", "ang_vel_xy_l2", "track_lin_vel_xy_exp", "track_ang_vel_z_exp", "upward"],
    "joint(7)": ["joint_torques_l2", "joint_acc_l2", "joint_pos_limits", "joint_power", "stand_still", "joint_pos_penalty", "action_rate_l2"],
    "body(4)": ["undesired_contacts", "contact_forces", "feet_contact_without_cmd", "feet_height_body"],
    "mirror": ["joint_mirror"],
}
def keep(sel):
    def f(d):
        blobs = [ctypes.string_at(ctypes.addressof(d.task.rewards[names.index(n)]), sz) for n in sel]
        for j, b in enumerate(blobs):
            ctypes.memmove(ctypes.addressof(d.task.rewards[j]), b, sz)
        d.task.n_rewards = len(blobs)
    return f
base = run(R, mutate=keep([]))
print(f"none {1e3*base:.1f}")
acc = []
for g, sel in groups.items():
    t = run(R, mutate=keep(sel))
    print(f"only {g:10s} +{1e3*(t-base):5.1f} us")
t = run(R)
print(f"all        +{1e3*(t-base):5.1f} us")
