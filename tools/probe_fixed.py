import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from kernel_probe import run
R = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
def m(**kw):
    def f(d):
        t = d.task
        if kw.get("norew"): t.n_rewards = 0
        if kw.get("nopol"): t.n_policy = 0
        if kw.get("nocrit"): t.n_critic = 0
        if kw.get("noscan"): t.n_critic = t.n_critic - 1
        if kw.get("nopush"): t.ev_push = 0
        if kw.get("d0"): d.sim.decimation = 0
    return f
for name, kw in [("full", {}), ("norew", dict(norew=1)), ("norew nopol", dict(norew=1, nopol=1)), ("norew nopol noscan", dict(norew=1, nopol=1, noscan=1)),
                 ("norew nopol nocrit", dict(norew=1, nopol=1, nocrit=1)), ("norew nopol nocrit d0", dict(norew=1, nopol=1, nocrit=1, d0=1)),
                 ("d0 only", dict(d0=1))]:
    print(f"{name:26s} {1e3*run(R, mutate=m(**kw)):7.1f} us", flush=True)
