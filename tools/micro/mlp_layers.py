import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from robot_lab_amd.policy import MlpPolicy
def run(dims, N=4096):
    rng = np.random.default_rng(0)
    ws = [(rng.standard_normal((dims[i+1], dims[i]))/np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims)-1)]
    bs = [np.zeros(d, dtype=np.float32) for d in dims[1:]]
    p = MlpPolicy(ws, bs, "elu", device="cuda:0")
    x = torch.randn(N, dims[0], device="cuda:0")
    for _ in range(20): p(x)
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): p(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1)/200*1e3
    fl = 2*N*sum(dims[i]*dims[i+1] for i in range(len(dims)-1))
    print(f"{dims}: {us:.1f} us, {fl/us/1e6:.1f} TFLOP/s")
    return us
a = run([512]*3); b = run([512]*5); c = run([512]*9)
print("per 512x512 layer:", (c-b)/4, "us; fixed:", b - 4*(c-b)/4)
run([256]*9); run([128]*9)
