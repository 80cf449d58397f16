import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from robot_lab_amd.policy import MlpPolicy
dims = [512] * 9
rng = np.random.default_rng(0)
ws = [(rng.standard_normal((dims[i+1], dims[i]))/np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims)-1)]
p = MlpPolicy(ws, [np.zeros(d, dtype=np.float32) for d in dims[1:]], "elu", device="cuda:0")
x = torch.randn(4096, 512, device="cuda:0")
for _ in range(30): p(x)
torch.cuda.synchronize()
