#!/usr/bin/env python
"""Actor + critic of a rollout step in one launch (rl_mlp_forward_pair): time per call and share of the fp32-MFMA peak.
Variants come from the environment (read once per process): RL_MLP_WARM=0|1, RL_MLP_PAIR_SPLIT=0|1, RL_MLP_PAIR_RT=1|2.
    python tools/bench_pair.py [rows] [obs_dim] [critic_dim] [act_dim]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robot_lab_amd.policy import MlpPolicy  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
od, cd, A = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 45), (3, 235), (4, 12)))
PEAK = 157.3
rng = np.random.default_rng(0)


def net(dims):
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [0.1 * rng.standard_normal(d).astype(np.float32) for d in dims[1:]]
    return MlpPolicy(ws, bs, "elu", device="cuda:0"), ws, bs


da, dc = [od, 512, 256, 128, A], [cd, 512, 256, 128, 1]
(actor, wa, ba), (critic, wc, bc) = net(da), net(dc)
xa, xc = torch.rand(N, od, device="cuda:0") * 2 - 1, torch.rand(N, cd, device="cuda:0") * 2 - 1


def ref(x, ws, bs):
    h = x.double()
    for i, (w, b) in enumerate(zip(ws, bs)):
        h = h @ torch.tensor(w, device="cuda:0").double().T + torch.tensor(b, device="cuda:0").double()
        if i < len(ws) - 1:
            h = torch.nn.functional.elu(h)
    return h


with torch.inference_mode():
    ya, yc = actor.forward_pair(xa, critic, xc)
    ea, ec = float((ya.double() - ref(xa, wa, ba)).abs().max()), float((yc.double() - ref(xc, wc, bc)).abs().max())
    for _ in range(30):
        actor.forward_pair(xa, critic, xc)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if os.environ.get("RL_COLD") == "1":  # as inside the collection loop: other kernels ran since the last call (L2 no longer holds the weights)
        big0, big1 = torch.empty(64 << 20, device="cuda:0"), torch.empty(64 << 20, device="cuda:0")
        tot = 0.0
        for _ in range(100):
            big1.copy_(big0)
            e0.record()
            actor.forward_pair(xa, critic, xc)
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        us = tot / 100 * 1e3
    else:
        e0.record()
        for _ in range(300):
            actor.forward_pair(xa, critic, xc)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 300 * 1e3
flops = 2.0 * N * (sum(da[i] * da[i + 1] for i in range(4)) + sum(dc[i] * dc[i + 1] for i in range(4)))
tf = flops / us / 1e6
cfg = " ".join(f"{k}={os.environ[k]}" for k in ("RL_MLP_PAIR_MODE", "RL_MLP_FUSED_WAVES", "RL_MLP_PAIR_RT", "RL_COLD") if k in os.environ)
print(f"pair {od}/{cd}->{A} rows {N} [{cfg or 'defaults'}]: {us:.1f} us  {tf:.1f} TFLOP/s = {100 * tf / PEAK:.1f}% of fp32 MFMA peak; max |err| actor {ea:.2e} critic {ec:.2e}")
