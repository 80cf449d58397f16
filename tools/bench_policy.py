#!/usr/bin/env python
"""Fused MLP inference kernel (include/rl_policy.h) on MI355X: time per call, achieved fp32-MFMA rate, and the
same network in eager torch (rocBLAS GEMMs + elementwise kernels) beside it for context.
    python tools/bench_policy.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robot_lab_amd.policy import MlpPolicy  # noqa: E402

PEAK_F32_MFMA_TF = 157.3  # MI355X_MICROARCH.md: exact-f32 MFMA = the fp32 vector rate


def timeit(fn, reps=200):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'network':34s} {'rows':>6s} {'fused us':>9s} {'TFLOP/s':>8s} {'of f32 MFMA peak':>17s} {'torch eager us':>15s}")
for name, dims, N in (("A1 actor 45-512-256-128-12", [45, 512, 256, 128, 12], 4096), ("A1 critic 235-512-256-128-1", [235, 512, 256, 128, 1], 4096),
                      ("G1 actor 96-512-256-128-29", [96, 512, 256, 128, 29], 2048), ("A1 actor, 65536 rows", [45, 512, 256, 128, 12], 65536)):
    rng = np.random.default_rng(0)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [np.zeros(dims[i + 1], dtype=np.float32) for i in range(len(dims) - 1)]
    pol = MlpPolicy(ws, bs, "elu", device="cuda:0")
    x = torch.rand(N, dims[0], device="cuda:0") * 2 - 1
    layers = []
    for i in range(len(ws)):
        lin = torch.nn.Linear(dims[i], dims[i + 1])
        lin.weight.data, lin.bias.data = torch.tensor(ws[i]), torch.tensor(bs[i])
        layers += [lin] + ([torch.nn.ELU()] if i < len(ws) - 1 else [])
    net = torch.nn.Sequential(*layers).cuda().eval()
    with torch.inference_mode():
        t_f = timeit(lambda: pol(x))
        t_t = timeit(lambda: net(x))
    flops = 2.0 * N * sum(dims[i] * dims[i + 1] for i in range(len(ws)))
    tf = flops / (t_f * 1e-6) / 1e12
    print(f"{name:34s} {N:6d} {t_f:9.1f} {tf:8.2f} {100 * tf / PEAK_F32_MFMA_TF:16.1f}% {t_t:15.1f}")
    pol.close()
