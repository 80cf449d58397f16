#!/usr/bin/env python
"""bench.py - env-steps/sec of `ManagerBasedRLEnv.step()` on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one `env.step(actions)` over all environments of the rank: one pass of the hot path
(action processing, 4 physics substeps, term stack, resets, observations) = one HIP kernel launch.
Workload at N=1: BASELINE.json configs[1] = Unitree-A1 Velocity-Rough, 4096 envs, random actions
~U(-1,1) (the `scripts/tools/random_agent.py:68` distribution) already resident in HBM.  Environments
shard embarrassingly: every rank owns its own 4096 envs, seed = 42 + rank (as
`scripts/reinforcement_learning/rsl_rl/train.py:148`); the only collective is one RCCL all-reduce of
the packed episode-metric vector after the timed region ("weak" scaling).

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch
(3.4 KB per env-step x 4096 envs) / mean kernel duration measured with HIP events
on the launch stream; `cpu_baseline` = the fp64 numpy oracle timed on a bounded sample on this
box's host cores (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per env-step (Rough); Flat drops the 187 scan rays read + written
ALGO_BYTES_PER_ENV_STEP = {"A1": 857 * 4, "Go2": 3600, "Go2W": 3800, "G1": 5500}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--task", type=str, default="RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=512)
    ap.add_argument("--cpu-steps", type=int, default=30)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # RL_BENCH_SHARE_GPU=1 (self-test of the N > 1 control flow on a 1-GPU box): every rank uses cuda:0 and the two tiny
    # collectives run over gloo on host tensors (RCCL refuses two ranks on one device).  Never set by the driver.
    share = os.environ.get("RL_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    coll = "cpu" if share else dev  # where the collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(dev))  # RCCL on ROCm

    from robot_lab_amd.env import ManagerBasedRLEnv

    N = args.num_envs
    env = ManagerBasedRLEnv(args.task, num_envs=N, seed=42 + rank, device=dev)
    A = env.num_actions
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    ring = [torch.rand(N, A, device=dev, generator=gen) * 2 - 1 for _ in range(32)]  # synthetic actions, resident in HBM
    env.reset()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        env.step(ring[i % 32])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        env.step(ring[i % 32])
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=coll, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # kernel duration with HIP events on the launch stream (no Python-side extras in the loop)
    native, stream = env._native, env._stream()
    ptrs = [r.data_ptr() for r in ring]
    for i in range(20):
        native.step(ptrs[i % 32], stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    KREP = max(50, min(args.steps, 500))
    e0.record()
    for i in range(KREP):
        native.step(ptrs[i % 32], stream)
    e1.record()
    torch.cuda.synchronize()
    kernel_ms = e0.elapsed_time(e1) / KREP

    # the one collective of the path: packed episode-metric vector, off the timed region
    log_vec = env._bufs["LOG"][native.log_slot()].clone().to(coll)  # the last step's ring slot
    if world > 1:
        dist.all_reduce(log_vec, op=dist.ReduceOp.SUM)

    traffic = sq = None
    try:  # measured separately with rocprofv3 PMC passes (cannot be collected inside this process)
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tr = json.load(f).get(args.task)
        if tr and N == tr.get("num_envs", 4096):
            traffic = tr["fetch_bytes"] + tr["write_bytes"]
            sq = tr.get("sq")  # where a wavefront's cycles go (SQ counters, same offline pass): the kernel is issue / latency bound
    except (OSError, ValueError, KeyError):
        pass
    value = world * N * args.steps / elapsed
    robot = args.task.replace("RobotLab-Isaac-Velocity-", "").replace("-v0", "").split("-", 1)[1].replace("Unitree-", "")
    per_step = ALGO_BYTES_PER_ENV_STEP.get(robot, ALGO_BYTES_PER_ENV_STEP["Go2W" if A == 16 else "A1"])  # other quadrupeds: as their Unitree twin
    algo_bytes = (per_step - (2 * 187 * 4 if "Flat" in args.task else 0)) * N
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    out = {
        "metric": f"env-steps/sec (whole node) at {N} envs/GPU, {robot} Velocity-{'Flat' if 'Flat' in args.task else 'Rough'}",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.task}, {N} envs/GPU, random actions U(-1,1), seed 42+rank", "envs_per_gpu": N,
                   "parallelism": f"env-shard x{world}"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes,
                     "kernel_only_env_steps_per_s": N / (kernel_ms * 1e-3), "wavefront_cycle_breakdown": sq},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.task, args.cpu_envs, args.cpu_steps)
    env.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(task, n_envs, steps):
    """fp64 numpy oracle (the CPU restatement of the same step) on a bounded sample of the same workload."""
    import numpy as np

    from oracle.env import OracleEnv
    from robot_lab_amd.scene import build_world, load_bundle

    desc, extra = load_bundle(task)
    h, to, eo = build_world(desc, extra, n_envs, 0)
    ora = OracleEnv(desc, h, to, n_envs, 42, eo)
    ora.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (steps + 2, n_envs, desc.model.num_dof))
    ora.step(acts[0])
    t0 = time.perf_counter()
    for s in range(steps):
        ora.step(acts[s + 1])
    dt = time.perf_counter() - t0
    return {"value": n_envs * steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n_envs} envs x {steps} steps of the same task, fp64 numpy oracle (oracle/env.py), single process",
            "host_cores_available": os.cpu_count()}


if __name__ == "__main__":
    main()
