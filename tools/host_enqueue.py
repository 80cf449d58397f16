#!/usr/bin/env python
"""Host-side readiness for the 8-GPU run (no node needed, no scaling claimed): what ENQUEUEING one env.step() costs a
rank's host thread while `--ranks` ranks are alive on the box's CPU quota (the gpurun boxes: cgroup quota of 16 CPUs).  Every rank is its
own process with its own env (seed 42 + rank, 4096 envs) on the ONE GPU of the box (RL_SHARE_GPU=1: a self-test aid); all ranks start a
round together (file barrier), enqueue `--burst` steps WITHOUT a device sync and stop the clock before synchronising.  The GPU is shared
by the ranks, so kernel throughput here means nothing - only the enqueue loop is timed.  An 8-GPU run is host-bound if this figure is
above the kernel time (36 us at 4096 A1 envs); the burst is short enough not to fill the HIP queue (a full queue would make enqueue =
kernel time and hide the host cost: the per-burst figures are printed so that a saturating burst is visible).

    python tools/host_enqueue.py --ranks 8 [--num-envs 4096] [--burst 64] [--rounds 12] [--out profiles/r06_host_enqueue_8ranks.json]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def barrier(d, name, rank, n, timeout=300.0):
    open(os.path.join(d, f"{name}.{rank}"), "w").close()
    t0 = time.time()
    while sum(os.path.exists(os.path.join(d, f"{name}.{r}")) for r in range(n)) < n:
        if time.time() - t0 > timeout:
            raise SystemExit(f"rank {rank}: barrier {name} timed out")
        time.sleep(0.002)


def worker(a):
    sys.path.insert(0, ROOT)
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    os.environ["RL_SHARE_GPU"] = "1"
    env = ManagerBasedRLEnv(a.task, num_envs=a.num_envs, seed=42 + a.worker, device="cuda:0")
    env.log_episodes = a.logs
    env.reset()
    g = torch.Generator(device="cuda:0").manual_seed(a.worker)
    acts = [torch.rand(a.num_envs, env.num_actions, device="cuda:0", generator=g) * 2 - 1 for _ in range(8)]
    with torch.inference_mode():
        for s in range(50):
            env.step(acts[s % 8])
        torch.cuda.synchronize()
        per = []
        for r in range(a.rounds):
            barrier(a.dir, f"round{r}", a.worker, a.ranks)
            t0 = time.perf_counter()
            for s in range(a.burst):
                env.step(acts[s % 8])
            dt = time.perf_counter() - t0
            torch.cuda.synchronize()
            per.append(1e6 * dt / a.burst)
    per = sorted(per[2:])  # (the first two rounds warm the caches of the launch path)
    print("ENQUEUE " + json.dumps(dict(rank=a.worker, step_kernel=env.step_kernel, us_per_step_median=per[len(per) // 2], us_per_step_min=per[0],
                                       us_per_step_max=per[-1])), flush=True)
    env.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--task", default="RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--burst", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--logs", type=int, default=1, help="extras['log'] views per step as rsl_rl sees them (0: off)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--worker", type=int, default=-1)
    ap.add_argument("--dir", default=None)
    a = ap.parse_args()
    if a.worker >= 0:
        return worker(a)
    with tempfile.TemporaryDirectory(prefix="rl_enqueue_") as d:
        env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), "--dir", d, "--ranks", str(a.ranks), "--task", a.task,
                                   "--num-envs", str(a.num_envs), "--burst", str(a.burst), "--rounds", str(a.rounds), "--logs", str(a.logs)],
                                  env=env, stdout=subprocess.PIPE, text=True) for r in range(a.ranks)]
        rows = []
        for p in procs:
            out, _ = p.communicate(timeout=900)
            rows += [json.loads(l[len("ENQUEUE "):]) for l in out.splitlines() if l.startswith("ENQUEUE ")]
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    res = dict(what="host time to ENQUEUE one env.step() (Python binding + ctypes + hipLaunchKernel), no device sync inside the timed burst; all ranks alive and bursting together",
               task=a.task, num_envs=a.num_envs, ranks=a.ranks, burst=a.burst, rounds=a.rounds, extras_log_views=bool(a.logs),
               logical_cpus=len(os.sched_getaffinity(0)), cgroup_cpu_quota=quota, per_rank=sorted(rows, key=lambda r: r["rank"]),
               worst_rank_median_us=max(r["us_per_step_median"] for r in rows) if rows else None,
               note="the ranks share ONE GPU here (RL_SHARE_GPU=1), so nothing about kernel throughput or scaling follows from this run")
    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
