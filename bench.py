#!/usr/bin/env python
"""bench.py - env-steps/sec of `ManagerBasedRLEnv.step()` on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python bench.py --gpus 8 --steps 1000 --warmup 100          # spawns 8 ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                   # or is launched as a rank (the reference's contract:
                                                                  # scripts/reinforcement_learning/rsl_rl/train.py:143-150, README.md:323-337)

A "step" is one `env.step(actions)` over all environments of the rank: one pass of the hot path
(action processing, 4 physics substeps, term stack, resets, observations) = one HIP kernel launch.
Workload at N=1: BASELINE.json configs[1] = Unitree-A1 Velocity-Rough, 4096 envs, random actions
~U(-1,1) (the `scripts/tools/random_agent.py:68` distribution) already resident in HBM.  Environments
shard embarrassingly: every rank owns its own 4096 envs, seed = 42 + rank (as `train.py:148`); the only
collective is one RCCL all-reduce of the packed episode-metric vector after the timed region ("weak" scaling).
`--gpus N` must equal the number of ranks that actually run: without WORLD_SIZE in the environment the script
re-executes itself under `torch.distributed.run` with N ranks; with it, a mismatch is an error.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch
(3.4 KB per env-step x 4096 envs) / mean kernel duration measured with HIP events
on the launch stream; `cpu_baseline` = the same env-step program compiled for the host (tests/emu: the source
hipcc compiles, g++ -O3 -march=native) as a CPU program - one environment per thread, one pinned thread per physical
core - on a bounded sample timed three times (rank 0, N=1 only), with the fp64 numpy oracle's figure next to it.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per env-step (Rough); Flat drops the 187 scan rays read + written
ALGO_BYTES_PER_ENV_STEP = {"A1": 857 * 4, "Go2": 3600, "Go2W": 3800, "G1": 5500}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--task", type=str, default="RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0")
    ap.add_argument("--preroll", type=int, default=300, help="untimed steps before the warm-up, after the episode clocks were randomised")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--large-batch-envs", type=int, default=65536, help="0: skip the large-batch legs (rank 0 of a 1-GPU run, after the timed region)")
    ap.add_argument("--mid-batch-envs", type=int, default=8192, help="second such leg (two sub-lanes per limb at this size); skipped with --large-batch-envs 0")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work the baseline leg may spend on the lane-program port")
    ap.add_argument("--cpu-oracle-envs", type=int, default=256)
    ap.add_argument("--cpu-oracle-steps", type=int, default=10)
    return ap.parse_args(argv)


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no rank environment: become the launcher of N ranks (one per GPU)."""
    import socket

    share = os.environ.get("RL_BENCH_SHARE_GPU") == "1"
    if not share:
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) are visible; refusing to run a {args.gpus}-GPU "
                  f"benchmark on fewer GPUs", file=sys.stderr)
            return 2
    with socket.socket() as s:  # a free rendezvous port on the loop-back interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the number of ranks must match the flag")
    # RL_BENCH_SHARE_GPU=1 (self-test of the N > 1 control flow on a 1-GPU box): every rank uses cuda:0 and the two tiny
    # collectives run over gloo on host tensors (RCCL refuses two ranks on one device).  Never set by the driver.
    share = os.environ.get("RL_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but {torch.cuda.device_count()} HIP device(s) are visible")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    coll = "cpu" if share else dev  # where the collectives' tensors live
    # RL_BENCH_FORCE_DIST=1 (self-test on a 1-GPU box): run the RCCL leg - init_process_group("nccl", device_id), the float64
    # all_gather, the MAX and SUM all-reduces on device tensors - with a world of ONE rank.  Never set by the driver.
    use_dist = world > 1 or os.environ.get("RL_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if "MASTER_PORT" not in os.environ:  # (only without a launcher: the forced single-rank self-test)
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(dev))  # RCCL on ROCm
        assert dist.get_world_size() == world

    from robot_lab_amd.env import ManagerBasedRLEnv

    N = args.num_envs
    env = ManagerBasedRLEnv(args.task, num_envs=N, seed=42 + rank, device=dev)
    A = env.num_actions
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    ring = [torch.rand(N, A, device=dev, generator=gen) * 2 - 1 for _ in range(32)]  # synthetic actions, resident in HBM
    env.reset()
    # Steady state before the driver's warm-up: a window right after reset() is a cold one (robots still falling, no resets for 1000
    # steps).  As rsl_rl's `init_at_random_ep_len` (train.py:224) the episode clocks are spread over [0, max), then `--preroll` untimed
    # steps bring contacts, commands and curricula to where a training run keeps them; every later window sees ~N/1000 time-outs per step.
    env.episode_length_buf = torch.randint(0, int(env.max_episode_length), (N,), device=dev, generator=gen)
    for i in range(args.preroll):
        env.step(ring[i % 32])

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        env.step(ring[i % 32])
    ep_before = env.episode_length_buf.clone()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        env.step(ring[i % 32])
    barrier()
    elapsed_rank = time.perf_counter() - t0
    elapsed = elapsed_rank
    per_rank = [N * args.steps / elapsed_rank]
    if use_dist:
        t = torch.tensor([elapsed_rank], device=coll, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank = [N * args.steps / float(x.item()) for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # what the timed window held (a cold 20-step window after reset looks different from steady state: robots still
    # falling, no resets): envs that were reset inside it, and how many bodies touch the ground at its end
    envs_reset = int((env.episode_length_buf < ep_before + args.steps).sum())
    env._export_stamp = -1
    env._export()
    bodies_in_contact = float((env._bufs["CONTACT_TIMERS"][: N, :, 1] > 0).sum(dim=1).float().mean())

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

    # the one collective of the path: the packed episode-metric vector, SUM all-reduced on a side stream, off the
    # timed region (robot_lab_amd/dist.py reduce_episode_log - what a --distributed training run's rank-0 log uses, too)
    from robot_lab_amd.dist import LOG_SLOT_NUM_ENVS, reduce_episode_log

    log_vec = reduce_episode_log(env).vector()  # word LOG_SLOT_NUM_ENVS: the envs behind the reduced vector

    traffic = sq = prof_src = None
    try:  # measured separately with rocprofv3 PMC passes (cannot be collected inside this process)
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tj = json.load(f)
        tr, cal = tj.get(args.task), tj.get("calibration", {})
        if tr and N == tr.get("num_envs", 4096):
            # raw counters x the factors calibrated on this kernel's own access pattern (tools/micro/stream_rows.hip: FETCH_SIZE reads 1/2)
            traffic = tr["fetch_bytes"] * cal.get("fetch_factor", 1.0) + tr["write_bytes"] * cal.get("write_factor", 1.0)
            sq = tr.get("sq")  # where a wavefront's cycles go (SQ counters, same offline pass): the kernel is issue / latency bound
            prof_src = tr.get("source", "profiles/traffic.json")
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
                   "parallelism": f"env-shard x{world}",
                   # which step kernel ran: the one specialised on this task (csrc/env_spec.h, id of tools/gen_specs.py) or the term-stack interpreter
                   "step_kernel": env.step_kernel,
                   "envs_per_wavefront": env._native.envs_per_wavefront()},
        "rccl_ranks": dist.get_world_size() if use_dist else 1, "collective_backend": ("gloo (RL_BENCH_SHARE_GPU self-test)" if share else "nccl (RCCL)") if use_dist else None,
        "per_rank_env_steps_per_s": per_rank, "envs_behind_reduced_log": float(log_vec[LOG_SLOT_NUM_ENVS]),
        "window": {"envs_reset_in_window": envs_reset, "mean_bodies_in_contact_at_end": bodies_in_contact,
                   "preroll_steps": args.preroll,
                   "note": "rank 0; episode clocks randomised over [0, 1000) and `preroll_steps` untimed steps before the warm-up: steady state"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes,
                     "kernel_only_env_steps_per_s": N / (kernel_ms * 1e-3), "wavefront_cycle_breakdown": sq,
                     # `achieved` / `kernel_ms` are measured live in this run; `traffic` and the cycle breakdown are NOT: they are the
                     # rocprofv3 PMC passes of an earlier run of the same command, read from the file named here
                     "offline": {"fields": ["traffic", "wavefront_cycle_breakdown"], "source": prof_src} if traffic is not None else None},
    }
    env.close()
    if rank == 0 and world == 1 and args.large_batch_envs > N:
        out["large_batch"] = large_batch(args.task, args.large_batch_envs, dev)
        if args.mid_batch_envs > N and args.mid_batch_envs != args.large_batch_envs:
            out["mid_batch"] = large_batch(args.task, args.mid_batch_envs, dev)
        for lb in (out["large_batch"], out.get("mid_batch")):  # the same algorithmic bytes per env-step, priced with the loop time of that leg
            if lb is not None:
                lb["achieved_GBps"] = (algo_bytes / N) * lb["envs_per_gpu"] / (lb["ms_per_step"] * 1e-3) / 1e9
                lb["roofline_frac"] = lb["achieved_GBps"] / HBM_PEAK_GBS
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.task, N, args.cpu_seconds, args.cpu_oracle_envs, args.cpu_oracle_steps)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio, which a pipe buffers until exit
        import ctypes

        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


def large_batch(task: str, n_envs: int, dev: str) -> dict:
    """Not the headline (BASELINE.json quotes 4096 envs/GPU): the same env.step() loop at a launch size that fills the chip several
    times over, where rl_env_create picks the one-lane-per-limb mapping (16 envs per wavefront, csrc/rl_env.hip envs_per_wave).
    Reported beside `value`, never instead of it."""
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    env = ManagerBasedRLEnv(task, num_envs=n_envs, seed=42, device=dev)
    A = env.num_actions
    gen = torch.Generator(device=dev).manual_seed(99)
    ring = [torch.rand(n_envs, A, device=dev, generator=gen) * 2 - 1 for _ in range(4)]
    env.reset()
    for i in range(30):
        env.step(ring[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 200
    for i in range(K):
        env.step(ring[i % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ept = int(env._native.envs_per_wavefront()) if hasattr(env._native, "envs_per_wavefront") else None
    env.close()
    return {"envs_per_gpu": n_envs, "value": n_envs * K / dt, "unit": "env-steps/s", "ms_per_step": 1e3 * dt / K, "steps": K,
            "envs_per_wavefront": ept, "note": "same task and loop as `value`, larger launch; the lane mapping is chosen from the launch size"}


def build_host_port(only: int | None = None) -> str:
    """The lane program (robot_lab_amd/csrc/env_step.h + env_terms.h, the source hipcc compiles) built for THIS box's CPU:
    g++ -O3 -march=native of tests/emu/rl_env_emu.cpp into a per-box cache (a -march=native object must not travel)."""
    import hashlib

    src = os.path.join(ROOT, "tests", "emu", "rl_env_emu.cpp")
    csrc = os.path.join(ROOT, "robot_lab_amd", "csrc")
    h = hashlib.sha1()
    for p in [src] + sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".inl"))) + [os.path.join(ROOT, "include", "rl_env.h")]:
        h.update(open(p, "rb").read())
    cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rl_env_host_port_{h.hexdigest()[:12]}_{only or 'all'}.so")
    if not os.path.isfile(cache):
        flags = [f"-DRL_EMU_ONLY={only}"] if only else []  # one lane-program instance: seconds instead of minutes of g++
        subprocess.run(["g++", "-O3", "-march=native", "-std=c++17", "-pthread", "-shared", "-fPIC", *flags, "-o", cache + ".tmp", src], check=True)
        os.replace(cache + ".tmp", cache)
    return cache


def physical_cores() -> list[int]:
    """One logical CPU per PHYSICAL core this process may run on (the first hardware thread of every sibling set): two hardware
    threads of a core share its FP units, so a second thread per core adds little for this fp32 chain - and choosing the count by a
    probe made the round-3 figure swing 1.75x between boxes."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = list(range(os.cpu_count() or 1))
    firsts, seen = [], set()
    for c in allowed:
        try:
            with open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") as f:
                sib = f.read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            firsts.append(c)
    # ... capped by the CPU time the container may use: a cgroup quota below the core count makes every thread beyond it time-share
    # (the GPU boxes of this pool show 256 logical CPUs and `cpu.max = 1600000 100000`, i.e. 16 CPUs: 16 pinned threads run at
    # 34.7 k env-steps/s each, 128 at 2.6 k each - profiles/r04b_cpu_scaling.txt; round 3's "2.5 k per core" was the quota, not the program)
    quota = cpu_quota()
    if quota is not None and quota < len(firsts):
        firsts = firsts[:max(1, int(quota))]
    return firsts


def cpu_quota():
    """CPUs' worth of time the cgroup grants (cgroup v2 cpu.max / v1 cfs quota), None when unlimited or unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(task, n_envs, budget_s, oracle_envs, oracle_steps):
    """CPU figures on this box's host cores, same task and env count as the GPU line:
    (i) `value` (kind "port"): the env-step program compiled for the host (g++ -O3 -march=native of the source hipcc compiles) as a
        CPU PROGRAM - ONE ENVIRONMENT PER THREAD: a thread walks its environments one after the other, the four limbs of an
        environment are coroutines of that thread that hand over at every cross-limb sum (a user-space switch of six registers: no
        barrier, no second core, tests/emu/rl_env_emu.cpp FiberSet), one pinned thread per PHYSICAL core (fixed, not probed), every
        thread owning whole state tiles (no cache line is written by two cores) - SURVEY.md 8(d)'s "scalar restatement with one env
        per iteration over all cores".  Timed three times (`value` = the median); `repeats` carries all three and their relative spread.
    (ii) `oracle`: the fp64 numpy oracle (oracle/env.py, single process) on a smaller sample."""
    import numpy as np

    from oracle.env import OracleEnv
    from robot_lab_amd.scene import build_world, load_bundle

    cpus = physical_cores()
    cores = len(cpus)
    rng = np.random.default_rng(0)
    desc, extra = load_bundle(task)
    D = desc.model.num_dof
    # the one-lane-per-limb instance of a 3-joint-leg quadruped (A1, Go2: key 31) when that is the task; everything else: all instances
    m = desc.model
    quad3 = m.num_trunk == 0 and m.chain_len == 3 and all(m.chain_nj[k] == 3 for k in range(4))
    lib = build_host_port(31 if quad3 else None)

    def run(steps):
        """env-steps/s with one pinned thread per physical core, a fresh process (and pool) per measurement"""
        code = ("import sys, time, numpy as np\nsys.path.insert(0, %r)\n"
                "from robot_lab_amd.capi import NativeEnv\nfrom robot_lab_amd.scene import build_world, load_bundle\n"
                "desc, extra = load_bundle(%r)\nh, to, eo = build_world(desc, extra, %d, 0)\n"
                "nat = NativeEnv(desc, h, to, eo, %d, 42, 0, %r)\nnat.reset()\n"
                "a = np.random.default_rng(0).uniform(-1, 1, (8, %d, %d)).astype(np.float32)\nnat.step(a[0].ctypes.data)\nnat.step(a[1].ctypes.data)\n"
                "t0 = time.perf_counter()\nfor s in range(%d): nat.step(a[s %% 8].ctypes.data)\nprint(time.perf_counter() - t0)\n"
                % (ROOT, task, n_envs, n_envs, lib, n_envs, D, steps))
        env = dict(os.environ, RL_EMU_TEAMS=str(cores), RL_EMU_CPUS=",".join(map(str, cpus)), RL_EMU_FIBERS="1")
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            return 0.0, 0.0
        dt = float(p.stdout.strip().splitlines()[-1])
        return n_envs * steps / dt, dt

    probe = run(3)[0]  # sizes the sample only; the thread count is fixed
    steps = int(max(5, min(4000, budget_s / 3.0 * probe / n_envs))) if probe > 0 else 5
    reps = [run(steps), run(steps), run(steps)]  # three repeats: the boxes of the pool are shared, one repeat in three is off by 10 % or more
    vals = sorted(r[0] for r in reps)
    port = vals[1]  # the median
    spread = (vals[2] - vals[0]) / max(port, 1e-9)
    # (ii) fp64 numpy oracle (one process; numpy's own threading aside)
    h, to, eo = build_world(desc, extra, oracle_envs, 0)
    ora = OracleEnv(desc, h, to, oracle_envs, 42, eo)
    ora.reset()
    oa = rng.uniform(-1, 1, (oracle_steps + 1, oracle_envs, D))
    ora.step(oa[0])
    t0 = time.perf_counter()
    for s in range(oracle_steps):
        ora.step(oa[s + 1])
    odt = time.perf_counter() - t0
    return {"value": port, "unit": "env-steps/s", "cores": cores, "kind": "port", "per_core": port / max(cores, 1),
            "sample": f"{n_envs} envs x {steps} steps of the same task, timed three times ({' + '.join('%.1f' % r[1] for r in reps)} s; value = the median): the env-step program "
                      f"compiled for the host (g++ -O3 -march=native), one environment per thread - {cores} threads pinned one per physical "
                      f"core (as many as the container's CPU quota grants), each walking its own state tiles, the four limbs of an environment as "
                      f"coroutines of that thread",
            "repeats": {"values": [r[0] for r in reps], "relative_spread": spread},
            "logical_cpus_available": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count(),
            "cgroup_cpu_quota": cpu_quota(),
            "oracle": {"value": oracle_envs * oracle_steps / odt, "unit": "env-steps/s", "cores": 1, "kind": "oracle",
                       "sample": f"{oracle_envs} envs x {oracle_steps} steps, fp64 numpy oracle (oracle/env.py), single process"},
            "reference_terms_cpu": reference_terms_offline(task)}


def reference_terms_offline(task):
    """The part of the REFERENCE that runs without IsaacLab - its own VEL/mdp reward + observation term functions on torch CPU, term stack
    ONLY (no physics, sensors, managers) - timed in the build container by tools/time_reference_terms.py and committed: /root/reference does
    not exist on the GPU box, so this figure is quoted from the file, never measured here, and it is an upper bound of a CPU reference path."""
    import json

    path = os.path.join(ROOT, "profiles", "r06_reference_terms_cpu.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    if d.get("task") != task:
        return None
    return {"value": d["value"], "unit": d["unit"], "cores": d["threads"], "kind": "reference (term stack only)", "measured": "offline: " + d["host"],
            "sample": f"{d['num_envs']} envs, {d['calls']} calls of the {d['reward_terms']} reward + {d['observation_terms']} observation term functions in {d['seconds']:.1f} s "
                      f"({d['ms_per_term_stack_call']:.2f} ms per call), torch {d['torch']} CPU fp32", "source": "profiles/r06_reference_terms_cpu.json"}


if __name__ == "__main__":
    main()
