"""world_size-2 check of the multi-GPU design on CPU with gloo: envs shard with NO data-path
collective - rank r simulates its own envs with seed 42+r - and the only exchange is one all-reduce (SUM) of
the packed episode-metric vector.  The env behind the C-ABI is the CPU lane emulator here (same lane-program
source as the HIP kernel); on the MI355X node bench.py does the same with RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TASK = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
N, STEPS = 16, 4


def _shard(rank, emu_lib):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import host_view
    from robot_lab_amd.capi import NativeEnv
    from robot_lab_amd.scene import build_world, load_bundle

    desc, extra = load_bundle(TASK)
    h, to, eo = build_world(desc, extra, N, 0)
    nat = NativeEnv(desc, h, to, eo, N, 42 + rank, 0, emu_lib)
    nat.reset()
    ep = np.zeros(N, dtype=np.int64)
    ep[::2] = nat.max_episode_length - 2  # make half of the envs time out inside the window
    host_view(nat, "EPISODE_LENGTH")[:] = ep
    rng = np.random.default_rng(100 + rank)
    log = np.zeros(64, dtype=np.float32)
    rew = 0.0
    for _ in range(STEPS):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        nat.step(a.ctypes.data)
        log += host_view(nat, "LOG")[nat.log_slot()].sum(axis=0)  # this step's own accumulators, its partial rows summed (rl_env_read_log would repeat the last reset's)
        rew += float(host_view(nat, "REWARD").sum())
    nat.close()
    return log, rew


def _worker(rank, world, port, emu_lib, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log, rew = _shard(rank, emu_lib)
    t = torch.from_numpy(log.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)  # the one collective of the path
    q.put((rank, t.numpy().copy(), log, rew))
    dist.barrier()
    dist.destroy_process_group()


def test_env_shards_and_metric_allreduce(emu_lib):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, 29517, emu_lib, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single-process emulation of the two logical shards
    ref = [_shard(r, emu_lib) for r in range(world)]
    total = ref[0][0] + ref[1][0]
    for rank, reduced, local, rew in got:
        np.testing.assert_array_equal(local, ref[rank][0])       # a shard is bit-identical to the same-seed single run
        assert rew == ref[rank][1]
        np.testing.assert_allclose(reduced, total, rtol=1e-6)    # all-reduce == sum over shards
    assert total[0] == world * N // 2                            # every forced time-out was counted exactly once
    assert not np.array_equal(ref[0][0], ref[1][0])              # different seeds -> different shards


# ------------------------------------------------------------------------------------------------
# bench.py --gpus N: the flag decides how many ranks run (train.py:143-150 launch contract)
# ------------------------------------------------------------------------------------------------
def _bench(args, env_extra=None, timeout=300):
    import subprocess

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason="box has >= 2 GPUs: the launch would succeed")
def test_bench_refuses_more_gpus_than_the_box_has():
    r = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 2 and "--gpus 2" in r.stderr and "HIP device" in r.stderr
    assert r.stdout.strip() == ""  # no JSON line that could be mistaken for a 2-GPU result


def test_bench_rejects_a_rank_count_that_differs_from_the_flag():
    r = _bench(["--gpus", "4", "--steps", "2", "--warmup", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_bench_spawns_the_ranks_itself(monkeypatch):
    """`python bench.py --gpus 2` with no rank environment re-executes itself under torch.distributed.run with 2 ranks: checked
    through the launcher's command line (the ranks themselves need GPUs)."""
    import bench

    seen = {}
    monkeypatch.setenv("RL_BENCH_SHARE_GPU", "1")  # skips the device-count check of the launcher
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    assert bench.spawn_ranks(bench.parse_args(["--gpus", "2", "--steps", "3"])) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "2", "--steps", "3"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


# ------------------------------------------------------------------------------------------------
# robot_lab_amd.dist.reduce_episode_log: the collective itself, as product code (SURVEY.md 8(e) "Collective")
# ------------------------------------------------------------------------------------------------
class _EmuEnvView:
    """What reduce_episode_log reads of a `ManagerBasedRLEnv`, over a NativeEnv of the CPU lane emulator (host pointers)."""

    def __init__(self, nat, desc, N):
        from helpers import host_view

        self._native, self.desc, self.num_envs, self.unwrapped = nat, desc, N, self
        self._bufs = {"LOG": torch.from_numpy(host_view(nat, "LOG"))}
        self.max_episode_length_s = float(desc.task.episode_length_s)
        self._levels = host_view(nat, "TERRAIN_LEVEL")

    @property
    def common_step_counter(self):
        return self._native.step_count

    @property
    def terrain_levels(self):
        return torch.from_numpy(self._levels[: self.num_envs].copy())


def _log_shard(rank, emu_lib):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import host_view
    from robot_lab_amd.capi import NativeEnv
    from robot_lab_amd.scene import build_world, load_bundle

    desc, extra = load_bundle(TASK)
    h, to, eo = build_world(desc, extra, N, 0)
    nat = NativeEnv(desc, h, to, eo, N, 42 + rank, 0, emu_lib)
    nat.reset()
    ep = np.zeros(N, dtype=np.int64)
    ep[: 3 + 2 * rank] = nat.max_episode_length - 1  # 3 envs of rank 0, 5 of rank 1 time out on the first step
    ep[10:12] = nat.max_episode_length - 3           # ... and two more of each rank on the third
    host_view(nat, "EPISODE_LENGTH")[:] = ep
    rng = np.random.default_rng(7 + rank)
    for _ in range(4):  # the second and the fourth step reset nobody: the log a caller sees after them is still their predecessor's
        nat.step(rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32).ctypes.data)
    return _EmuEnvView(nat, desc, N)


def _log_worker(rank, world, port, emu_lib, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from robot_lab_amd.dist import pack_episode_log, reduce_episode_log

    env = _log_shard(rank, emu_lib)
    assert float(pack_episode_log(env)[0]) == 2.0  # one step: the log of the most recent step that reset an env (the third)
    local = pack_episode_log(env, 4).numpy().copy()  # the window of a rollout: every episode that ended inside it, once
    fut = reduce_episode_log(env, steps=4)
    res = {k: float(v) for k, v in fut.result().items()}
    q.put((rank, local, fut.vector().numpy().copy(), res))
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_episode_log_gives_the_jobs_means(emu_lib):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_log_worker, args=(r, world, 29519, emu_lib, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    total = got[0][1] + got[1][1]
    assert got[0][1][0] == 5 and got[1][1][0] == 7 and got[0][1][7] == N  # each rank's own vector: its resets (3 + 2, 5 + 2), its env count
    assert got[0][1][62] == 1.0 and total[62] == world                     # a 1 per rank: the reduced vector knows the world size
    for rank, local, reduced, res in got:
        np.testing.assert_allclose(reduced, total, rtol=1e-6)          # every rank holds the same reduced vector = the sum over the shards
        assert res["episodes"] == 12 and res["num_envs"] == world * N
        assert res["Episode_Termination/time_out"] == 12
        # a mean over the job's 12 ended episodes, not over one rank's: (sum_0 + sum_1) / 12 / 20 s
        want = total[8] / 12.0 / 20.0
        first = [k for k in res if k.startswith("Episode_Reward/")][0]
        assert abs(res[first] - want) <= 1e-6 * max(1.0, abs(want))
        assert abs(res["Curriculum/terrain_levels"] - total[6] / (world * N)) < 1e-6
    # without a process group (a single-GPU run) the same call is the local log, no collective
    from robot_lab_amd.dist import reduce_episode_log

    env = _log_shard(0, emu_lib)
    r = reduce_episode_log(env).result()  # (first call of this env: the one-step form)
    assert float(r["episodes"]) == 2 and float(r["num_envs"]) == N
    r = reduce_episode_log(env, steps=4).result()
    assert float(r["episodes"]) == 5
