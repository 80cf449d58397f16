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
        log += nat.read_log()
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
