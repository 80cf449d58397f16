"""`-m gpu`: the rank body of `bench.py --gpus N` executed on the one GPU a test box has.

The reference's multi-GPU contract is one process per GPU under `torch.distributed.run` (scripts/reinforcement_learning/rsl_rl/
train.py:143-150, README.md:323-337); `bench.py` follows it.  A 1-GPU box cannot hold two RCCL ranks (RCCL refuses two ranks on
one device), so the N > 1 control flow and the RCCL calls are covered in two halves:

* `RL_BENCH_SHARE_GPU=1 --gpus 2`: the launcher re-executes two ranks under torch.distributed.run, both on cuda:0, every line of
  the rank body runs (per-rank seeds, barrier + max-over-ranks timing, the all_gather of the per-rank times, the reduced
  episode-metric vector) with the two tiny collectives on gloo;
* `RL_BENCH_FORCE_DIST=1 --gpus 1`: the same body with `init_process_group("nccl", device_id=...)` and the collectives on device
  tensors over RCCL, world size 1.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *flags):
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", *flags],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_share_the_gpu():
    out = _bench({"RL_BENCH_SHARE_GPU": "1"}, "--gpus", "2")
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    assert len(out["per_rank_env_steps_per_s"]) == 2 and all(v > 0 for v in out["per_rank_env_steps_per_s"])
    assert out["envs_behind_reduced_log"] == 2 * 4096  # the SUM all-reduce saw both ranks
    assert out["config"]["parallelism"] == "env-shard x2" and out["scaling"] == "weak"
    # whole-job value = all ranks' env-steps over the slowest rank's time
    assert out["value"] <= sum(out["per_rank_env_steps_per_s"]) * 1.0001
    assert "gloo" in out["collective_backend"]


def test_rccl_leg_with_one_rank():
    out = _bench({"RL_BENCH_FORCE_DIST": "1"}, "--gpus", "1")
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["collective_backend"] == "nccl (RCCL)"
    assert out["envs_behind_reduced_log"] == 4096
    assert out["roofline"]["kernel_ms"] > 0
