"""`-m gpu`: two ranks of the reference's `train.py --distributed` body, end to end on the GPU.

`python -m torch.distributed.run --nproc_per_node=2 tests/train_body_rank.py --distributed` - the README's launch line
(`/root/reference/README.md:323-337`) around the body of `train.py:118-224` (the file itself needs /root/reference, which the GPU box
does not have: it runs, two ranks, in `tests/test_distributed_train.py`).  A box with one GPU holds both ranks under `RL_SHARE_GPU=1`
(logical cuda:1 lives on cuda:0, collectives over gloo: RCCL refuses two ranks on one device); with two or more GPUs the same test runs
the RCCL path unchanged."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(nproc, out_dir, args, share):
    env = dict(os.environ, RL_TEST_OUT=str(out_dir), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RL_SHARE_GPU"):
        env.pop(k, None)
    if share:
        env["RL_SHARE_GPU"] = "1"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc_per_node={nproc}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "train_body_rank.py"), *args]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return [json.load(open(os.path.join(str(out_dir), f"rank{r}.json"))) for r in range(nproc)], p


def test_two_ranks_train_one_model(tmp_path):
    import torch

    share = torch.cuda.device_count() < 2
    recs, p = _launch(2, tmp_path, ["--distributed", "--num_envs", "256", "--max_iterations", "3", "--headless"], share)
    for r, rec in enumerate(recs):
        assert rec["world"] == 2 and rec["launcher_local_rank"] == r and rec["launcher_global_rank"] == r
        assert rec["sim_device"] == f"cuda:{r}" and rec["agent_device"] == f"cuda:{r}" and rec["env_seed"] == 42 + r  # train.py:143-150
        assert rec["env_device"] == ("cuda:0" if share else f"cuda:{r}")
        assert rec["backend"] == ("gloo" if share else "nccl") and rec["iterations"] == 3 and rec["finite"]
    a, b = recs
    assert a["param_sha"] == b["param_sha"], "the two learners drifted apart: the gradient all-reduce is not tying them together"
    assert a["learning_rate"] == b["learning_rate"]
    assert a["reward_sha"] != b["reward_sha"]  # ... while every rank simulated its own environments (seed 42 + rank)
    assert a["episode_log_envs"] == 2 * 256 and b["episode_log_envs"] is None  # rank 0's log stands for both ranks' envs
    # rank 0 alone logs and checkpoints
    assert os.path.isfile(os.path.join(str(tmp_path), "logs", "model_3.pt"))
    assert p.stdout.count("[rsl_rl stand-in] iteration 3/3") == 1
    d = torch.load(os.path.join(str(tmp_path), "logs", "model_3.pt"), map_location="cpu", weights_only=False)
    assert d["iter"] == 3


def test_eight_ranks_on_one_node(tmp_path):
    """The launch line the reference documents for a node (README.md:323-337: --nproc_per_node=8): eight ranks - on a box with fewer GPUs they
    share what there is (RL_SHARE_GPU=1) -, eight distinct seeds and env sets, ONE model, ONE log whose episode statistics stand for all
    8 x 128 environments (the episode-metric all-reduce of SURVEY.md 8(e))."""
    import torch

    share = torch.cuda.device_count() < 8
    recs, p = _launch(8, tmp_path, ["--distributed", "--num_envs", "128", "--max_iterations", "1", "--headless"], share)
    assert sorted(r["env_seed"] for r in recs) == [42 + r for r in range(8)]
    assert len({r["param_sha"] for r in recs}) == 1 and len({r["reward_sha"] for r in recs}) == 8
    assert all(r["world"] == 8 and r["iterations"] == 1 and r["finite"] for r in recs)
    assert recs[0]["episode_log_envs"] == 8 * 128 and all(r["episode_log_envs"] is None for r in recs[1:])
    assert p.stdout.count("[rsl_rl stand-in] iteration 1/1") == 1 and "episode log over 1024 envs" in p.stdout
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "train_distributed_8ranks.json"), "w") as f:
            json.dump(dict(ranks=recs, rank0_log=[l for l in p.stdout.splitlines() if "rsl_rl stand-in" in l]), f, indent=1)


def test_a_second_rank_without_the_flag_is_refused(tmp_path):
    """two ranks launched WITHOUT --distributed both ask for cuda:0 (upstream AppLauncher reads the rank environment only with the flag):
    the runner refuses rank 1, as rsl_rl does - never two unrelated learners on one device writing one log directory."""
    env = dict(os.environ, RL_TEST_OUT=str(tmp_path), RL_SHARE_GPU="1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc_per_node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "train_body_rank.py"), "--num_envs", "64", "--max_iterations", "1"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "expected 'cuda:1'" in p.stderr
