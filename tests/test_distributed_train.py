"""The reference's own multi-GPU entry on the shims: `python -m torch.distributed.run --nproc_per_node=N
scripts/reinforcement_learning/rsl_rl/train.py --task=<id> --headless --distributed` (`/root/reference/README.md:323-337`).

`train.py:143-150` puts rank r on `cuda:{app_launcher.local_rank}` with seed `agent_cfg.seed + app_launcher.local_rank`, so the launcher
shim must take its ranks from `LOCAL_RANK` / `RANK` (until round 4 it hard-coded 0: eight ranks on cuda:0 with one seed, silently), and
the learner must either be tied across ranks (rsl_rl: parameters broadcast from rank 0, gradient all-reduced SUM / world per mini-batch,
rank 0 logs) or refuse - never N unrelated trainings.  CPU tier: the script runs as a file under the real launcher with two ranks up to
the drop-in boundary (the env has no CPU path; `tests/ref_script_rank.py` records what it was handed), and the learner's collectives run
over gloo on a synthetic batch.  GPU tier: `tests/test_gpu_distributed_train.py` trains two ranks end to end.
"""
import json
import os
import socket
import subprocess
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
TASK = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
TRAIN = "scripts/reinforcement_learning/rsl_rl/train.py"


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(nproc, out_dir, script_args, extra_env=None, timeout=600):
    """the README's launch line, with the rank helper in front of the reference script"""
    env = dict(os.environ, RL_TEST_OUT=str(out_dir), HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc_per_node={nproc}", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "ref_script_rank.py"), TRAIN, *script_args]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=str(out_dir))
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return [json.load(open(os.path.join(str(out_dir), f"rank{r}.json"))) for r in range(nproc)], p


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")
def test_train_py_distributed_puts_rank_r_on_cuda_r_with_seed_plus_r(tmp_path):
    recs, _ = launch(2, tmp_path, ["--task", TASK, "--num_envs", "16", "--headless", "--max_iterations", "1", "--distributed"])
    for r, rec in enumerate(recs):
        assert rec["world"] == 2 and rec["rank"] == r
        assert rec["launcher_local_rank"] == r and rec["launcher_global_rank"] == r  # AppLauncher reads LOCAL_RANK / RANK
        assert rec["sim_device"] == f"cuda:{r}"                                       # train.py:144
        assert rec["env_seed"] == 42 + r                                              # train.py:148-149 (agent cfg seed 42)
        assert rec["num_envs"] == 16
    assert {rec["env_seed"] for rec in recs} == {42, 43}


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")
def test_without_the_flag_the_launcher_is_rank_zero_as_upstream(tmp_path):
    """[UPSTREAM AppLauncher] reads the rank environment only when `distributed` is set; two ranks launched WITHOUT the flag both ask
    for cuda:0 with the same seed - upstream rsl_rl then refuses rank 1 (device != cuda:local_rank), and so does the stand-in
    (`test_learner_group_refuses_a_rank_on_somebody_elses_device`)."""
    recs, _ = launch(2, tmp_path, ["--task", TASK, "--num_envs", "16", "--headless", "--max_iterations", "1"])
    for rec in recs:
        assert rec["launcher_local_rank"] == 0 and rec["sim_device"] == "cuda:0" and rec["env_seed"] == 42


def test_app_launcher_reads_the_rank_environment(monkeypatch):
    import argparse

    from robot_lab_amd import shims

    shims.install()
    from isaaclab.app import AppLauncher

    monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setenv("RANK", "11")
    a = AppLauncher(argparse.Namespace(distributed=True, device="cuda:0"))
    assert (a.local_rank, a.global_rank, a.device_id) == (3, 11, 3)
    a = AppLauncher({"distributed": False, "device": "cuda:2"})
    assert (a.local_rank, a.global_rank, a.device_id) == (0, 0, 2)
    a = AppLauncher(argparse.Namespace(headless=True))  # zero_agent.py / play.py: no such flag at all
    assert (a.local_rank, a.global_rank) == (0, 0)
    a = AppLauncher(distributed=True)
    assert a.local_rank == 3


def test_learner_group_refuses_a_rank_on_somebody_elses_device(monkeypatch):
    from robot_lab_amd.dist import LearnerGroup

    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    with pytest.raises(ValueError, match="expected 'cuda:1'"):
        LearnerGroup("cuda:0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    g = LearnerGroup("cuda:0")  # a world of one: no process group, every collective a no-op
    assert not g.enabled and g.is_main
    lin = torch.nn.Linear(3, 2)
    before = [p.detach().clone() for p in lin.parameters()]
    g.broadcast_parameters(lin)
    lin(torch.ones(1, 3)).sum().backward()
    g.reduce_gradients(lin)
    assert all(torch.equal(a, b) for a, b in zip(before, lin.parameters())) and float(g.mean(torch.tensor(0.25))) == 0.25


def test_physical_device_wraps_only_under_the_self_test_switch(monkeypatch):
    from robot_lab_amd import dist as rd

    monkeypatch.delenv("RL_SHARE_GPU", raising=False)
    assert str(rd.physical_device("cuda:5")) == "cuda:5"
    monkeypatch.setenv("RL_SHARE_GPU", "1")
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    assert str(rd.physical_device("cuda:5")) == "cuda:1" and str(rd.physical_device("cuda:0")) == "cuda:0"
    assert str(rd.physical_device("cpu")) == "cpu"


# ------------------------------------------------------------------------------------------------------------------------------------
# the learner's collectives over gloo, world size 2
# ------------------------------------------------------------------------------------------------------------------------------------
def _learner_rank(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from test_ppo import _fake_storage

    from robot_lab_amd.dist import LearnerGroup
    from robot_lab_amd.ppo import PPO, ActorCritic

    group = LearnerGroup(f"cuda:{rank}")  # the device STRING is the contract; the tensors of this test live on the host (gloo)
    torch.manual_seed(100 + rank)         # train.py:148: every rank has its own seed, hence its own initialisation ...
    pol = ActorCritic(10, 14, 3, actor_hidden=(32, 32), critic_hidden=(32, 32))
    first = pol.actor[0].weight.detach().clone()
    group.broadcast_parameters(pol)       # ... until rank 0's is broadcast
    after_bcast = torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).clone()
    # one mini-batch by hand: the reduced gradient is the mean of the two ranks' gradients
    st = _fake_storage(pol, seed=7 + rank)  # every rank collected its own batch
    loss = pol.actor(st.observations.reshape(-1, 10)).pow(2).mean() + pol.critic(st.privileged_observations.reshape(-1, 14)).pow(2).mean()
    loss.backward()
    local = torch.cat([p.grad.reshape(-1) for p in pol.parameters() if p.grad is not None]).clone()
    group.reduce_gradients(pol)
    reduced = torch.cat([p.grad.reshape(-1) for p in pol.parameters() if p.grad is not None]).clone()
    # a whole update: 5 epochs x 4 mini-batches, adaptive learning rate from the rank-averaged KL
    alg = PPO(pol, learning_rate=1e-3, group=group)
    out = alg.update(st, torch.Generator().manual_seed(1 + rank))
    final = torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).clone()
    q.put(dict(rank=rank, first=first.numpy(), after_bcast=after_bcast.numpy(), local=local.numpy(), reduced=reduced.numpy(), final=final.numpy(),
               lr=out["learning_rate"], kl=out["kl"], is_main=group.is_main, backend=group.backend))
    import torch.distributed as dist

    dist.barrier()
    dist.destroy_process_group()


def test_two_learners_stay_one_model_over_gloo():
    import numpy as np
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_learner_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(2)), key=lambda d: d["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = got
    assert a["backend"] == "gloo" and a["is_main"] and not b["is_main"]
    assert not np.array_equal(a["first"], b["first"])                 # different seeds, different initialisations ...
    np.testing.assert_array_equal(a["after_bcast"], b["after_bcast"])  # ... one model after the broadcast
    assert not np.array_equal(a["local"], b["local"])                 # different batches, different gradients ...
    np.testing.assert_array_equal(a["reduced"], b["reduced"])          # ... one gradient after the all-reduce: the mean
    np.testing.assert_allclose(a["reduced"], 0.5 * (a["local"] + b["local"]), rtol=1e-6, atol=1e-9)
    np.testing.assert_array_equal(a["final"], b["final"])              # 20 Adam steps later still bit-identical replicas
    assert a["lr"] == b["lr"] and a["kl"] == b["kl"]                   # the adaptive schedule decided alike on both
    assert not np.array_equal(a["final"], a["after_bcast"])


def test_stand_in_runner_refuses_what_it_does_not_implement(monkeypatch):
    """options of the agent cfg the stand-in cannot honour are refused, not ignored (ANYmal-D's `critic <- ['policy']`,
    a log-parameterised noise, empirical normalisation)."""
    from robot_lab_amd import shims

    shims.install()
    from rsl_rl.runners import OnPolicyRunner

    base = dict(policy=dict(class_name="ActorCritic", activation="elu", init_noise_std=1.0, actor_hidden_dims=[8], critic_hidden_dims=[8]),
                algorithm=dict(class_name="PPO"), num_steps_per_env=4, seed=1, save_interval=10)
    env = types.SimpleNamespace(unwrapped=None)
    for patch, needle in (({"obs_groups": {"policy": ["policy"], "critic": ["policy"]}}, "obs_groups"),
                          ({"policy": dict(base["policy"], noise_std_type="log")}, "noise_std_type"),
                          ({"empirical_normalization": True}, "empirical_normalization"),
                          ({"algorithm": dict(class_name="PPO", normalize_advantage_per_mini_batch=True)}, "normalize_advantage")):
        with pytest.raises(NotImplementedError, match=needle):
            OnPolicyRunner(env, dict(base, **patch), device="cuda:0")
