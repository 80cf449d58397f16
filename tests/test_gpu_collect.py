"""The 24-step collection iteration as one hipGraph launch (robot_lab_amd/collect.py) against the same iteration launched
kernel by kernel: same env seed, same storage seed, same networks -> the storages must agree bit for bit after every iteration
(fresh noise and fresh step counts on every replay come from the device words, include/rl_env.h `rl_env_graph_*`)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TASK = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"


def _setup(use_graph, N=256, T=24, overlap=False, critic_small=False, task=TASK, specialise=None, fused_act=True, clip_actions=None):
    import torch

    from robot_lab_amd.collect import Collector
    from robot_lab_amd.env import ManagerBasedRLEnv
    from robot_lab_amd.policy import MlpPolicy
    from robot_lab_amd.rollout import RolloutStorage

    env = ManagerBasedRLEnv(task, num_envs=N, seed=11, device="cuda:0", specialise=specialise)
    obs, _ = env.reset()
    od, cd, A = obs["policy"].shape[1], obs["critic"].shape[1], env.num_actions
    rng = np.random.default_rng(0)

    def net(dims):
        ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
        return MlpPolicy(ws, [0.05 * rng.standard_normal(d).astype(np.float32) for d in dims[1:]], "elu", device="cuda:0")

    actor, critic = net([od, 512, 256, 128, A]), net([cd, 512, 256, 128, 1])
    storage = RolloutStorage(N, T, od, cd, A, seed=3, device="cuda:0")
    std = torch.full((A,), 0.5, device="cuda:0")
    return env, storage, Collector(env, actor, critic, storage, std, use_graph=use_graph, overlap=overlap, critic_small=critic_small, fused_act=fused_act,
                                   clip_actions=clip_actions)


@pytest.mark.parametrize("overlap", [True, False])
def test_graph_replay_matches_eager_iterations(overlap):
    """`overlap`: the critic of step t on a second stream under env step t, time outs bootstrapped by compute_returns (the default);
    False: actor + critic as one launch in front of act."""
    import torch

    env_e, st_e, eager = _setup(False, overlap=overlap)
    env_g, st_g, graph = _setup(True, overlap=overlap)
    for it in range(4):  # iteration 0 of the graphed collector is eager + capture, 1.. are replays
        oe, og = eager.collect(), graph.collect()
        torch.cuda.synchronize()
        assert env_e.common_step_counter == env_g.common_step_counter == 24 * (it + 1)
        for name in ("observations", "privileged_observations", "actions", "mu", "actions_log_prob", "values", "rewards", "dones", "returns",
                     "advantages"):
            a, b = getattr(st_e, name), getattr(st_g, name)
            assert torch.equal(a, b), f"iteration {it}: {name} differs (max |d| {float((a.float() - b.float()).abs().max()):.3e})"
        assert torch.equal(oe["policy"], og["policy"]) and torch.equal(oe["critic"], og["critic"])
        assert torch.equal(env_e.episode_length_buf, env_g.episode_length_buf)
        if it >= 1:  # the replays draw fresh noise: not the actions of the previous iteration
            assert not torch.equal(st_g.actions, prev)
        prev = st_g.actions.clone()
    # the episode log of the last step is readable after a replay and equal to the eager one
    le, lg = dict(eager.env.extras["log"]), dict(graph.env.extras["log"])
    assert le.keys() == lg.keys()
    for k in le:
        assert float(le[k]) == float(lg[k]), k
    # direct steps and replays can be mixed: two direct steps on both sides, then another iteration
    for c in (eager, graph):
        for _ in range(2):
            c.obs, *_ = c.env.step(torch.zeros(c.env.num_envs, c.env.num_actions, device="cuda:0"))
    eager.collect(), graph.collect()
    torch.cuda.synchronize()
    assert torch.equal(st_e.actions, st_g.actions) and torch.equal(st_e.advantages, st_g.advantages)
    for e in (env_e, env_g):
        e.close()


@pytest.mark.parametrize("small", [False, True])
def test_overlapped_collection_equals_the_serial_one(small):
    """The critic off the critical path changes WHEN things run, not what is computed: same seeds -> the storage of the overlapped
    collector (graph replays included) equals the serial collector's - bit for bit with the critic through the same kernel family,
    to fp32 round-off with the small-footprint critic launch (exact-f32 MFMA instead of the bf16 x 3 split) -, time-out bootstraps included
    (N = 256 envs with episode clocks spread so that several envs time out inside the 24 steps)."""
    import torch

    env_s, st_s, serial = _setup(True, overlap=False)
    env_o, st_o, over = _setup(True, overlap=True, critic_small=small)
    for env in (env_s, env_o):
        ep = torch.arange(env.num_envs) % 50
        ep[::7] = env.max_episode_length - 1 - (torch.arange(len(ep[::7])) % 60)
        env.episode_length_buf = ep
    timeouts = 0
    for it in range(3):
        serial.collect(), over.collect()
        torch.cuda.synchronize()
        for name in ("observations", "privileged_observations", "actions", "mu", "actions_log_prob", "dones"):
            a, b = getattr(st_s, name), getattr(st_o, name)
            assert torch.equal(a, b), f"iteration {it}: {name} differs (max |d| {float((a.float() - b.float()).abs().max()):.3e})"
        for name in ("values", "rewards", "returns", "advantages"):
            a, b = getattr(st_s, name), getattr(st_o, name)
            if small:
                assert torch.allclose(a, b, rtol=1e-4, atol=2e-5), f"iteration {it}: {name} differs by {float((a - b).abs().max()):.3e}"
            else:
                assert torch.equal(a, b), f"iteration {it}: {name} differs (max |d| {float((a - b).abs().max()):.3e})"
        assert int(st_o.dones.view(torch.uint8).max()) <= 1  # the time-out marks (bit 1) are gone after compute_returns
        timeouts += int(st_o.dones.sum())
    assert timeouts > 0
    for e in (env_s, env_o):
        e.close()


def test_graph_replay_on_a_run_time_specialised_env(monkeypatch, tmp_path):
    """The captured collection loop around an env whose step kernel is a run-time plugin (robot_lab_amd/jit.py): the plugin's launches are
    captured and replayed like the library's own - graph == eager bit for bit, and both on the specialised kernel."""
    import torch

    monkeypatch.setenv("RL_ENV_JIT_CACHE", str(tmp_path))
    task = "RobotLab-Isaac-Velocity-Rough-Unitree-B2-v0"
    env_e, st_e, eager = _setup(False, task=task, specialise=True)
    env_g, st_g, graph = _setup(True, task=task, specialise=True)
    assert env_e._native.spec_id() >= 1000 and env_g._native.spec_id() == env_e._native.spec_id()
    for it in range(3):
        eager.collect(), graph.collect()
        torch.cuda.synchronize()
        for name in ("observations", "privileged_observations", "actions", "values", "rewards", "dones", "returns", "advantages"):
            a, b = getattr(st_e, name), getattr(st_g, name)
            assert torch.equal(a, b), f"iteration {it}: {name} differs (max |d| {float((a.float() - b.float()).abs().max()):.3e})"
    for e in (env_e, env_g):
        e.close()


@pytest.mark.parametrize("clip,rt", [(None, "2"), (0.8, "2"), (None, "1")])
def test_act_in_the_actor_launch_equals_the_act_kernel(clip, rt, monkeypatch):
    """The step's stochastic head in the epilogue of the actor + critic launch (include/rl_act.h, rl_mlp_forward_pair_act; VERDICT r5 item 2,
    second half) against actor + critic, then the act kernel: the same sampling function, the same slots - the storages must agree bit for
    bit, graph replays included, with and without the wrapper's action clamp.  Both split-precision pair kernels carry the epilogue:
    RL_MLP_SPLIT_RT=2 puts the 256-env pair on the one rollout sizes >= 4096 run (32 rows x one network per workgroup), =1 on the one
    smaller batches run (16 rows x both networks); the exact-f32 kernels have none and the caller falls back to act()."""
    import torch

    monkeypatch.setenv("RL_MLP_SPLIT_RT", rt)
    env_a, st_a, fused = _setup(True, fused_act=True, clip_actions=clip)
    env_b, st_b, plain = _setup(True, fused_act=False, clip_actions=clip)
    # the fused launch is really taken at this size
    from robot_lab_amd.rollout import ActEpilogue
    import ctypes as C

    ep = ActEpilogue()
    assert st_a.lib.rl_rollout_act_epilogue(st_a.handle, C.c_void_p(fused.std.data_ptr()), C.c_void_p(st_a._actions.data_ptr()), -1.0, C.byref(ep)) == 0
    obs = fused.obs
    assert fused.actor.forward_pair_act(C.c_void_p(obs["policy"].data_ptr()), fused.critic, C.c_void_p(obs["critic"].data_ptr()), ep) == 0
    torch.cuda.synchronize()
    for it in range(3):  # 0: eager + capture, 1..: replays
        oa, ob = fused.collect(), plain.collect()
        torch.cuda.synchronize()
        for name in ("observations", "privileged_observations", "actions", "mu", "sigma", "actions_log_prob", "values", "rewards", "dones", "returns",
                     "advantages"):
            a, b = getattr(st_a, name), getattr(st_b, name)
            assert torch.equal(a, b), f"iteration {it}: {name} differs (max |d| {float((a.float() - b.float()).abs().max()):.3e})"
        assert torch.equal(oa["policy"], ob["policy"]) and torch.equal(oa["critic"], ob["critic"])
        assert torch.equal(env_a.episode_length_buf, env_b.episode_length_buf)
    if clip is not None:
        assert float(st_a.actions.abs().max()) > clip  # the storage keeps the sample; the env saw the clamped action (equal states above)
    # a pair kernel without the epilogue (the exact-f32 path) reports so and launches nothing
    monkeypatch.setenv("RL_MLP_PRECISION", "f32")
    assert fused.actor.forward_pair_act(C.c_void_p(obs["policy"].data_ptr()), fused.critic, C.c_void_p(obs["critic"].data_ptr()), ep) == 1
    for e in (env_a, env_b):
        e.close()
