"""Rollout storage / GAE (include/rl_rollout.h): the oracle against the textbook definitions on CPU; the HIP
kernels against the oracle on the GPU.  rsl-rl-lib 3.0.1 itself is absent here (third-party): parity with the library
is unpinned, the oracle restates its published arithmetic (oracle/rollout.py)."""
import ctypes
import os

import numpy as np
import pytest

from oracle.rollout import RolloutOracle, standard_normal
from robot_lab_amd.rollout import ROLLOUT_EXPORTS, ROLLOUT_LIB

GAMMA, LAM = 0.99, 0.95  # .../unitree_a1/agents/rsl_rl_ppo_cfg.py:33-34


def _fill(ora, rng, p_done=0.05, p_timeout=0.03):
    N, T, A = ora.N, ora.T, ora.A
    std = np.exp(rng.uniform(-1.0, 0.3, A))
    feeds = []
    for t in range(T):
        f = dict(obs=rng.standard_normal(ora.obs.shape[1:]), critic=rng.standard_normal(ora.critic_obs.shape[1:]),
                 mean=rng.standard_normal((N, A)), std=std, values=rng.standard_normal(N) * 3.0, rewards=rng.standard_normal(N),
                 terminated=rng.random(N) < p_done, time_outs=rng.random(N) < p_timeout)
        f = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in f.items()}
        feeds.append(f)
        ora.act(f["obs"], f["critic"], f["mean"], f["std"], f["values"])
        ora.record(f["rewards"], f["terminated"], f["time_outs"], GAMMA)
    last = (rng.standard_normal(N) * 3.0).astype(np.float32)
    return feeds, last


def test_gae_matches_the_definition():
    """A_t = sum_k (gamma lam)^k delta_{t+k}, the sum cut at the first done (Schulman et al. 2016, eq. 16)."""
    rng = np.random.default_rng(0)
    ora = RolloutOracle(64, 24, 5, 7, 3, seed=1)
    feeds, last = _fill(ora, rng, p_done=0.1)
    ora.compute_returns(last, GAMMA, LAM, normalize_advantage=False)
    V = np.concatenate([ora.values, last[None].astype(np.float64)], 0)
    nt = 1.0 - ora.dones
    delta = ora.rewards + nt * GAMMA * V[1:] - V[:-1]
    for t in (0, 5, 23):
        want, w = np.zeros(ora.N), np.ones(ora.N)
        for k in range(t, ora.T):
            want += w * delta[k]
            w = w * GAMMA * LAM * nt[k]
        np.testing.assert_allclose(ora.advantages[t], want, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ora.returns, ora.advantages + ora.values, rtol=1e-12, atol=1e-12)
    # bootstrapping on time outs: the stored reward carries gamma * V_t where the step timed out
    to = np.stack([f["time_outs"] for f in feeds])
    raw = np.stack([f["rewards"] for f in feeds]).astype(np.float64)
    np.testing.assert_allclose(ora.rewards, raw + GAMMA * ora.values * to, rtol=0, atol=0)
    assert to.any() and ora.dones.any()


def test_normalised_advantages_and_overflow():
    rng = np.random.default_rng(1)
    ora = RolloutOracle(128, 8, 4, 4, 2, seed=2)
    _, last = _fill(ora, rng)
    ora.compute_returns(last, GAMMA, LAM)
    assert abs(ora.advantages.mean()) < 1e-12 and abs(ora.advantages.std(ddof=1) - 1.0) < 1e-6
    with pytest.raises(OverflowError):
        ora.act(np.zeros((128, 4)), np.zeros((128, 4)), np.zeros((128, 2)), np.ones(2), np.zeros(128))


def test_action_noise_is_standard_normal():
    z = standard_normal(seed=5, n_envs=20000, counter=3, act_dim=12)
    assert z.shape == (20000, 12) and np.isfinite(z).all()
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and abs((z**3).mean()) < 0.03 and abs((z**4).mean() - 3.0) < 0.1
    assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.02  # the two outputs of one Box-Muller pair
    assert not np.array_equal(z, standard_normal(5, 20000, 4, 12))  # the counter moves the stream
    np.testing.assert_array_equal(z[:, :5], standard_normal(5, 20000, 3, 5))  # a draw does not depend on act_dim


def test_log_prob_is_the_gaussian_density():
    from scipy.stats import norm

    ora = RolloutOracle(16, 1, 3, 3, 5, seed=9)
    rng = np.random.default_rng(2)
    mean, std = rng.standard_normal((16, 5)), np.exp(rng.uniform(-1, 1, 5))
    a = ora.act(np.zeros((16, 3)), np.zeros((16, 3)), mean, std, np.zeros(16))
    np.testing.assert_allclose(ora.log_prob[0], norm.logpdf(a, mean, std).sum(axis=1), rtol=1e-12, atol=1e-12)


def test_rollout_library_exports():
    assert os.path.isfile(ROLLOUT_LIB), "librl_rollout_hip.so not built"
    lib = ctypes.CDLL(ROLLOUT_LIB)
    for name in ROLLOUT_EXPORTS:
        assert hasattr(lib, name), name


@pytest.mark.gpu
@pytest.mark.parametrize("N,T,obs,critic,A", [
    (4096, 24, 45, 235, 12),  # A1 rough: rsl_rl_ppo_cfg.py:11, obs / critic widths of the A1 bundles
    (2048, 24, 96, 310, 29),  # G1-sized action rows (29 is not a multiple of the 4-wide Philox block)
    (37, 3, 5, 7, 1),         # ragged everything
    (37, 3, 45, 235, 12),     # A1 widths on a ragged env count: slots of the [T][N][dim] storage are only 4-byte aligned
])
def test_hip_rollout_matches_oracle(N, T, obs, critic, A):
    import torch

    from robot_lab_amd.rollout import RolloutStorage

    seed = 1234 + N
    ora = RolloutOracle(N, T, obs, critic, A, seed)
    feeds, last = _fill(ora, np.random.default_rng(N))
    st = RolloutStorage(N, T, obs, critic, A, seed=seed, device="cuda:0")
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda:0")  # noqa: E731
    for t, f in enumerate(feeds):
        assert st.step == t
        a = st.act(dev(f["obs"]), dev(f["critic"]), dev(f["mean"]), dev(f["std"]), dev(f["values"]))
        np.testing.assert_allclose(a.cpu().numpy(), ora.actions[t], rtol=1e-5, atol=2e-6)
        st.process_env_step(dev(f["rewards"]), dev(f["terminated"]), dev(f["time_outs"]), GAMMA)
    with pytest.raises(Exception, match="overflow"):
        st.act(dev(feeds[0]["obs"]), dev(feeds[0]["critic"]), dev(feeds[0]["mean"]), dev(feeds[0]["std"]), dev(feeds[0]["values"]))
    st.compute_returns(dev(last), GAMMA, LAM, normalize_advantage=False)
    torch.cuda.synchronize()
    # byte / copy work: exact
    np.testing.assert_array_equal(st.observations.cpu().numpy(), ora.obs.astype(np.float32))
    np.testing.assert_array_equal(st.privileged_observations.cpu().numpy(), ora.critic_obs.astype(np.float32))
    np.testing.assert_array_equal(st.mu.cpu().numpy(), ora.mu.astype(np.float32))
    np.testing.assert_array_equal(st.sigma.cpu().numpy(), ora.sigma.astype(np.float32))
    np.testing.assert_array_equal(st.values.cpu().numpy(), ora.values.astype(np.float32))
    np.testing.assert_array_equal(st.dones.cpu().numpy(), ora.dones)
    # fp32 arithmetic: tolerances in the test, as the oracle is fp64
    np.testing.assert_allclose(st.actions.cpu().numpy(), ora.actions, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(st.actions_log_prob.cpu().numpy(), ora.log_prob, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(st.rewards.cpu().numpy(), ora.rewards, rtol=1e-6, atol=1e-6)
    ora.compute_returns(last, GAMMA, LAM, normalize_advantage=False)
    np.testing.assert_allclose(st.returns.cpu().numpy(), ora.returns, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(st.advantages.cpu().numpy(), ora.advantages, rtol=1e-5, atol=2e-5)
    raw = st.advantages.clone()
    st.compute_returns(dev(last), GAMMA, LAM, normalize_advantage=True)
    ora.compute_returns(last, GAMMA, LAM, normalize_advantage=True)
    np.testing.assert_allclose(st.advantages.cpu().numpy(), ora.advantages, rtol=1e-4, atol=2e-5)
    if N * T > 1:  # against torch on the kernel's own raw advantages
        want = (raw - raw.mean()) / (raw.std() + 1e-8)
        np.testing.assert_allclose(st.advantages.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-6)
    # clear() reopens the storage; the noise stream keeps running (different actions for the same inputs)
    st.clear()
    f = feeds[0]
    a2 = st.act(dev(f["obs"]), dev(f["critic"]), dev(f["mean"]), dev(f["std"]), dev(f["values"])).cpu().numpy()
    assert st.step == 0 and not np.array_equal(a2, ora.actions[0].astype(np.float32))
    st.close()


@pytest.mark.gpu
def test_hip_rollout_is_reproducible_and_checks_order():
    import torch

    from robot_lab_amd.rollout import RlRolloutError, RolloutStorage

    N, T, A = 512, 4, 12
    rng = np.random.default_rng(3)
    outs = []
    for rep in range(2):
        st = RolloutStorage(N, T, 45, 235, A, seed=77, device="cuda:0")
        g = torch.Generator(device="cpu").manual_seed(5)
        for t in range(T):
            mk = lambda *s: torch.randn(*s, generator=g).cuda()  # noqa: E731
            st.act(mk(N, 45), mk(N, 235), mk(N, A), torch.full((A,), 0.7).cuda(), mk(N))
            if t == 0 and rep == 0:
                z = lambda *s: torch.zeros(*s, device="cuda:0")  # noqa: E731
                with pytest.raises(RlRolloutError, match="twice"):
                    st.act(z(N, 45), z(N, 235), z(N, A), torch.full((A,), 0.7).cuda(), z(N))
            st.process_env_step(mk(N), torch.zeros(N, dtype=torch.uint8).cuda(), (torch.rand(N, generator=g) < 0.1).cuda(), GAMMA)
        st.compute_returns(torch.randn(N, generator=g).cuda(), GAMMA, LAM)
        outs.append((st.actions.cpu().numpy().copy(), st.advantages.cpu().numpy().copy(), st.returns.cpu().numpy().copy()))
        st.close()
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)  # same seed, same inputs -> same bits (no atomics in the reductions)
    st = RolloutStorage(N, T, 45, 235, A, device="cuda:0")
    with pytest.raises(RlRolloutError, match="rl_rollout_act"):
        st.process_env_step(torch.zeros(N).cuda(), torch.zeros(N, dtype=torch.uint8).cuda(), torch.zeros(N, dtype=torch.uint8).cuda(), GAMMA)
    with pytest.raises(RlRolloutError, match="full storage"):
        st.compute_returns(torch.zeros(N).cuda(), GAMMA, LAM)
    st.close()


@pytest.mark.gpu
def test_fused_step_record_equals_the_separate_record_kernel():
    """rl_env_step_record (the env kernel writes rewards + time-out bootstrap and dones into the storage slot) against
    rl_env_step followed by rl_rollout_record on a twin env: identical bits, including envs that time out."""
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv
    from robot_lab_amd.rollout import RolloutStorage

    task, N, T = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0", 256, 4
    envs = [ManagerBasedRLEnv(task, num_envs=N, seed=9, device="cuda:0") for _ in range(2)]
    sts = [RolloutStorage(N, T, 45, 235, 12, seed=3, device="cuda:0") for _ in range(2)]
    ep = torch.zeros(N, dtype=torch.int64)
    ep[::5] = envs[0].max_episode_length - 2  # a fifth of the envs time out at the second step
    obs = []
    for e in envs:
        o, _ = e.reset()
        e.episode_length_buf = ep
        obs.append(o)
    g = torch.Generator().manual_seed(1)
    std = torch.full((12,), 0.8, device="cuda:0")
    for t in range(T):
        mean, values = torch.randn(N, 12, generator=g).cuda(), torch.randn(N, generator=g).cuda()
        acts = [st.act(o["policy"], o["critic"], mean, std, values) for st, o in zip(sts, obs)]
        assert torch.equal(acts[0], acts[1])
        o0, rew, term, tout, _ = envs[0].step(acts[0], rollout=sts[0], gamma=GAMMA)       # fused
        o1, rew1, term1, tout1, _ = envs[1].step(acts[1])                                 # separate
        sts[1].process_env_step(rew1, term1, tout1, GAMMA)
        obs = [o0, o1]
        assert torch.equal(rew, rew1) and torch.equal(tout, tout1)
        assert sts[0].step == sts[1].step == t + 1
    torch.cuda.synchronize()
    assert torch.equal(sts[0].rewards, sts[1].rewards) and torch.equal(sts[0].dones, sts[1].dones)
    assert bool(sts[0].dones.any()) and not torch.equal(sts[0].rewards[1], torch.zeros(N, device="cuda:0"))
    boot = sts[0].dones[1] & ~torch.zeros(N, dtype=torch.bool, device="cuda:0")
    assert int(boot.sum()) >= N // 5
    for x in sts + envs:
        x.close()
