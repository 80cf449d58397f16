"""Fused MLP inference (include/rl_policy.h): oracle pinned to torch on CPU; HIP kernel vs oracle on the GPU."""
import ctypes
import os

import numpy as np
import pytest

from oracle.policy import mlp_forward
from robot_lab_amd.policy import POLICY_EXPORTS, POLICY_LIB


def _net(dims, seed):
    rng = np.random.default_rng(seed)
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    bs = [(0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32) for i in range(len(dims) - 1)]
    return ws, bs


@pytest.mark.parametrize("act", ["elu", "relu", "tanh"])
def test_oracle_matches_torch_sequential(act):
    import torch

    dims = [45, 512, 256, 128, 12]  # rsl_rl_ppo_cfg.py:18-20 actor for the A1 policy observation
    ws, bs = _net(dims, 0)
    layers = []
    for i in range(len(ws)):
        lin = torch.nn.Linear(dims[i], dims[i + 1]).double()
        lin.weight.data = torch.tensor(ws[i], dtype=torch.float64)
        lin.bias.data = torch.tensor(bs[i], dtype=torch.float64)
        layers.append(lin)
        if i < len(ws) - 1:
            layers.append({"elu": torch.nn.ELU(), "relu": torch.nn.ReLU(), "tanh": torch.nn.Tanh()}[act])
    x = np.random.default_rng(1).uniform(-3, 3, (37, 45))
    want = torch.nn.Sequential(*layers)(torch.tensor(x)).detach().numpy()
    np.testing.assert_allclose(mlp_forward(x, ws, bs, act), want, rtol=1e-12, atol=1e-12)


def test_policy_library_exports():
    assert os.path.isfile(POLICY_LIB), "librl_policy_hip.so not built"
    lib = ctypes.CDLL(POLICY_LIB)
    for name in POLICY_EXPORTS:
        assert hasattr(lib, name), name


# both contraction paths of csrc/rl_policy.hip: "" = the default (split-bf16: three exact bf16 planes per operand, six bf16 MFMAs per
# 32-deep block, fp32 accumulate), "f32" = RL_MLP_PRECISION=f32, the exact-f32 MFMA kernels.  Same tolerance for both.
PRECISIONS = ["", "f32"]


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("dims,act,N", [
    ([45, 512, 256, 128, 12], "elu", 4096),     # A1 actor
    ([235, 512, 256, 128, 1], "elu", 4096),     # A1 critic
    ([96, 512, 256, 128, 29], "elu", 2048),     # G1 actor
    ([57, 512, 256, 128, 16], "elu", 100),      # Go2W actor, ragged row count
    ([7, 33, 5], "tanh", 19),                   # odd widths: every padding path
    ([48, 64], "relu", 1),                      # single layer, single row
])
def test_hip_mlp_matches_oracle(dims, act, N, prec, monkeypatch):
    import torch

    from robot_lab_amd.policy import MlpPolicy

    if prec:
        monkeypatch.setenv("RL_MLP_PRECISION", prec)

    ws, bs = _net(dims, 3)
    pol = MlpPolicy(ws, bs, act, device="cuda:0")
    x = np.random.default_rng(4).uniform(-2, 2, (N, dims[0])).astype(np.float32)
    got = pol(torch.from_numpy(x).cuda()).cpu().numpy()
    want = mlp_forward(x, ws, bs, act)
    # exact-fp32 MFMA = a k-ordered fmaf chain: round-off of a K <= 512 dot product; the split path sees fewer roundings
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
    err = np.abs(got - want) / (1.0 + np.abs(want))
    print(f"\n[mlp {prec or 'split-bf16'}] {dims} x {N}: max |err| / (1 + |y|) = {err.max():.2e}")
    pol.close()


@pytest.mark.gpu
def test_split_path_survives_large_and_tiny_values():
    """The bf16 planes keep fp32's exponent range: activations of 1e4 and inputs of 1e-20 go through the split path like through
    the f32 kernels (an fp16 split would overflow / flush them)."""
    import torch

    from robot_lab_amd.policy import MlpPolicy

    dims = [40, 64, 8]
    ws, bs = _net(dims, 11)
    ws[0] *= 300.0
    pol = MlpPolicy(ws, bs, "relu", device="cuda:0")
    rng = np.random.default_rng(12)
    x = rng.uniform(-30, 30, (64, 40)).astype(np.float32)
    x[::2] *= 1e-20
    got = pol(torch.from_numpy(x).cuda()).cpu().numpy()
    want = mlp_forward(x, ws, bs, "relu")
    assert np.abs(want).max() > 1e3
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
    pol.close()


@pytest.mark.gpu
def test_hip_mlp_from_rsl_rl_state_dict_and_env_obs():
    """state_dict layout of rsl_rl's ActorCritic + the env's policy observation as input (play.py:207,246)."""
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv
    from robot_lab_amd.policy import MlpPolicy

    env = ManagerBasedRLEnv("RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0", num_envs=256, seed=1, device="cuda:0")
    obs, _ = env.reset()
    dims = [45, 512, 256, 128, 12]
    ws, bs = _net(dims, 9)
    sd = {}
    for i, (w, b) in enumerate(zip(ws, bs)):
        sd[f"actor.{2 * i}.weight"], sd[f"actor.{2 * i}.bias"] = torch.tensor(w), torch.tensor(b)
    sd["std"] = torch.ones(12)
    pol = MlpPolicy.from_state_dict(sd, "actor", "elu", device="cuda:0")
    for _ in range(3):
        act = pol(obs)
        np.testing.assert_allclose(act.cpu().numpy(), mlp_forward(obs["policy"].cpu().numpy(), ws, bs), rtol=2e-5, atol=2e-5)
        obs, *_ = env.step(act.clamp(-1, 1))
    env.close()
    pol.close()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("N,da,dc", [(4096, [45, 512, 256, 128, 12], [235, 512, 256, 128, 1]), (37, [45, 512, 256, 128, 12], [235, 512, 256, 128, 1]),
                                     (4099, [45, 512, 256, 128, 12], [235, 512, 256, 128, 1]),  # fused kernel, ragged last row tile
                                     (3100, [96, 400, 200, 29], [310, 512, 136, 1]),            # widths that are no multiples of 128 / 16
                                     (3072, [45, 64, 12], [235, 512, 1])])                        # two layers, very different networks
def test_hip_mlp_pair_is_bitwise_the_two_single_launches(N, da, dc, prec, monkeypatch):
    """rl_mlp_forward_pair (actor + critic in one launch; from 3072 rows on the fused kernel that runs both networks on a row
    tile) = the same k-ordered MFMA chains as the single launches: identical bits."""
    import torch

    from robot_lab_amd.policy import MlpPolicy

    if prec:
        monkeypatch.setenv("RL_MLP_PRECISION", prec)

    wa, ba = _net(da, 5)
    wc, bc = _net(dc, 6)
    actor, critic = MlpPolicy(wa, ba, "elu", device="cuda:0"), MlpPolicy(wc, bc, "elu", device="cuda:0")
    rng = np.random.default_rng(7)
    xo = torch.from_numpy(rng.uniform(-2, 2, (N, da[0])).astype(np.float32)).cuda()
    xc = torch.from_numpy(rng.uniform(-2, 2, (N, dc[0])).astype(np.float32)).cuda()
    ya, yc = actor(xo).clone(), critic(xc).clone()
    pa, pc = actor.forward_pair(xo, critic, xc)
    assert torch.equal(pa, ya) and torch.equal(pc, yc)
    np.testing.assert_allclose(pc.cpu().numpy(), mlp_forward(xc.cpu().numpy(), wc, bc, "elu"), rtol=2e-5, atol=2e-5)
    actor.close()
    critic.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rt", ["1", "2"])
@pytest.mark.parametrize("N,da,dc", [(37, [45, 512, 256, 128, 12], [235, 512, 256, 128, 1]), (4099, [45, 512, 256, 128, 12], [235, 512, 256, 128, 1]),
                                     (1000, [96, 400, 200, 29], [310, 512, 136, 1])])
def test_split_kernels_agree_bitwise(N, da, dc, rt, monkeypatch):
    """The two kernels of the split-bf16 path - 16 rows x both networks per workgroup (RL_MLP_SPLIT_RT=1) and 32 rows x one network
    per workgroup (RL_MLP_SPLIT_RT=2, what a 4096-row rollout step launches) - run the same per-tile MFMA chains: forced onto every
    shape (ragged row counts, odd widths), each gives the bits of the default single launches and stays inside the oracle tolerance."""
    import torch

    from robot_lab_amd.policy import MlpPolicy

    wa, ba = _net(da, 15)
    wc, bc = _net(dc, 16)
    actor, critic = MlpPolicy(wa, ba, "elu", device="cuda:0"), MlpPolicy(wc, bc, "elu", device="cuda:0")
    rng = np.random.default_rng(17)
    xo = torch.from_numpy(rng.uniform(-2, 2, (N, da[0])).astype(np.float32)).cuda()
    xc = torch.from_numpy(rng.uniform(-2, 2, (N, dc[0])).astype(np.float32)).cuda()
    ya, yc = actor(xo).clone(), critic(xc).clone()  # default dispatch
    monkeypatch.setenv("RL_MLP_SPLIT_RT", rt)
    pa, pc = actor.forward_pair(xo, critic, xc)
    sa = actor(xo).clone()
    assert torch.equal(pa, ya) and torch.equal(pc, yc) and torch.equal(sa, ya)
    np.testing.assert_allclose(pa.cpu().numpy(), mlp_forward(xo.cpu().numpy(), wa, ba, "elu"), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(pc.cpu().numpy(), mlp_forward(xc.cpu().numpy(), wc, bc, "elu"), rtol=2e-5, atol=2e-5)
    actor.close()
    critic.close()
