"""The learner of robot_lab_amd/ppo.py (a restatement of rsl_rl's PPO.update; rsl-rl-lib itself is absent, so UNPINNED against the
library): its pieces against their definitions, on CPU.  The end-to-end check - a robot learns to follow velocity commands on this
simulator - is tests/test_gpu_train.py / tools/train_demo.py."""
import types

import numpy as np
import torch

from robot_lab_amd.ppo import PPO, ActorCritic, gaussian_entropy, gaussian_kl, gaussian_log_prob


def test_gaussian_pieces_match_torch_distributions():
    g = torch.Generator().manual_seed(0)
    mu, mu2 = torch.randn(64, 12, generator=g), torch.randn(64, 12, generator=g)
    sd, sd2 = torch.rand(64, 12, generator=g) + 0.2, torch.rand(64, 12, generator=g) + 0.2
    a = torch.randn(64, 12, generator=g)
    p, q = torch.distributions.Normal(mu, sd), torch.distributions.Normal(mu2, sd2)
    torch.testing.assert_close(gaussian_log_prob(a, mu, sd), p.log_prob(a).sum(-1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gaussian_entropy(sd), p.entropy().sum(-1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gaussian_kl(mu, sd, mu2, sd2), torch.distributions.kl_divergence(p, q).sum(-1), rtol=1e-4, atol=1e-4)


def _fake_storage(policy, T=8, N=64, od=10, cd=14, A=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    obs, cobs = torch.randn(T, N, od, generator=g), torch.randn(T, N, cd, generator=g)
    with torch.no_grad():
        mu, sd = policy.distribution(obs)
        act = mu + sd * torch.randn(mu.shape, generator=g)
        logp = gaussian_log_prob(act, mu, sd)
        val = policy.critic(cobs).squeeze(-1)
    # a batch in which action dimension 0 being positive is "good": the update must raise the policy's mean along it
    adv = act[..., 0].clone()
    adv = (adv - adv.mean()) / adv.std()
    ret = val + adv
    return types.SimpleNamespace(num_transitions_per_env=T, num_envs=N, observations=obs, privileged_observations=cobs, actions=act, values=val.unsqueeze(-1),
                                 returns=ret.unsqueeze(-1), advantages=adv.unsqueeze(-1), actions_log_prob=logp.unsqueeze(-1), mu=mu, sigma=sd.expand_as(mu).contiguous())


def test_update_follows_the_advantage_and_adapts_the_learning_rate():
    torch.manual_seed(0)
    pol = ActorCritic(10, 14, 3, actor_hidden=(32, 32), critic_hidden=(32, 32))
    alg = PPO(pol, learning_rate=1e-3)
    st = _fake_storage(pol)
    with torch.no_grad():
        m0 = pol.actor(st.observations.reshape(-1, 10))[:, 0].mean().item()
        v0 = ((pol.critic(st.privileged_observations.reshape(-1, 14)).view(-1) - st.returns.view(-1)) ** 2).mean().item()
    out = alg.update(st, torch.Generator().manual_seed(1))
    with torch.no_grad():
        m1 = pol.actor(st.observations.reshape(-1, 10))[:, 0].mean().item()
        v1 = ((pol.critic(st.privileged_observations.reshape(-1, 14)).view(-1) - st.returns.view(-1)) ** 2).mean().item()
    assert m1 > m0, "the policy mean did not move along the advantage"
    assert v1 < v0, "the value loss did not go down"
    assert all(np.isfinite(v) for v in out.values()) and 1e-5 <= out["learning_rate"] <= 1e-2 and out["kl"] >= 0.0
    # gradient-norm clipping and the clipped surrogate keep a single update small: ratio stays near the clip range
    with torch.no_grad():
        mu, sd = pol.distribution(st.observations.reshape(-1, 10))
        ratio = torch.exp(gaussian_log_prob(st.actions.reshape(-1, 3), mu, sd) - st.actions_log_prob.view(-1))
    assert 0.5 < ratio.mean().item() < 1.5


def test_state_dict_layout_is_rsl_rls():
    pol = ActorCritic(45, 235, 12)
    keys = set(pol.state_dict())
    assert {"std", "actor.0.weight", "actor.6.bias", "critic.0.weight", "critic.6.weight"} <= keys
    from robot_lab_amd.policy import MlpPolicy  # the inference side reads the same layout (from_state_dict)

    idx = sorted({int(k.split(".")[1]) for k in keys if k.startswith("actor.") and k.endswith(".weight")})
    assert idx == [0, 2, 4, 6] and hasattr(MlpPolicy, "from_state_dict") and hasattr(MlpPolicy, "set_weights")
