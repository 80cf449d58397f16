"""The learner half of a PPO iteration, so that a policy can actually be TRAINED on this env without the (absent) rsl-rl-lib:

    collect (HIP: csrc/rl_env.hip + rl_policy.hip + rl_rollout.hip, one hipGraph launch per iteration - robot_lab_amd/collect.py)
    -> update (this file: torch autograd on the stored batch) -> push the new parameters into the inference kernels in place
    (rl_mlp_set_weights) -> collect ...

What the reference gets from `runner.learn(...)` (scripts/reinforcement_learning/rsl_rl/train.py:224 -> rsl_rl `OnPolicyRunner.learn`
-> `PPO.update`), with the hyper-parameters of `.../unitree_a1/agents/rsl_rl_ppo_cfg.py:10-37`.  rsl-rl-lib (3.0.1) is third-party
and not in the reference tree: this restates its published update rule - clipped surrogate, clipped value loss, entropy bonus,
KL-adaptive learning rate, gradient-norm clipping, 5 epochs x 4 mini-batches over the T x N transitions - and is NOT pinned to the
library (no copy of it exists here); `tests/test_ppo.py` checks the pieces against their definitions.  The update is plain PyTorch
(autograd + rocBLAS / hipBLASLt GEMMs): it is off the env-step path this repository is about, host-side plumbing like the launcher,
and it is what makes `tools/train_demo.py` - the end-to-end check that the simulator is a learnable environment - possible."""
from __future__ import annotations

import torch
from torch import nn


def mlp(dims, activation=nn.ELU):
    layers = []
    for i in range(len(dims) - 1):
        layers.append(nn.Linear(dims[i], dims[i + 1]))
        if i + 2 < len(dims):
            layers.append(activation())
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    """rsl_rl `ActorCritic` as the cfg builds it (rsl_rl_ppo_cfg.py:15-22): two ELU MLPs, a state-independent standard deviation
    (`noise_std_type="scalar"`, init 1.0), no observation normalisation.  `state_dict()` keys follow rsl_rl (`actor.<2l>.weight`,
    `critic.<2l>.weight`, `std`), so checkpoints load into `robot_lab_amd.policy.MlpPolicy.from_state_dict` and the shim's exporters."""

    def __init__(self, obs_dim, critic_obs_dim, act_dim, actor_hidden=(512, 256, 128), critic_hidden=(512, 256, 128), init_noise_std=1.0):
        super().__init__()
        self.actor = mlp([obs_dim, *actor_hidden, act_dim])
        self.critic = mlp([critic_obs_dim, *critic_hidden, 1])
        self.std = nn.Parameter(init_noise_std * torch.ones(act_dim))

    def distribution(self, obs):
        mean = self.actor(obs)
        return mean, self.std.expand_as(mean)


def gaussian_log_prob(actions, mean, std):
    """sum over the action dimensions of log N(a; mean, std^2) - torch.distributions.Normal(mean, std).log_prob(a).sum(-1)"""
    var = std * std
    return (-0.5 * (actions - mean) ** 2 / var - torch.log(std) - 0.9189385332046727).sum(-1)


def gaussian_entropy(std):
    return (0.5 + 0.9189385332046727 + torch.log(std)).sum(-1)


def gaussian_kl(mu_old, sigma_old, mu, sigma):
    """KL(old || new) of diagonal Gaussians, summed over the action dimensions (rsl_rl PPO.update's adaptive-schedule statistic)."""
    return (torch.log(sigma / sigma_old + 1e-5) + (sigma_old * sigma_old + (mu_old - mu) ** 2) / (2.0 * sigma * sigma) - 0.5).sum(-1)


class PPO:
    """`PPO.update()` of rsl_rl on a `robot_lab_amd.rollout.RolloutStorage` that `Collector.collect()` has filled
    (observations, privileged_observations, actions, values, returns, advantages, actions_log_prob, mu, sigma: [T, N, ...] views of
    the HIP storage; advantages already normalised over the batch by `rl_rollout_compute_returns`)."""

    def __init__(self, policy: ActorCritic, value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01,
                 num_learning_epochs=5, num_mini_batches=4, learning_rate=1.0e-3, schedule="adaptive", desired_kl=0.01, max_grad_norm=1.0,
                 group=None):
        self.policy = policy
        self.group = group  # robot_lab_amd.dist.LearnerGroup of a multi-GPU run (rsl_rl's gradient all-reduce); None = single learner
        self.value_loss_coef, self.use_clipped_value_loss, self.clip_param, self.entropy_coef = value_loss_coef, use_clipped_value_loss, clip_param, entropy_coef
        self.num_learning_epochs, self.num_mini_batches = num_learning_epochs, num_mini_batches
        self.learning_rate, self.schedule, self.desired_kl, self.max_grad_norm = learning_rate, schedule, desired_kl, max_grad_norm
        self.optimizer = torch.optim.Adam(policy.parameters(), lr=learning_rate)

    def update(self, storage, generator: torch.Generator | None = None) -> dict:
        T, N = storage.num_transitions_per_env, storage.num_envs
        flat = lambda t: t.reshape(T * N, *t.shape[2:])  # noqa: E731
        obs, cobs, actions = flat(storage.observations), flat(storage.privileged_observations), flat(storage.actions)
        values, returns, adv = flat(storage.values).view(-1), flat(storage.returns).view(-1), flat(storage.advantages).view(-1)
        logp_old, mu_old, sigma_old = flat(storage.actions_log_prob).view(-1), flat(storage.mu), flat(storage.sigma)
        B = T * N
        mb = B // self.num_mini_batches
        stats = dict(value_loss=0.0, surrogate_loss=0.0, entropy=0.0, kl=0.0)
        n_updates = 0
        # rsl_rl's `mini_batch_generator` draws ONE permutation per update and walks it once per epoch
        perm = torch.randperm(B, device=obs.device, generator=generator)
        for _ in range(self.num_learning_epochs):
            for i in range(self.num_mini_batches):
                idx = perm[i * mb:(i + 1) * mb]
                mean, std = self.policy.distribution(obs[idx])
                logp = gaussian_log_prob(actions[idx], mean, std)
                value = self.policy.critic(cobs[idx]).view(-1)
                entropy = gaussian_entropy(std)
                if self.schedule == "adaptive" and self.desired_kl is not None:
                    with torch.inference_mode():
                        kl = gaussian_kl(mu_old[idx], sigma_old[idx], mean, std).mean()
                        if self.group is not None:
                            kl = self.group.mean(kl)  # the same statistic on every rank -> the same learning rate on every rank
                        if kl > 2.0 * self.desired_kl:
                            self.learning_rate = max(1e-5, self.learning_rate / 1.5)
                        elif 0.0 < kl < self.desired_kl / 2.0:
                            self.learning_rate = min(1e-2, self.learning_rate * 1.5)
                        for g in self.optimizer.param_groups:
                            g["lr"] = self.learning_rate
                        stats["kl"] += float(kl)
                ratio = torch.exp(logp - logp_old[idx])
                a = adv[idx]
                surrogate = torch.max(-a * ratio, -a * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
                if self.use_clipped_value_loss:
                    v_clipped = values[idx] + (value - values[idx]).clamp(-self.clip_param, self.clip_param)
                    value_loss = torch.max((value - returns[idx]) ** 2, (v_clipped - returns[idx]) ** 2).mean()
                else:
                    value_loss = ((returns[idx] - value) ** 2).mean()
                loss = surrogate + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean()
                self.optimizer.zero_grad(set_to_none=True)
                loss.backward()
                if self.group is not None:
                    self.group.reduce_gradients(self.policy)  # SUM / world, one flat all-reduce (rsl_rl `reduce_parameters`)
                nn.utils.clip_grad_norm_(self.policy.parameters(), self.max_grad_norm)
                self.optimizer.step()
                # a standard deviation pushed to (or through) zero would put NaN into the log-densities: the PARAMETER is floored, so that the
                # collector's sampling kernel, the stored sigma and this update all see the same distribution (rsl_rl 3.0.1 does not
                # clamp at all and fails with torch's Normal on a non-positive std; 1e-6 is far below any std training reaches)
                with torch.no_grad():
                    self.policy.std.clamp_(min=1e-6)
                stats["value_loss"] += float(value_loss.detach())
                stats["surrogate_loss"] += float(surrogate.detach())
                stats["entropy"] += float(entropy.detach().mean())
                n_updates += 1
        out = {k: v / max(n_updates, 1) for k, v in stats.items()}
        out["learning_rate"] = self.learning_rate
        return out


class Trainer:
    """collect (HIP, one graph launch) -> update (torch) -> push parameters, repeated: `OnPolicyRunner.learn` in miniature."""

    def __init__(self, env, num_steps_per_env=24, gamma=0.99, lam=0.95, seed=1, use_graph=True, actor_hidden=(512, 256, 128),
                 critic_hidden=(512, 256, 128), init_noise_std=1.0, clip_actions=None, group=None, **ppo_kw):
        from .collect import Collector
        from .policy import MlpPolicy
        from .rollout import RolloutStorage

        obs, _ = env.reset()
        od, cd, A = obs["policy"].shape[1], obs["critic"].shape[1], env.num_actions
        torch.manual_seed(seed)
        self.env, self.device = env, obs["policy"].device
        self.policy = ActorCritic(od, cd, A, tuple(actor_hidden), tuple(critic_hidden), init_noise_std).to(self.device)
        self.alg = PPO(self.policy, group=group, **ppo_kw)
        if group is not None:
            group.broadcast_parameters(self.policy)  # before the inference images are built from them
        lin = lambda m: [x for x in m if isinstance(x, nn.Linear)]  # noqa: E731
        host = lambda t: t.detach().cpu().numpy()  # noqa: E731
        self.actor = MlpPolicy([host(x.weight) for x in lin(self.policy.actor)], [host(x.bias) for x in lin(self.policy.actor)], "elu", device=str(self.device))
        self.critic = MlpPolicy([host(x.weight) for x in lin(self.policy.critic)], [host(x.bias) for x in lin(self.policy.critic)], "elu", device=str(self.device))
        self.storage = RolloutStorage(env.num_envs, num_steps_per_env, od, cd, A, seed=seed, device=str(self.device))
        self.std = self.policy.std.detach().clone()  # the tensor the sampling kernel reads: refreshed in place after every update
        self.collector = Collector(env, self.actor, self.critic, self.storage, self.std, gamma=gamma, lam=lam, use_graph=use_graph, clip_actions=clip_actions)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)
        self.iteration = 0

    def push_parameters(self):
        self.actor.load_linear_layers(self.policy.actor)
        self.critic.load_linear_layers(self.policy.critic)
        self.std.copy_(self.policy.std.detach().clamp_min(1e-6))

    def iterate(self) -> dict:
        self.collector.collect()
        st = self.storage
        out = dict(mean_reward=float(st.rewards.mean()), done_rate=float(st.dones.float().mean()))
        out.update(self.alg.update(st, self.gen))
        self.push_parameters()
        out["action_std"] = float(self.policy.std.detach().mean())
        self.iteration += 1
        return out
