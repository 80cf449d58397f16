"""TEST INFRASTRUCTURE ONLY (oracle).  fp64 numpy restatement of the on-policy rollout arithmetic of rsl-rl-lib 3.0.1
(third-party, pinned by /root/reference/scripts/reinforcement_learning/rsl_rl/train.py:65-75, absent from the
reference tree - PARITY UNPINNED against the library itself; anchored on the reference's call sites
train.py:206-224 and .../unitree_a1/agents/rsl_rl_ppo_cfg.py:11,16,33-34 and on the published algorithm):

  PPO.act / process_env_step  (rsl_rl/algorithms/ppo.py)
  RolloutStorage.add_transitions / compute_returns  (rsl_rl/storage/rollout_storage.py)

The action noise is this project's own stream (Philox block -> Box-Muller), identical uniforms to the kernel
(oracle/philox.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import numpy as np

from . import philox as px

STREAM_POLICY = 7


def standard_normal(seed, n_envs, counter, act_dim):
    """eps[n_envs, act_dim]: Philox block b = j >> 2 of (env, counter, STREAM_POLICY) gives uniforms u0..u3;
    (u0, u1) -> actions 4b, 4b+1 and (u2, u3) -> 4b+2, 4b+3 by Box-Muller on (1 - u_even, u_odd)."""
    env = np.arange(n_envs, dtype=np.uint32)[:, None]
    j = np.arange(act_dim, dtype=np.uint32)[None]
    pair = (j >> np.uint32(1)) << np.uint32(1)  # index of the pair's first uniform
    u1 = 1.0 - px.uniform(seed, env, counter, STREAM_POLICY, pair)
    u2 = px.uniform(seed, env, counter, STREAM_POLICY, pair + np.uint32(1))
    r = np.sqrt(-2.0 * np.log(u1))
    th = 2.0 * np.pi * u2
    return np.where(j & np.uint32(1), r * np.sin(th), r * np.cos(th))


class RolloutOracle:
    def __init__(self, num_envs, num_steps, obs_dim, critic_dim, act_dim, seed):
        self.N, self.T, self.A, self.seed = num_envs, num_steps, act_dim, seed
        self.obs = np.zeros((num_steps, num_envs, obs_dim))
        self.critic_obs = np.zeros((num_steps, num_envs, critic_dim))
        self.actions = np.zeros((num_steps, num_envs, act_dim))
        self.mu = np.zeros((num_steps, num_envs, act_dim))
        self.sigma = np.zeros((num_steps, num_envs, act_dim))
        self.log_prob = np.zeros((num_steps, num_envs))
        self.values = np.zeros((num_steps, num_envs))
        self.rewards = np.zeros((num_steps, num_envs))
        self.dones = np.zeros((num_steps, num_envs), dtype=bool)
        self.returns = np.zeros((num_steps, num_envs))
        self.advantages = np.zeros((num_steps, num_envs))
        self.step, self.counter = 0, 0

    def act(self, obs, critic_obs, mean, std, values):
        """ppo.py `act`: sample, log-prob, stash the transition's first half."""
        t = self.step
        if t >= self.T:
            raise OverflowError("Rollout buffer overflow")  # rollout_storage.py add_transitions
        mean = np.asarray(mean, dtype=np.float64)
        std = np.broadcast_to(np.asarray(std, dtype=np.float64), mean.shape)
        a = mean + std * standard_normal(self.seed, self.N, self.counter, self.A)
        self.obs[t], self.critic_obs[t] = obs, critic_obs
        self.actions[t], self.mu[t], self.sigma[t] = a, mean, std
        self.log_prob[t] = np.sum(-0.5 * ((a - mean) / std) ** 2 - np.log(std) - 0.5 * np.log(2.0 * np.pi), axis=1)
        self.values[t] = values
        return a

    def record(self, rewards, terminated, time_outs, gamma):
        """ppo.py `process_env_step`: bootstrapping on time outs, then add_transitions."""
        t = self.step
        to = np.asarray(time_outs).astype(bool)
        self.rewards[t] = np.asarray(rewards, dtype=np.float64) + gamma * self.values[t] * to
        self.dones[t] = np.asarray(terminated).astype(bool) | to
        self.step += 1
        self.counter += 1

    def compute_returns(self, last_values, gamma, lam, normalize_advantage=True):
        """rollout_storage.py `compute_returns`."""
        adv = 0.0
        for t in reversed(range(self.T)):
            nxt = np.asarray(last_values, dtype=np.float64) if t == self.T - 1 else self.values[t + 1]
            nt = 1.0 - self.dones[t]
            delta = self.rewards[t] + nt * gamma * nxt - self.values[t]
            adv = delta + nt * gamma * lam * adv
            self.returns[t] = adv + self.values[t]
        self.advantages = self.returns - self.values
        if normalize_advantage:
            self.advantages = (self.advantages - self.advantages.mean()) / (self.advantages.std(ddof=1) + 1e-8)

    def clear(self):
        self.step = 0
