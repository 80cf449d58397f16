"""The data-collection half of a PPO iteration as ONE hipGraph launch.

What the reference runs through `runner.learn(...)` (scripts/reinforcement_learning/rsl_rl/train.py:224 -> rsl_rl
`OnPolicyRunner.learn`): per iteration `num_steps_per_env` (= 24, .../unitree_a1/agents/rsl_rl_ppo_cfg.py:11) times

    actions = alg.act(obs);  obs, rewards, dones, extras = env.step(actions);  alg.process_env_step(...)

then `alg.compute_returns(obs)`.  Here every piece is a HIP kernel of this repo (actor + critic: csrc/rl_policy.hip; noise,
log-prob, storage, GAE: csrc/rl_rollout.hip; env: csrc/rl_env.hip, which also writes the transition's rewards / dones into
the storage slot), 2 x 24 + 4 launches per iteration.  The only launch arguments that change from one iteration to the next -
the env's step count and the storage's random counter - are read by the kernels from device words (`rl_env_graph_*`,
`rl_rollout_graph_*`), so the whole iteration is stream-captured once and replayed: no Python, ctypes or launch latency between
the kernels.  Host plumbing only: torch provides the capture (`torch.cuda.CUDAGraph`) and the stream.
"""
from __future__ import annotations

import torch


class Collector:
    def __init__(self, env, actor, critic, storage, action_std: torch.Tensor, gamma: float = 0.99, lam: float = 0.95,
                 normalize_advantage: bool = True, use_graph: bool = True, clip_actions: float | None = None):
        self.env, self.actor, self.critic, self.storage = env, actor, critic, storage
        self.std, self.gamma, self.lam, self.normalize = action_std, gamma, lam, normalize_advantage
        self.T = storage.num_transitions_per_env
        self.clip_actions = clip_actions  # RslRlVecEnvWrapper.step clamps what the env sees; the storage keeps the sampled action
        if self.T % 2:
            use_graph = False  # the env's observation buffers alternate: a captured loop needs an even number of steps
        self.use_graph = use_graph
        self._graph, self._warm = None, False
        self.obs = env.get_observations()

    def _iteration(self, obs):
        st, env = self.storage, self.env
        st.clear()
        for _ in range(self.T):
            mean, values = self.actor.forward_pair(obs["policy"], self.critic, obs["critic"])
            actions = st.act(obs["policy"], obs["critic"], mean, self.std, values)
            if self.clip_actions is not None:
                actions = actions.clamp(-self.clip_actions, self.clip_actions)
            obs, _, _, _, _ = env.step(actions, rollout=st, gamma=self.gamma)
        st.compute_returns(self.critic(obs["critic"]), self.gamma, self.lam, self.normalize)
        return obs

    def _capture(self):
        env, st = self.env, self.storage
        g = torch.cuda.CUDAGraph()
        env.graph_begin()
        st.graph_begin()
        start = self.obs
        with torch.cuda.graph(g):
            end = self._iteration(start)
            n_env, n_ro = env.graph_end(), st.graph_end()
        assert n_env == self.T and n_ro == self.T, (n_env, n_ro)
        # an even number of steps: the loop ends in the observation buffers it started from, so the replay consumes its own output
        assert end["policy"].data_ptr() == start["policy"].data_ptr()
        self._graph = g

    @torch.inference_mode()
    def collect(self):
        """One iteration: fills the storage (observations ... advantages) and returns the observations of the next one."""
        if not self.use_graph:
            self.obs = self._iteration(self.obs)
            return self.obs
        if not self._warm:  # eager once: every library has its per-device setup behind it
            self._warm = True
            self.obs = self._iteration(self.obs)
            return self.obs
        if self._graph is None:
            self._capture()
        self.env.graph_launching(self.T)
        self.storage.graph_launching()
        self._graph.replay()
        self.obs = self.env.get_observations()
        return self.obs
