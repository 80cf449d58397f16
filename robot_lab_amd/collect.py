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
    """`overlap=True`: the actor gates the env step, the critic does not - V(s_t) is needed by the storage, the time-out
    bootstrap and GAE only.  So per step the main stream runs  actor -> act (sampling, log-prob, storage) -> env step  and a second
    stream runs the critic of step t (privileged observations into the slot, V straight into the values slot) UNDER env step t; the
    env kernel records the raw reward and marks time outs (rl_env_step_record with values NULL), `compute_returns` bootstraps them.
    Ordering: critic t reads the env's observation buffer t % 2, which env step t + 1 overwrites - the main stream waits for the
    event of critic t - 1 before env step t; everything joins before the last critic call and GAE.  Captured as one hipGraph like
    the serial form (cross-stream events become graph edges).  `overlap=False` (default): actor + critic as one launch in front of act.
    Measured, one call (profiles/r06b_collect.txt): A1 Rough 4096 envs 84.7 us / step serial; 107.0 overlapped with the critic through its
    production kernel - its workgroups (512 - 1024 threads, > 83 KB of LDS) do not fit on a CU beside the env kernel's workgroup (one
    wavefront per SIMD at 312 registers, 78 KB of LDS), so nothing overlaps and the pair launch's sharing of the chip between actor and
    critic is lost (actor alone 22.3 us + critic alone 27.6 us against 36.6 us for the pair); 126.1 with the critic through the
    small-footprint launch that DOES fit (`critic_small`, rl_mlp_forward_small: four wavefronts, 140 registers, 64 KB of LDS) - co-resident
    MFMA wavefronts take issue slots from an env wavefront that is latency-bound with its SIMD to itself, and the env step slows by more
    than the critic's time.  G1 2048: 138.6 / 140.6 / 153.1.  So the serial loop stays the default.  Same numbers bit for bit (tests/test_gpu_collect.py)."""

    def __init__(self, env, actor, critic, storage, action_std: torch.Tensor, gamma: float = 0.99, lam: float = 0.95,
                 normalize_advantage: bool = True, use_graph: bool = True, clip_actions: float | None = None, overlap: bool = False, critic_small: bool = True,
                 fused_act: bool = True):
        self.env, self.actor, self.critic, self.storage = env, actor, critic, storage
        self.overlap = overlap
        # sampling / log-prob / the slot's first half in the epilogue of the actor + critic launch (include/rl_act.h): one launch and one gap
        # less per step than actor + critic, then `act` (second half); same storage bit for bit (tests/test_gpu_collect.py)
        self.fused_act = fused_act
        self.critic_small = critic_small  # (overlap) the critic through the small-footprint launch that fits beside the env kernel's workgroups
        self._side = torch.cuda.Stream(device=env.device) if overlap else None
        self.std, self.gamma, self.lam, self.normalize = action_std, gamma, lam, normalize_advantage
        self.T = storage.num_transitions_per_env
        self.clip_actions = clip_actions  # RslRlVecEnvWrapper.step clamps what the env sees; the storage keeps the sampled action
        if self.T % 2:
            use_graph = False  # the env's observation buffers alternate: a captured loop needs an even number of steps
        self.use_graph = use_graph
        self._graph, self._warm = None, False
        self.obs = env.get_observations()

    def _iteration(self, obs):
        if self.overlap:
            return self._iteration_overlapped(obs)
        st, env = self.storage, self.env
        st.clear()
        for _ in range(self.T):
            if self.fused_act:
                actions = st.act_fused(self.actor, self.critic, obs["policy"], obs["critic"], self.std, self.clip_actions)
            else:
                mean, values = self.actor.forward_pair(obs["policy"], self.critic, obs["critic"])
                actions = st.act(obs["policy"], obs["critic"], mean, self.std, values)
                if self.clip_actions is not None:
                    actions = actions.clamp(-self.clip_actions, self.clip_actions)
            obs, _, _, _, _ = env.step(actions, rollout=st, gamma=self.gamma)
        st.compute_returns(self.critic(obs["critic"]), self.gamma, self.lam, self.normalize)
        return obs

    def _iteration_overlapped(self, obs):
        st, env, side = self.storage, self.env, self._side
        main = torch.cuda.current_stream(env.device)
        st.clear()
        critic_done = None  # event of the critic of the previous step
        for _ in range(self.T):
            side.wait_stream(main)  # the observations of this step (env step t - 1) and everything before it
            with torch.cuda.stream(side):
                st.critic_half(self.critic, obs["critic"], small=self.critic_small)
                ev = torch.cuda.Event()
                ev.record(side)
            mean = self.actor(obs["policy"])
            actions = st.act(obs["policy"], None, mean, self.std, None)
            if self.clip_actions is not None:
                actions = actions.clamp(-self.clip_actions, self.clip_actions)
            if critic_done is not None:
                main.wait_event(critic_done)  # env step t writes the observation buffer the critic of step t - 1 read
            obs, _, _, _, _ = env.step(actions, rollout=st, gamma=self.gamma, defer_bootstrap=True)
            critic_done = ev
        main.wait_stream(side)
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
