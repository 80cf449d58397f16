#!/usr/bin/env python
"""The data-collection half of one PPO iteration as the reference runs it through `runner.learn(...)`
(scripts/reinforcement_learning/rsl_rl/train.py:224; rsl_rl OnPolicyRunner.learn): for each of `num_steps_per_env`
(= 24) steps  actions = alg.act(obs); obs, rewards, dones, extras = env.step(actions); alg.process_env_step(...),
then alg.compute_returns(obs) - every piece on the HIP kernels of this repo (actor + critic: rl_policy.hip, noise /
log-prob / storage / GAE: rl_rollout.hip, env: rl_env.hip).  Random network weights: the arithmetic does not
depend on them.
    python tools/bench_collect.py [task] [num_envs] [iterations]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robot_lab_amd.env import ManagerBasedRLEnv  # noqa: E402
from robot_lab_amd.policy import MlpPolicy  # noqa: E402
from robot_lab_amd.rollout import RolloutStorage  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ITERS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
T, GAMMA, LAM = 24, 0.99, 0.95  # rsl_rl_ppo_cfg.py:11,33-34
FUSED = os.environ.get("RL_FUSED_RECORD", "1") == "1"
PAIR = os.environ.get("RL_PAIR", "1") == "1"  # actor + critic in one launch (rl_mlp_forward_pair)
env = ManagerBasedRLEnv(task, num_envs=N, seed=42, device="cuda:0")
obs, _ = env.reset()
od, cd, A = obs["policy"].shape[1], obs["critic"].shape[1], env.num_actions
rng = np.random.default_rng(0)


def net(dims):
    ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
    return MlpPolicy(ws, [np.zeros(d, dtype=np.float32) for d in dims[1:]], "elu", device="cuda:0")


actor, critic = net([od, 512, 256, 128, A]), net([cd, 512, 256, 128, 1])
std = torch.ones(A, device="cuda:0")  # init_noise_std = 1.0 (rsl_rl_ppo_cfg.py:16)
storage = RolloutStorage(N, T, od, cd, A, seed=1, device="cuda:0")


GRAPH = os.environ.get("RL_GRAPH", "1") == "1"  # the whole iteration as one hipGraph launch (robot_lab_amd/collect.py)
SMALL = os.environ.get("RL_CRITIC_SMALL", "1") == "1"  # (overlap) the critic through rl_mlp_forward_small
FUSED_ACT = os.environ.get("RL_FUSED_ACT", "1") == "1"  # sampling / log-prob / the slot's first half in the actor + critic launch's epilogue (0: the act kernel)
OVERLAP = os.environ.get("RL_OVERLAP", "0") == "1"  # the critic of step t on a second stream under env step t (0: actor + critic as one launch in front of act)
if GRAPH and FUSED and PAIR:
    from robot_lab_amd.collect import Collector  # noqa: E402

    col = Collector(env, actor, critic, storage, std, GAMMA, LAM, use_graph=True, overlap=OVERLAP, critic_small=SMALL, fused_act=FUSED_ACT)
    iteration = lambda obs: col.collect()  # noqa: E731
else:

    def iteration(obs):
        storage.clear()
        for _ in range(T):
            mean, values = actor.forward_pair(obs["policy"], critic, obs["critic"]) if PAIR else (actor(obs["policy"]), critic(obs["critic"]))
            actions = storage.act(obs["policy"], obs["critic"], mean, std, values)
            if FUSED:  # the env kernel writes the transition's rewards / dones into the storage slot (rl_env_step_record)
                obs, rew, term, tout, extras = env.step(actions, rollout=storage, gamma=GAMMA)
            else:
                obs, rew, term, tout, extras = env.step(actions)
                storage.process_env_step(rew, term, tout, GAMMA)
        storage.compute_returns(critic(obs["critic"]), GAMMA, LAM)
        return obs


with torch.inference_mode():
    for _ in range(3):
        obs = iteration(obs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ITERS):
        obs = iteration(obs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
ok = bool(torch.isfinite(storage.advantages).all() and torch.isfinite(storage.returns).all())
print(f"{task} N={N} graph={int(GRAPH and FUSED and PAIR)} overlap={int(OVERLAP and GRAPH and FUSED and PAIR)} critic_small={int(SMALL)}: collection of {T} steps + GAE {1e3 * dt / ITERS:.3f} ms / iteration = {N * T * ITERS / dt / 1e6:.1f} M env-steps/s "
      f"({1e6 * dt / ITERS / T:.1f} us / step; finite: {ok}; adv mean {float(storage.advantages.mean()):+.2e} std {float(storage.advantages.std()):.4f})")
