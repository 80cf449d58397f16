"""[UPSTREAM isaaclab_rl.rsl_rl] agent cfg containers (`.../unitree_a1/agents/rsl_rl_ppo_cfg.py:6`) and
`RslRlVecEnvWrapper` (`train.py:100,202`)."""
from __future__ import annotations

from dataclasses import MISSING

import torch

from isaaclab.utils.configclass import GenericCfg, configclass


@configclass
class RslRlPpoActorCriticCfg:
    class_name: str = "ActorCritic"
    init_noise_std: float = MISSING
    noise_std_type: str = "scalar"
    actor_obs_normalization: bool = False
    critic_obs_normalization: bool = False
    actor_hidden_dims: list = MISSING
    critic_hidden_dims: list = MISSING
    activation: str = MISSING


@configclass
class RslRlPpoAlgorithmCfg:
    class_name: str = "PPO"
    num_learning_epochs: int = MISSING
    num_mini_batches: int = MISSING
    learning_rate: float = MISSING
    schedule: str = MISSING
    gamma: float = MISSING
    lam: float = MISSING
    entropy_coef: float = MISSING
    desired_kl: float = MISSING
    max_grad_norm: float = MISSING
    value_loss_coef: float = MISSING
    use_clipped_value_loss: bool = MISSING
    clip_param: float = MISSING
    normalize_advantage_per_mini_batch: bool = False
    symmetry_cfg = None
    rnd_cfg = None


@configclass
class RslRlBaseRunnerCfg:
    seed: int = 42
    device: str = "cuda:0"
    num_steps_per_env: int = MISSING
    max_iterations: int = MISSING
    empirical_normalization = None
    obs_groups: dict = {"policy": ["policy"], "critic": ["critic"]}
    clip_actions = None
    save_interval: int = MISSING
    experiment_name: str = MISSING
    run_name: str = ""
    logger: str = "tensorboard"
    neptune_project: str = "isaaclab"
    wandb_project: str = "isaaclab"
    resume: bool = False
    load_run: str = ".*"
    load_checkpoint: str = "model_.*.pt"


@configclass
class RslRlOnPolicyRunnerCfg(RslRlBaseRunnerCfg):
    class_name: str = "OnPolicyRunner"
    policy = MISSING
    algorithm = MISSING


def handle_deprecated_rsl_rl_cfg(agent_cfg, installed_version=None):
    return agent_cfg


class RslRlVecEnvWrapper:
    """[UPSTREAM B10] clamps actions, steps, returns (obs, rew, dones, extras + time_outs)."""

    def __init__(self, env, clip_actions: float | None = None):
        self.env = env
        self.clip_actions = clip_actions
        u = self.unwrapped
        self.num_envs = u.num_envs
        self.device = u.device
        self.max_episode_length = u.max_episode_length
        self.num_actions = u.num_actions
        self.env.reset()

    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def cfg(self):
        return self.unwrapped.cfg

    @property
    def episode_length_buf(self):
        return self.unwrapped.episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, value):
        self.unwrapped.episode_length_buf = value

    @property
    def observation_space(self):
        return self.env.observation_space

    @property
    def action_space(self):
        return self.env.action_space

    def seed(self, seed: int = -1) -> int:
        return self.unwrapped.seed(seed)

    def get_observations(self):
        return self._as_tensordict(self.unwrapped.get_observations())

    def reset(self):
        obs, extras = self.env.reset()
        return self._as_tensordict(obs), extras

    def step(self, actions):
        if self.clip_actions is not None:
            actions = torch.clamp(actions, -self.clip_actions, self.clip_actions)
        obs, rew, terminated, truncated, extras = self.env.step(actions)
        dones = (terminated | truncated).to(dtype=torch.long)
        if not getattr(self.unwrapped.cfg, "is_finite_horizon", False):
            extras["time_outs"] = truncated
        return self._as_tensordict(obs), rew, dones, extras

    def close(self):
        return self.env.close()

    def _as_tensordict(self, obs):
        try:
            from tensordict import TensorDict

            return TensorDict(obs, batch_size=[self.num_envs])
        except ImportError:
            return obs


def __getattr__(name):
    if name.startswith("__") or not name[:1].isupper():
        raise AttributeError(name)
    val = type(name, (GenericCfg,), {"__module__": __name__})
    globals()[name] = val
    return val
