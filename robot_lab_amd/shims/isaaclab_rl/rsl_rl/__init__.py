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


class _PolicyExporter:
    """[UPSTREAM isaaclab_rl.rsl_rl.exporter] what `play.py:209-233` exports: normaliser followed by the actor MLP (the
    feed-forward policies of the reference's agent cfgs; recurrent policies are not used by any robot_lab task)."""

    @staticmethod
    def build(policy, normalizer=None):
        import copy

        import torch

        actor = getattr(policy, "actor", None)
        if actor is None:
            actor = getattr(policy, "student", None)
        if actor is None and isinstance(policy, torch.nn.Module):
            actor = policy
        if actor is None:
            raise TypeError("export_policy_as_*: the policy has no `actor` / `student` module")

        class Exported(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.normalizer = copy.deepcopy(normalizer) if normalizer is not None else torch.nn.Identity()
                self.actor = copy.deepcopy(actor)

            def forward(self, x):
                return self.actor(self.normalizer(x))

        return Exported().to("cpu").eval()


def export_policy_as_jit(policy, normalizer=None, path: str = ".", filename: str = "policy.pt"):
    """TorchScript export of the inference policy (`scripts/reinforcement_learning/rsl_rl/play.py:232`)."""
    import os

    import torch

    os.makedirs(path, exist_ok=True)
    module = _PolicyExporter.build(policy, normalizer)
    torch.jit.script(module).save(os.path.join(path, filename))


def export_policy_as_onnx(policy, normalizer=None, path: str = ".", filename: str = "policy.onnx", verbose: bool = False):
    """ONNX export of the inference policy (`play.py:233`); needs the `onnx` package, as upstream does."""
    import os

    import torch

    try:
        import onnx  # noqa: F401
    except ImportError as e:
        raise ImportError("export_policy_as_onnx needs the `onnx` package (torch.onnx.export); export_policy_as_jit does not") from e
    os.makedirs(path, exist_ok=True)
    module = _PolicyExporter.build(policy, normalizer)
    first = next(p for p in module.actor.parameters() if p.dim() == 2)
    torch.onnx.export(module, torch.zeros(1, first.shape[1]), os.path.join(path, filename), export_params=True, opset_version=11,
                      verbose=verbose, input_names=["obs"], output_names=["actions"], dynamic_axes={}, dynamo=False)


def __getattr__(name):
    if name.startswith("__") or not name[:1].isupper():
        raise AttributeError(name)
    val = type(name, (GenericCfg,), {"__module__": __name__})
    globals()[name] = val
    return val
