"""[UPSTREAM isaaclab.managers] cfg containers + the two base classes the reference subclasses
(`ManagerTermBase` for `GaitReward`, `VEL/mdp/rewards.py:156`; `CommandTerm` for
`DiscreteCommandController`, `VEL/mdp/commands.py:100`)."""
from __future__ import annotations

from dataclasses import MISSING

from isaaclab.utils.configclass import GenericCfg, configclass


@configclass
class SceneEntityCfg:
    name: str = MISSING
    joint_names = None
    joint_ids = slice(None)
    fixed_tendon_names = None
    fixed_tendon_ids = slice(None)
    body_names = None
    body_ids = slice(None)
    object_collection_names = None
    object_collection_ids = slice(None)
    preserve_order: bool = False


@configclass
class ManagerTermBaseCfg:
    func = MISSING
    params: dict = {}


@configclass
class RewardTermCfg(ManagerTermBaseCfg):
    weight: float = MISSING


@configclass
class TerminationTermCfg(ManagerTermBaseCfg):
    time_out: bool = False


@configclass
class CurriculumTermCfg(ManagerTermBaseCfg):
    pass


@configclass
class EventTermCfg(ManagerTermBaseCfg):
    mode: str = MISSING
    interval_range_s = None
    is_global_time: bool = False
    min_step_count_between_reset: int = 0


@configclass
class ObservationTermCfg(ManagerTermBaseCfg):
    modifiers = None
    noise = None
    clip = None
    scale = None
    history_length: int = 0
    flatten_history_dim: bool = True


@configclass
class ObservationGroupCfg:
    concatenate_terms: bool = True
    concatenate_dim: int = -1
    enable_corruption: bool = False
    history_length = None
    flatten_history_dim: bool = True


@configclass
class ActionTermCfg:
    class_type = None
    asset_name: str = MISSING
    debug_vis: bool = False
    clip = None


@configclass
class CommandTermCfg:
    class_type = None
    resampling_time_range = MISSING
    debug_vis: bool = False


class ManagerTermBase:
    def __init__(self, cfg, env):
        self.cfg = cfg
        self._env = env

    @property
    def num_envs(self):
        return self._env.num_envs

    @property
    def device(self):
        return self._env.device

    def reset(self, env_ids=None):
        pass


class CommandTerm(ManagerTermBase):
    """[UPSTREAM B7] compute(dt): metrics -> time_left -= dt -> resample -> update."""

    def __init__(self, cfg, env):
        import torch

        super().__init__(cfg, env)
        self.metrics = {}
        self.time_left = torch.zeros(self.num_envs, device=self.device)
        self.command_counter = torch.zeros(self.num_envs, device=self.device, dtype=torch.long)

    def reset(self, env_ids=None):
        import torch

        if env_ids is None:
            env_ids = slice(None)
        extras = {}
        for name, value in self.metrics.items():
            extras[name] = torch.mean(value[env_ids]).item()
            value[env_ids] = 0.0
        self.command_counter[env_ids] = 0
        self._resample(env_ids)
        return extras

    def compute(self, dt):
        self._update_metrics()
        self.time_left -= dt
        ids = (self.time_left <= 0.0).nonzero().flatten()
        if len(ids) > 0:
            self._resample(ids)
        self._update_command()

    def _resample(self, env_ids):
        if len(env_ids) != 0:
            self.time_left[env_ids] = self.time_left[env_ids].uniform_(*self.cfg.resampling_time_range)
            self._resample_command(env_ids)
            self.command_counter[env_ids] += 1


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    if name[:1].isupper():
        val = type(name, (GenericCfg,), {"__module__": __name__})
        globals()[name] = val
        return val
    raise AttributeError(name)
