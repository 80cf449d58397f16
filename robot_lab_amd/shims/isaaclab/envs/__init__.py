"""`isaaclab.envs` as seen by the reference: the env cfg base class and - the drop-in boundary of
SURVEY.md section 8(b) - `ManagerBasedRLEnv`, which here is the HIP-backed MI355X environment
(`robot_lab_amd.env.ManagerBasedRLEnv`).  `gym.register(entry_point="isaaclab.envs:ManagerBasedRLEnv")`
in `VEL/config/quadruped/unitree_a1/__init__.py:12-32` therefore resolves to it unchanged."""
from __future__ import annotations

from dataclasses import MISSING

from isaaclab.utils.configclass import GenericCfg, configclass


@configclass
class PhysxCfg:
    gpu_max_rigid_patch_count: int = 5 * 2**15
    solver_type: int = 1


@configclass
class SimulationCfg:
    dt: float = 1.0 / 60.0
    render_interval: int = 1
    gravity = (0.0, 0.0, -9.81)
    device: str = "cuda:0"
    physics_material = None
    physx = PhysxCfg()


@configclass
class ViewerCfg:
    eye = (7.5, 7.5, 7.5)
    lookat = (0.0, 0.0, 0.0)
    origin_type: str = "world"
    env_index: int = 0
    asset_name = None


@configclass
class ManagerBasedEnvCfg:
    viewer = ViewerCfg()
    sim = SimulationCfg()
    seed = None
    decimation: int = MISSING
    scene = MISSING
    observations = MISSING
    actions = MISSING
    events = None
    recorders = None
    rerender_on_reset: bool = False
    wait_for_textures: bool = True
    xr = None


@configclass
class ManagerBasedRLEnvCfg(ManagerBasedEnvCfg):
    is_finite_horizon: bool = False
    episode_length_s: float = MISSING
    rewards = MISSING
    terminations = MISSING
    curriculum = None
    commands = None


@configclass
class DirectRLEnvCfg:
    pass


@configclass
class DirectMARLEnvCfg:
    pass


class DirectRLEnv:  # other reference tasks subclass these; never instantiated here
    pass


class DirectMARLEnv:
    pass


class ManagerBasedEnv:
    pass


def multi_agent_to_single_agent(env):
    return env


def __getattr__(name):
    if name == "ManagerBasedRLEnv":
        from robot_lab_amd.env import ManagerBasedRLEnv

        return ManagerBasedRLEnv
    if name == "mdp":
        import importlib

        return importlib.import_module("isaaclab.envs.mdp")
    if name.startswith("__"):
        raise AttributeError(name)
    if name[:1].isupper():
        val = type(name, (GenericCfg,), {"__module__": __name__})
        globals()[name] = val
        return val
    raise AttributeError(name)
