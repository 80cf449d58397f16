"""[UPSTREAM isaaclab_tasks.utils.parse_cfg]"""
from __future__ import annotations

import importlib
import os
import re


def load_cfg_from_registry(task_name: str, entry_point_key: str):
    import gymnasium as gym

    spec = gym.spec(task_name.split(":")[-1])
    ep = spec.kwargs.get(entry_point_key)
    if ep is None:
        raise ValueError(f"Could not find configuration for the environment: '{task_name}' (key {entry_point_key}).")
    if isinstance(ep, str):
        mod_name, attr = ep.split(":")
        cls = getattr(importlib.import_module(mod_name), attr)
    else:
        cls = ep
    return cls() if callable(cls) else cls


def parse_env_cfg(task_name: str, device: str = "cuda:0", num_envs: int | None = None, use_fabric: bool | None = None):
    cfg = load_cfg_from_registry(task_name, "env_cfg_entry_point")
    cfg.sim.device = device
    if num_envs is not None:
        cfg.scene.num_envs = num_envs
    return cfg


def get_checkpoint_path(log_path: str, run_dir: str = ".*", checkpoint: str = ".*", other_dirs=None, sort_alpha: bool = True) -> str:
    runs = sorted(d for d in os.listdir(log_path) if os.path.isdir(os.path.join(log_path, d)) and re.match(run_dir, d))
    if not runs:
        raise ValueError(f"No runs present in the directory: '{log_path}' match: '{run_dir}'.")
    run_path = os.path.join(log_path, runs[-1], *(other_dirs or []))
    files = [f for f in os.listdir(run_path) if re.match(checkpoint, f)]
    if not files:
        raise ValueError(f"No checkpoints in the directory: '{run_path}' match '{checkpoint}'.")
    files.sort(key=lambda m: f"{m:0>15}")
    return os.path.join(run_path, files[-1])
