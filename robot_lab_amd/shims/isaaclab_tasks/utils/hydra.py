"""[UPSTREAM isaaclab_tasks.utils.hydra] `hydra_task_config` decorator (`train.py:118`).  Hydra is not
installed here; `key=value` CLI overrides of the form `env.a.b=1` / `agent.x=2` are applied directly."""
from __future__ import annotations

import ast
import functools
import sys

from .parse_cfg import load_cfg_from_registry


def _apply(cfg, dotted, value):
    obj = cfg
    parts = dotted.split(".")
    for p in parts[:-1]:
        obj = obj[p] if isinstance(obj, dict) else getattr(obj, p)
    try:
        value = ast.literal_eval(value)
    except (ValueError, SyntaxError):
        pass
    if isinstance(obj, dict):
        obj[parts[-1]] = value
    else:
        setattr(obj, parts[-1], value)


def hydra_task_config(task_name: str, agent_cfg_entry_point: str):
    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            env_cfg = load_cfg_from_registry(task_name, "env_cfg_entry_point")
            agent_cfg = load_cfg_from_registry(task_name, agent_cfg_entry_point) if agent_cfg_entry_point else None
            for a in sys.argv[1:]:
                if "=" in a and not a.startswith("-"):
                    k, v = a.split("=", 1)
                    if k.startswith("env."):
                        _apply(env_cfg, k[4:], v)
                    elif k.startswith("agent.") and agent_cfg is not None:
                        _apply(agent_cfg, k[6:], v)
            return func(env_cfg, agent_cfg, *args, **kwargs)

        return wrapper

    return decorator
