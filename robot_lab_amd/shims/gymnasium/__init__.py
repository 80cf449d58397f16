"""Minimal `gymnasium` stand-in (not installed in this image, no network): the registry calls the
reference makes (`gym.register`, `gym.make`, `gym.spec`; `VEL/config/**/__init__.py`,
`scripts/tools/zero_agent.py:56`) and `spaces.Box` / `spaces.Dict` for `observation_space`."""
from __future__ import annotations

import importlib
from dataclasses import dataclass, field

from . import spaces  # noqa: F401


@dataclass
class EnvSpec:
    id: str
    entry_point: object = None
    kwargs: dict = field(default_factory=dict)
    disable_env_checker: bool = True


class _Registry(dict):
    pass


registry = _Registry()


def register(id, entry_point=None, kwargs=None, disable_env_checker=True, **_):
    registry[id] = EnvSpec(id, entry_point, dict(kwargs or {}), disable_env_checker)


def spec(id):
    if id not in registry:
        raise KeyError(f"No registered env with id: {id}")
    return registry[id]


class Env:
    metadata: dict = {}
    observation_space = None
    action_space = None
    spec = None

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped


def make(id, **kwargs):
    s = spec(id)
    ep = s.entry_point
    if isinstance(ep, str):
        mod, attr = ep.split(":")
        ep = getattr(importlib.import_module(mod), attr)
    kw = {k: v for k, v in s.kwargs.items() if not k.endswith("_entry_point")}
    kw.update(kwargs)
    env = ep(**kw)
    try:
        env.spec = s
    except AttributeError:
        pass
    return env


class _Vector:
    class VectorEnv(Env):
        pass


vector = _Vector()
