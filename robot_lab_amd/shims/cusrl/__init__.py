"""Stand-in for the third-party `cusrl` RL library (absent here, like rsl_rl): the reference's `agents/cusrl_ppo_cfg.py` files build
their trainer configuration out of `cusrl.*.Factory(...)` / `cusrl.hook.*(...)` objects at IMPORT time, and
`config/quadruped/agibot_d1/agents/__init__.py:4` imports that file unconditionally - without this package the D1 task ids never
register.  Nothing here trains: every attribute is a recorder that remembers its dotted name and the arguments it was called with,
so a cfg object can still be inspected (`cfg.agent_factory`), and `cusrl.environment.isaaclab.TrainerCfg` is a plain dataclass base.
Used only when the real package is not installed (robot_lab_amd.shims.install() appends this directory to sys.path)."""
from __future__ import annotations


class Recorded:
    """`cusrl.Actor.Factory(backbone_factory=...)` -> Recorded('cusrl.Actor.Factory', kwargs={...})"""

    def __init__(self, name, args=(), kwargs=None):
        self._name, self._args, self._kwargs = name, tuple(args), dict(kwargs or {})

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return Recorded(f"{self._name}.{item}")

    def __call__(self, *args, **kwargs):
        return Recorded(self._name, args, kwargs)

    def __repr__(self):
        inner = ", ".join([repr(a) for a in self._args] + [f"{k}={v!r}" for k, v in self._kwargs.items()])
        return f"{self._name}({inner})" if (self._args or self._kwargs) else self._name


def __getattr__(name):  # PEP 562: cusrl.ActorCritic, cusrl.hook, cusrl.Mlp, ...
    if name.startswith("__"):
        raise AttributeError(name)
    return Recorded(f"cusrl.{name}")
