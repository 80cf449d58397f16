"""Import shims that let the reference's task package (`robot_lab`) and launch scripts import
unchanged on a machine without IsaacLab / Isaac Sim / gymnasium (SURVEY.md section 7 step 0 and
section 8(b) "Entry": the registered entry point string is ``isaaclab.envs:ManagerBasedRLEnv``).

``install()`` puts this directory on ``sys.path`` (so ``isaaclab``, ``isaaclab_tasks``,
``isaaclab_rl``, ``gymnasium`` and ``toml`` resolve to the minimal packages here *only if the real
ones are absent*) and registers a fallback finder that fabricates permissive placeholder modules for
any other ``isaaclab.*`` / ``omni.*`` sub-module the reference imports but the MI355X env never uses
(viewer, USD converters, ...).  ``isaaclab.envs.ManagerBasedRLEnv`` is the HIP-backed environment.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
import types

_HERE = os.path.dirname(os.path.abspath(__file__))
_FAKE_ROOTS = ("isaaclab", "isaaclab_tasks", "isaaclab_rl", "isaaclab_assets", "omni", "isaacsim", "carb", "pxr")


class _Permissive(types.ModuleType):
    """Module whose unknown attributes are generic cfg classes (CamelCase) or named stub functions."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        from isaaclab.utils.configclass import GenericCfg, named_stub

        if name[:1].isupper() and not name.isupper():
            val = type(name, (GenericCfg,), {"__module__": self.__name__})
        elif name.isupper():
            val = ""  # path-like constants such as ISAAC_NUCLEUS_DIR
        else:
            val = named_stub(name, self.__name__)
        setattr(self, name, val)
        return val


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _FAKE_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        mod = _Permissive(spec.name)
        mod.__path__ = []
        return mod

    def exec_module(self, module):
        pass


_installed = False


def install(reference_source: str | None = None):
    """Make the shims importable; optionally also put the reference's `source/robot_lab` on the path."""
    global _installed
    if not _installed:
        have_real = importlib.util.find_spec("isaaclab") is not None and "shims" not in (
            importlib.util.find_spec("isaaclab").origin or "")
        if not have_real:
            sys.path.append(_HERE)
            sys.meta_path.append(_Finder())
        _installed = True
    if reference_source and reference_source not in sys.path and os.path.isdir(reference_source):
        sys.path.append(reference_source)


REFERENCE_SOURCE = "/root/reference/source/robot_lab"
