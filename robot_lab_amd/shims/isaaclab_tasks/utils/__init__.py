"""[UPSTREAM isaaclab_tasks.utils] registry helpers used by the reference's launch scripts
(`scripts/tools/zero_agent.py:43,52-54`, `scripts/reinforcement_learning/rsl_rl/train.py:102-103`)."""
from __future__ import annotations

import importlib
import os
import pkgutil
import re
import sys

from .parse_cfg import get_checkpoint_path, load_cfg_from_registry, parse_env_cfg  # noqa: F401


def import_packages(package_name: str, blacklist_pkgs: list | None = None):
    """Import every sub-package so that their `gym.register` calls run (`robot_lab/tasks/__init__.py:20-21`).
    Task families whose third-party dependencies are absent are skipped with a note instead of aborting."""
    blacklist_pkgs = blacklist_pkgs or []
    package = importlib.import_module(package_name)
    for _, name, ispkg in pkgutil.walk_packages(package.__path__, package.__name__ + ".", onerror=lambda n: None):
        if any(b in name for b in blacklist_pkgs):
            continue
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
        except (Exception, SystemExit) as e:  # noqa: BLE001 - optional task families (skrl/AMP/beyondmimic deps, argparse scripts)
            if os.environ.get("ROBOT_LAB_AMD_VERBOSE_IMPORT"):
                print(f"[robot_lab_amd] skipped {name}: {type(e).__name__}: {e}")
