"""`cusrl.environment.isaaclab.TrainerCfg`: the dataclass base of the reference's `*TrainerCfg` classes (agents/cusrl_ppo_cfg.py)."""
from dataclasses import dataclass
from typing import Any


@dataclass
class TrainerCfg:
    max_iterations: int = 1000
    save_interval: int = 50
    experiment_name: str = ""
    agent_factory: Any = None
