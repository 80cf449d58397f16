"""[UPSTREAM isaaclab.app] `AppLauncher` boots Omniverse Kit upstream
(`scripts/reinforcement_learning/rsl_rl/train.py:54`); on MI355X there is nothing to boot."""
import argparse


import os


class _App:
    """`simulation_app`: the reference's scripts loop `while simulation_app.is_running()` (zero_agent.py:63, play.py:242) until the
    viewer is closed.  Headless there is nobody to close it: RL_SHIM_MAX_STEPS=<n> ends the loop after n polls (tests, benchmarks);
    unset, it runs until interrupted, as upstream does headless."""

    def __init__(self):
        self._left = int(os.environ["RL_SHIM_MAX_STEPS"]) if os.environ.get("RL_SHIM_MAX_STEPS") else None

    def is_running(self):
        if self._left is None:
            return True
        self._left -= 1
        return self._left >= 0

    def close(self):
        pass


class AppLauncher:
    def __init__(self, launcher_args=None, **kwargs):
        self.app = _App()
        self.local_rank = 0
        self.global_rank = 0

    @staticmethod
    def add_app_launcher_args(parser: argparse.ArgumentParser):
        g = parser.add_argument_group("app_launcher arguments")
        g.add_argument("--headless", action="store_true", default=True)
        g.add_argument("--device", type=str, default="cuda:0")
        g.add_argument("--enable_cameras", action="store_true", default=False)
        g.add_argument("--livestream", type=int, default=-1)
        g.add_argument("--experience", type=str, default="")
        g.add_argument("--kit_args", type=str, default="")
