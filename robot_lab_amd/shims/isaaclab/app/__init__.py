"""[UPSTREAM isaaclab.app] `AppLauncher` boots Omniverse Kit upstream
(`scripts/reinforcement_learning/rsl_rl/train.py:54`); on MI355X there is nothing to boot.  What it still owns is the rank of the
process: `train.py:143-150` reads `app_launcher.local_rank` for the device (`cuda:{local_rank}`) and the seed offset of a
`--distributed` run (`README.md:323-337`: `python -m torch.distributed.run --nproc_per_node=N ... train.py --distributed`)."""
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
    """[UPSTREAM] `launcher_args` is the script's parsed `argparse.Namespace` (or a dict), `kwargs` override it.  With `distributed`
    set - the flag the reference's train.py defines itself (train.py:33-35) and hands over inside the namespace - the ranks come from
    the environment `torch.distributed.run` exports (`LOCAL_RANK`, `RANK`; upstream also adds the JAX twins), otherwise the process
    is rank 0 of a world of one whatever the environment says, as upstream."""

    def __init__(self, launcher_args=None, **kwargs):
        args = dict(vars(launcher_args)) if isinstance(launcher_args, argparse.Namespace) else dict(launcher_args or {})
        args.update(kwargs)
        self.app = _App()
        self.local_rank = 0
        self.global_rank = 0
        self.device_id = 0
        if args.get("distributed"):
            self.local_rank = int(os.getenv("LOCAL_RANK", "0")) + int(os.getenv("JAX_LOCAL_RANK", "0"))
            self.global_rank = int(os.getenv("RANK", "0")) + int(os.getenv("JAX_RANK", "0"))
            self.device_id = self.local_rank
        else:
            dev = str(args.get("device") or "cuda:0")
            self.device_id = int(dev.split(":")[1]) if ":" in dev and dev.split(":")[1].isdigit() else 0

    @staticmethod
    def add_app_launcher_args(parser: argparse.ArgumentParser):
        g = parser.add_argument_group("app_launcher arguments")
        g.add_argument("--headless", action="store_true", default=True)
        g.add_argument("--device", type=str, default="cuda:0")
        g.add_argument("--enable_cameras", action="store_true", default=False)
        g.add_argument("--livestream", type=int, default=-1)
        g.add_argument("--experience", type=str, default="")
        g.add_argument("--kit_args", type=str, default="")
