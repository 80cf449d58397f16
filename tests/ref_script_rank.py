"""Test infrastructure: ONE rank of `python -m torch.distributed.run --nproc_per_node=N <this file> <reference script> <its args>`.

Runs the reference's script as a file (runpy) through `robot_lab_amd.shims`, exactly as `tests/test_reference_scripts.py` does in-process,
and writes what this rank ended up with to `$RL_TEST_OUT/rank<r>.json`:

* what the script handed to the drop-in boundary - `env_cfg.sim.device`, `env_cfg.seed` - and the launcher's ranks
  (`scripts/reinforcement_learning/rsl_rl/train.py:143-150`), recorded at `ManagerBasedRLEnv.__init__`;
* without a HIP device (the CPU tier) the run stops there: the env has no CPU path;
* with one, the script trains on and the record also carries a checksum of the learner's parameters after `learn()` - equal on every
  rank iff the gradient all-reduce tied the learners together.
"""
import hashlib
import json
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


class _StopAtTheBoundary(Exception):
    pass


def main():
    sys.path.insert(0, ROOT)
    script, argv = sys.argv[1], sys.argv[2:]
    out_dir = os.environ["RL_TEST_OUT"]
    rank = int(os.environ.get("RANK", "0"))
    record = {"rank": rank, "local_rank_env": int(os.environ.get("LOCAL_RANK", "0")), "world": int(os.environ.get("WORLD_SIZE", "1"))}

    import torch

    from robot_lab_amd import shims

    shims.install(shims.REFERENCE_SOURCE)
    import robot_lab_amd.env as env_mod

    real_init = env_mod.ManagerBasedRLEnv.__init__

    def recording_init(self, cfg=None, *a, **k):
        main_mod = sys.modules["__main__"]
        launcher = getattr(main_mod, "app_launcher", None)
        record.update(sim_device=str(cfg.sim.device), env_seed=int(cfg.seed), num_envs=int(cfg.scene.num_envs),
                      launcher_local_rank=getattr(launcher, "local_rank", None), launcher_global_rank=getattr(launcher, "global_rank", None))
        if not torch.cuda.is_available():
            raise _StopAtTheBoundary()
        real_init(self, cfg, *a, **k)
        record["env_device"] = self.device

    env_mod.ManagerBasedRLEnv.__init__ = recording_init

    from rsl_rl import runners  # the labelled stand-in (robot_lab_amd/shims/rsl_rl) when rsl-rl-lib is absent

    real_runner_init, real_learn = runners.OnPolicyRunner.__init__, runners.OnPolicyRunner.learn

    def recording_runner_init(self, env, train_cfg, log_dir=None, device="cpu"):
        record.update(agent_device=str(device), agent_seed=int(train_cfg.get("seed", -1)))
        real_runner_init(self, env, train_cfg, log_dir=log_dir, device=device)

    def recording_learn(self, *a, **k):
        real_learn(self, *a, **k)
        flat = torch.cat([p.detach().reshape(-1).float().cpu() for p in self.alg.policy.parameters()])
        record.update(param_sha=hashlib.sha256(flat.numpy().tobytes()).hexdigest(), param_norm=float(flat.norm()), learning_rate=float(self.alg.learning_rate),
                      iterations=int(self.current_learning_iteration), log_dir=self.log_dir, backend=getattr(self.group, "backend", None))

    runners.OnPolicyRunner.__init__, runners.OnPolicyRunner.learn = recording_runner_init, recording_learn

    sys.argv = [script] + argv
    sys.path.insert(0, os.path.dirname(os.path.join(REF, script)))  # what `python script.py` does (train.py imports its sibling cli_args)
    try:
        runpy.run_path(os.path.join(REF, script), run_name="__main__")
        record["finished"] = True
    except _StopAtTheBoundary:
        record["finished"] = False
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
        json.dump(record, f)
    import torch.distributed as dist

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
