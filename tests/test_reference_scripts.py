"""The reference's own files, executed unmodified through robot_lab_amd.shims (skipped where
/root/reference is absent, e.g. on the GPU box - there tests/test_gpu_dropin.py runs the same loop bodies):

* `scripts/tools/zero_agent.py` / `random_agent.py` run as files (runpy) - argument parsing, AppLauncher, `import robot_lab.tasks`,
  `parse_env_cfg`, `gym.make(task, cfg=env_cfg)` - up to the one thing this container lacks: a HIP device (the env has no CPU path).
* `scripts/reinforcement_learning/rsl_rl/train.py` and `play.py` run as files too: rsl-rl-lib (third-party, not installable here) is
  replaced by the labelled stand-in `robot_lab_amd/shims/rsl_rl` when - and only when - the real package is absent.
* `VEL/mdp/utils.py` + `commands.py:61-85`: the "pits" restriction of UniformThresholdVelocityCommand never fires on the terrain
  the velocity tasks use (no sub-terrain of that name), which is why the lane program does not carry it.
* `export_policy_as_jit` (play.py:232) is a real exporter."""
import os
import runpy
import sys

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")
TASK = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"


def _run(script, argv):
    from robot_lab_amd import shims

    shims.install(shims.REFERENCE_SOURCE)
    old, old_path = sys.argv, list(sys.path)
    sys.argv = [script] + argv
    sys.path.insert(0, os.path.dirname(os.path.join(REF, script)))  # what `python script.py` does (train.py imports its sibling cli_args)
    try:
        runpy.run_path(os.path.join(REF, script), run_name="__main__")
    finally:
        sys.argv, sys.path[:] = old, old_path


@pytest.mark.parametrize("script", ["scripts/tools/zero_agent.py", "scripts/tools/random_agent.py"])
def test_agent_scripts_reach_gym_make(script, monkeypatch):
    import torch

    from robot_lab_amd.capi import RlEnvError

    monkeypatch.setenv("RL_SHIM_MAX_STEPS", "3")
    if torch.cuda.is_available():
        _run(script, ["--task", TASK, "--num_envs", "16"])  # runs its loop for 3 steps and exits
    else:
        with pytest.raises(RlEnvError, match="no HIP device|MI355X only"):
            _run(script, ["--task", TASK, "--num_envs", "16"])


def test_train_script_runs_as_a_file(tmp_path, monkeypatch):
    """`python scripts/reinforcement_learning/rsl_rl/train.py --task ... --headless --max_iterations 2`, unmodified (train.py:1-234): version
    check of rsl-rl-lib, hydra cfg loading, cli overrides, gym.make, RslRlVecEnvWrapper, OnPolicyRunner(...).learn(...), dump_yaml.
    rsl-rl-lib itself is third-party and not installable here: `robot_lab_amd/shims/rsl_rl` is a labelled STAND-IN (HIP collection loop +
    the torch PPO update of robot_lab_amd/ppo.py) that is only reachable when the real package is absent.  On a GPU box the script trains
    two iterations and leaves a checkpoint in rsl_rl's layout; without a HIP device it gets as far as gym.make (the env has no CPU path)."""
    import glob

    import torch

    from robot_lab_amd.capi import RlEnvError

    monkeypatch.chdir(tmp_path)  # train.py logs under ./logs/rsl_rl/<experiment>/<time stamp>
    argv = ["--task", TASK, "--num_envs", "64", "--headless", "--max_iterations", "2"]
    if not torch.cuda.is_available():
        with pytest.raises(RlEnvError, match="no HIP device|MI355X only"):
            _run("scripts/reinforcement_learning/rsl_rl/train.py", argv)
        return
    _run("scripts/reinforcement_learning/rsl_rl/train.py", argv)
    ckpt = glob.glob(os.path.join(str(tmp_path), "logs", "rsl_rl", "unitree_a1_rough", "*", "model_2.pt"))
    assert len(ckpt) == 1
    d = torch.load(ckpt[0], map_location="cpu", weights_only=False)
    assert d["iter"] == 2 and {"std", "actor.0.weight", "critic.6.bias"} <= set(d["model_state_dict"])
    assert os.path.isfile(os.path.join(os.path.dirname(ckpt[0]), "params", "agent.yaml"))
    # ... and play.py (play.py:1-260) loads that checkpoint, exports the policy and steps the env with it
    monkeypatch.setenv("RL_SHIM_MAX_STEPS", "3")
    _run("scripts/reinforcement_learning/rsl_rl/play.py", ["--task", TASK, "--num_envs", "16", "--headless", "--checkpoint", ckpt[0]])
    assert os.path.isfile(os.path.join(os.path.dirname(ckpt[0]), "exported", "policy.pt"))


def test_pits_branch_of_the_command_term_is_a_no_op():
    """commands.py:61-85 restricts commands on a sub-terrain named "pits"; utils.py:27-28 returns None for a generator without
    one, and is_robot_on_terrain then reports nobody on it - for the terrain cfg of every velocity task of BASELINE.json."""
    import torch

    from robot_lab_amd import shims

    shims.install(shims.REFERENCE_SOURCE)
    import robot_lab.tasks  # noqa: F401
    from isaaclab_tasks.utils import parse_env_cfg
    from robot_lab.tasks.manager_based.locomotion.velocity.mdp import utils as ref_utils

    for task in (TASK, "RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0", "RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0",
                 "RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0"):
        cfg = parse_env_cfg(task, device="cpu", num_envs=8)
        gen = cfg.scene.terrain.terrain_generator
        assert "pits" not in gen.sub_terrains
        assert ref_utils._get_terrain_column_range(gen, "pits", "cpu") is None
        # the live check the command term makes every step
        import types

        terrain = types.SimpleNamespace(cfg=cfg.scene.terrain, terrain_types=torch.zeros(8, dtype=torch.long), terrain_origins=torch.zeros(10, 20, 3))
        env = types.SimpleNamespace(num_envs=8, device="cpu", scene=types.SimpleNamespace(terrain=terrain))
        assert not ref_utils.is_robot_on_terrain(env, "pits").any()
        assert not ref_utils.is_env_assigned_to_terrain(env, "pits").any()


def test_export_policy_as_jit_round_trip(tmp_path):
    import torch

    from robot_lab_amd import shims

    shims.install()
    from isaaclab_rl.rsl_rl import export_policy_as_jit, export_policy_as_onnx

    assert callable(export_policy_as_jit) and callable(export_policy_as_onnx)
    policy = torch.nn.Module()
    policy.actor = torch.nn.Sequential(torch.nn.Linear(45, 512), torch.nn.ELU(), torch.nn.Linear(512, 256), torch.nn.ELU(),
                                       torch.nn.Linear(256, 128), torch.nn.ELU(), torch.nn.Linear(128, 12))
    export_policy_as_jit(policy, normalizer=None, path=str(tmp_path), filename="policy.pt")
    loaded = torch.jit.load(str(tmp_path / "policy.pt"))
    x = torch.randn(7, 45)
    torch.testing.assert_close(loaded(x), policy.actor(x))
    try:
        import onnx  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="onnx"):
            export_policy_as_onnx(policy, path=str(tmp_path))
