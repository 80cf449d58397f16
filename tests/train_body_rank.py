"""Test infrastructure: ONE rank of the reference's `train.py --distributed` on a box without /root/reference (the GPU box).

The body below is `scripts/reinforcement_learning/rsl_rl/train.py:41-55,118-150,177,202-224` call for call - parse `--distributed`, boot the
launcher, device `cuda:{app_launcher.local_rank}`, seed `agent_cfg.seed + app_launcher.local_rank`, `gym.make`, `RslRlVecEnvWrapper`,
`OnPolicyRunner(...).learn(...)` - with a committed descriptor bundle id where the script loads the cfg classes through hydra.  The script
itself runs as a file, two ranks, in `tests/test_distributed_train.py` (CPU tier, needs /root/reference).  Writes `$RL_TEST_OUT/rank<r>.json`."""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    sys.path.insert(0, ROOT)
    from robot_lab_amd import shims

    shims.install()
    from isaaclab.app import AppLauncher

    parser = argparse.ArgumentParser()
    parser.add_argument("--task", type=str, default="RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0")
    parser.add_argument("--num_envs", type=int, default=64)
    parser.add_argument("--max_iterations", type=int, default=2)
    parser.add_argument("--distributed", action="store_true", default=False)
    AppLauncher.add_app_launcher_args(parser)
    args_cli = parser.parse_args()
    app_launcher = AppLauncher(args_cli)

    import gymnasium as gym
    import torch
    from isaaclab_rl.rsl_rl import RslRlVecEnvWrapper
    from rsl_rl.runners import OnPolicyRunner

    agent = dict(seed=42, device="cuda:0", num_steps_per_env=24, max_iterations=args_cli.max_iterations, save_interval=100, experiment_name="unitree_a1_rough",
                 clip_actions=None,
                 policy=dict(class_name="ActorCritic", init_noise_std=1.0, actor_obs_normalization=False, critic_obs_normalization=False,
                             actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu"),
                 algorithm=dict(class_name="PPO", value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=5,
                                num_mini_batches=4, learning_rate=1.0e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0))
    sim_device, env_seed = args_cli.device, agent["seed"]
    if args_cli.distributed:  # train.py:143-150
        sim_device = f"cuda:{app_launcher.local_rank}"
        agent["device"] = f"cuda:{app_launcher.local_rank}"
        env_seed = agent["seed"] + app_launcher.local_rank
        agent["seed"] = env_seed
    log_dir = os.path.join(os.environ["RL_TEST_OUT"], "logs")
    gym.register(id=args_cli.task, entry_point="isaaclab.envs:ManagerBasedRLEnv", disable_env_checker=True, kwargs={})
    env = gym.make(args_cli.task, cfg=args_cli.task, num_envs=args_cli.num_envs, seed=env_seed, device=sim_device)
    env = RslRlVecEnvWrapper(env, clip_actions=agent["clip_actions"])
    runner = OnPolicyRunner(env, agent, log_dir=log_dir, device=agent["device"])
    runner.add_git_repo_to_log(__file__)
    runner.learn(num_learning_iterations=agent["max_iterations"], init_at_random_ep_len=True)

    rank = int(os.environ.get("RANK", "0"))
    flat = torch.cat([p.detach().reshape(-1).float().cpu() for p in runner.alg.policy.parameters()])
    rec = dict(rank=rank, world=int(os.environ.get("WORLD_SIZE", "1")), launcher_local_rank=app_launcher.local_rank, launcher_global_rank=app_launcher.global_rank,
               sim_device=sim_device, env_device=env.unwrapped.device, env_seed=env_seed, agent_device=agent["device"], backend=getattr(runner.group, "backend", None),
               param_sha=hashlib.sha256(flat.numpy().tobytes()).hexdigest(), finite=bool(torch.isfinite(flat).all()), learning_rate=float(runner.alg.learning_rate),
               iterations=runner.current_learning_iteration, mean_reward=float(runner.trainer.storage.rewards.mean()),
               reward_sha=hashlib.sha256(runner.trainer.storage.rewards.cpu().numpy().tobytes()).hexdigest(),
               # rank 0: the episode log of the last iteration, reduced over every rank's envs (robot_lab_amd/dist.py reduce_episode_log)
               episode_log_envs=None if runner.last_episode_log is None else float(runner.last_episode_log["num_envs"]))
    env.close()
    with open(os.path.join(os.environ["RL_TEST_OUT"], f"rank{rank}.json"), "w") as f:
        json.dump(rec, f)
    import torch.distributed as dist

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
