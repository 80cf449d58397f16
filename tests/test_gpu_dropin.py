"""`-m gpu`: the drop-in surface exercised the way the reference's callers use it:
gym registry with the reference's entry-point string, a zero_agent.py-style loop
(`scripts/tools/zero_agent.py:56-73`) and the `RslRlVecEnvWrapper` protocol (`train.py:202`).
/root/reference is absent on the GPU box, so the cfg is a committed descriptor bundle id."""
import pytest

pytestmark = pytest.mark.gpu
TASK = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"


def _make(num_envs=256):
    from robot_lab_amd import shims

    shims.install()
    import gymnasium as gym

    gym.register(id=TASK, entry_point="isaaclab.envs:ManagerBasedRLEnv", disable_env_checker=True, kwargs={})
    return gym.make(TASK, cfg=TASK, num_envs=num_envs, seed=3, device="cuda:0")


def test_zero_agent_loop():
    import torch

    env = _make()
    assert env.unwrapped.num_envs == 256 and env.unwrapped.device == "cuda:0"
    assert env.observation_space["policy"].shape == (256, 45) and env.action_space.shape == (256, 12)
    obs, _ = env.reset()
    with torch.inference_mode():
        for _ in range(20):
            actions = torch.zeros(env.action_space.shape, device=env.unwrapped.device)
            obs, rew, terminated, time_outs, extras = env.step(actions)
    assert obs["policy"].shape == (256, 45) and obs["critic"].shape == (256, 235)
    assert rew.dtype == torch.float32 and terminated.dtype == torch.bool and time_outs.dtype == torch.bool
    assert torch.isfinite(obs["critic"]).all()
    d = env.unwrapped.scene["robot"].data
    assert d.root_pos_w.shape == (256, 3) and d.root_quat_w.shape == (256, 4)
    assert abs(env.unwrapped.step_dt - 0.02) < 1e-6 and env.unwrapped.max_episode_length == 1000
    env.close()


def test_rsl_rl_wrapper_protocol():
    import torch

    from isaaclab_rl.rsl_rl import RslRlVecEnvWrapper

    env = RslRlVecEnvWrapper(_make(), clip_actions=1.0)
    assert (env.num_envs, env.num_actions, env.max_episode_length) == (256, 12, 1000)
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=env.max_episode_length)  # init_at_random_ep_len
    obs = env.get_observations()
    n_done = 0
    for _ in range(30):
        obs, rew, dones, extras = env.step(torch.randn(256, 12, device="cuda:0") * 3)
        assert dones.dtype == torch.long and "time_outs" in extras
        n_done += int(dones.sum())
        if int(dones.sum()):
            log = extras["log"]
            assert "Episode_Reward/track_lin_vel_xy_exp" in log and float(log["Episode_Termination/time_out"]) >= 1
    assert n_done > 0
    sums = env.unwrapped.reward_manager._episode_sums
    assert set(sums) == set(env.unwrapped.desc.reward_names) and sums["upward"].shape == (256,)
    assert env.unwrapped.command_manager.get_command("base_velocity").shape == (256, 3)
    env.close()


def test_train_and_play_bodies_with_the_runner_stand_in(tmp_path):
    """What `train.py:202-224` and `play.py:190-246` do after gym.make, call for call, against the `rsl_rl` STAND-IN of the shims (the
    real rsl-rl-lib is third-party and absent): wrap, build the runner from the agent cfg's dict (the A1 cfg's numbers,
    .../unitree_a1/agents/rsl_rl_ppo_cfg.py:10-37), learn two iterations with random initial episode lengths, checkpoint in rsl_rl's
    layout; then a second runner loads it, hands out the inference policy, the exporter writes the TorchScript policy and the play loop
    steps.  The scripts themselves run as files in tests/test_reference_scripts.py (needs /root/reference: not on this box)."""
    import importlib.metadata as metadata
    import os

    import torch
    from packaging import version

    from robot_lab_amd import shims

    shims.install()
    installed_version = metadata.version("rsl-rl-lib")
    assert version.parse(installed_version) >= version.parse("3.0.1")  # train.py:64-79
    from isaaclab_rl.rsl_rl import RslRlVecEnvWrapper, export_policy_as_jit
    from rsl_rl.runners import OnPolicyRunner

    agent = dict(seed=42, device="cuda:0", num_steps_per_env=24, max_iterations=2, save_interval=100, experiment_name="unitree_a1_rough", clip_actions=None,
                 policy=dict(class_name="ActorCritic", init_noise_std=1.0, actor_obs_normalization=False, critic_obs_normalization=False,
                             actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation="elu"),
                 algorithm=dict(class_name="PPO", value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=5,
                                num_mini_batches=4, learning_rate=1.0e-3, schedule="adaptive", gamma=0.99, lam=0.95, desired_kl=0.01, max_grad_norm=1.0))
    log_dir = str(tmp_path / "logs")
    env = RslRlVecEnvWrapper(_make(256), clip_actions=agent["clip_actions"])
    runner = OnPolicyRunner(env, agent, log_dir=log_dir, device=agent["device"])
    runner.add_git_repo_to_log(__file__)
    runner.learn(num_learning_iterations=agent["max_iterations"], init_at_random_ep_len=True)
    ckpt = os.path.join(log_dir, "model_2.pt")
    d = torch.load(ckpt, map_location="cpu", weights_only=False)
    assert d["iter"] == 2 and {"std", "actor.0.weight", "critic.6.bias"} <= set(d["model_state_dict"])
    env.close()
    # ---- play.py
    env = RslRlVecEnvWrapper(_make(64), clip_actions=None)
    runner = OnPolicyRunner(env, agent, log_dir=None, device=agent["device"])
    runner.load(ckpt)
    policy = runner.get_inference_policy(device=env.unwrapped.device)
    policy_nn = runner.alg.policy
    export_policy_as_jit(policy_nn, normalizer=None, path=str(tmp_path / "exported"), filename="policy.pt")
    assert os.path.isfile(str(tmp_path / "exported" / "policy.pt"))
    obs = env.get_observations()
    with torch.inference_mode():
        for _ in range(5):
            actions = policy(obs)
            obs, _, dones, _ = env.step(actions)
            policy_nn.reset(dones)
    # the inference policy IS the trained actor (HIP kernel fed by the checkpoint) - against the torch module on the same observations
    o = obs["policy"] if not torch.is_tensor(obs) else obs
    torch.testing.assert_close(policy(obs).clone(), policy_nn.actor(o).detach(), rtol=2e-5, atol=2e-5)
    jit = torch.jit.load(str(tmp_path / "exported" / "policy.pt"))
    torch.testing.assert_close(jit(o.cpu()), policy_nn.actor(o).detach().cpu(), rtol=1e-5, atol=1e-5)
    env.close()
