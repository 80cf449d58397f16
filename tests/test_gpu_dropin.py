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
