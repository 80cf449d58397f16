"""`-m gpu`: the env is a LEARNABLE environment - PPO (robot_lab_amd/ppo.py: the reference's hyper-parameters, rsl_rl_ppo_cfg.py:10-37)
on A1 Velocity-Flat for 60 iterations, collected by the HIP kernels as one hipGraph launch per iteration, parameters pushed into the
inference kernels in place (`rl_mlp_set_weights`).  What `scripts/reinforcement_learning/rsl_rl/train.py` does with rsl-rl-lib.
A from-scratch simulator can match its own oracle step for step and still be useless for learning (wrong sign of an action, a reward
nobody can earn, resets that leak state): this is the test that would notice."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_set_weights_in_place_matches_torch():
    import torch

    from robot_lab_amd.policy import MlpPolicy
    from robot_lab_amd.ppo import mlp

    torch.manual_seed(0)
    net = mlp([45, 512, 256, 128, 12]).cuda()
    lin = [m for m in net if isinstance(m, torch.nn.Linear)]
    pol = MlpPolicy([m.weight.detach().cpu().numpy() for m in lin], [m.bias.detach().cpu().numpy() for m in lin], "elu", device="cuda:0")
    x = torch.randn(4096, 45, device="cuda")
    torch.testing.assert_close(pol(x), net(x).detach(), rtol=2e-5, atol=2e-5)
    with torch.no_grad():
        for m in lin:
            m.weight.mul_(0.7).add_(0.01 * torch.randn_like(m.weight))
            m.bias.add_(0.1)
    pol.load_linear_layers(net)
    torch.testing.assert_close(pol(x), net(x).detach(), rtol=2e-5, atol=2e-5)
    x2 = torch.randn(64, 45, device="cuda")  # the small-batch kernel reads the same images
    torch.testing.assert_close(pol(x2), net(x2).detach(), rtol=2e-5, atol=2e-5)


def test_a1_learns_to_track_velocity_commands():
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv
    from robot_lab_amd.ppo import Trainer

    N, iters = 2048, 60
    env = ManagerBasedRLEnv("RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0", num_envs=N, seed=42, device="cuda:0")
    tr = Trainer(env, seed=42)
    env.episode_length_buf = torch.randint(0, env.max_episode_length, (N,), generator=torch.Generator().manual_seed(0))
    rew, err, done = [], [], []
    for it in range(iters):
        out = tr.iterate()
        rew.append(out["mean_reward"])
        done.append(out["done_rate"])
        ex = env.extras.get("log", {})
        if "Metrics/base_velocity/error_vel_xy" in ex:
            err.append(float(ex["Metrics/base_velocity/error_vel_xy"]))
        assert np.isfinite(out["value_loss"]) and np.isfinite(out["surrogate_loss"])
    first, last = float(np.mean(rew[:5])), float(np.mean(rew[-5:]))
    print(f"\n[train] A1 Flat {N} envs: reward/step {first:+.4f} -> {last:+.4f}; done/step {np.mean(done[:5]):.4f} -> {np.mean(done[-5:]):.4f}; "
          f"std {out['action_std']:.3f}; lr {out['learning_rate']:.1e}")
    assert last > first + 0.5 * abs(first) or last > first + 0.01, "the mean step reward did not improve in 60 PPO iterations"
    assert np.mean(done[-5:]) <= np.mean(done[:5]) + 1e-3, "episodes end more often than at the start: the policy is falling over more"
    env.close()
