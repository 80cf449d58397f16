"""`OnPolicyRunner` as the reference's scripts use it (train.py:205-224, play.py:190-246):

    runner = OnPolicyRunner(env, agent_cfg.to_dict(), log_dir=log_dir, device=agent_cfg.device)
    runner.add_git_repo_to_log(__file__); runner.load(path); runner.learn(num_learning_iterations=..., init_at_random_ep_len=True)
    policy = runner.get_inference_policy(device=...); runner.alg.policy  (exporters, `.reset(dones)`)

`env` is the `RslRlVecEnvWrapper` of the shims around `robot_lab_amd.env.ManagerBasedRLEnv`.  Collection runs on the HIP kernels as
one hipGraph launch per iteration (robot_lab_amd/collect.py), the update is robot_lab_amd/ppo.py, checkpoints use rsl_rl's layout
(`model_<it>.pt`: model_state_dict / optimizer_state_dict / iter / infos)."""
from __future__ import annotations

import os
import time

import torch


class OnPolicyRunner:
    def __init__(self, env, train_cfg: dict, log_dir: str | None = None, device: str = "cpu"):
        from robot_lab_amd.ppo import Trainer

        self.env, self.cfg, self.log_dir, self.device = env, train_cfg, log_dir, device
        pol, alg = dict(train_cfg.get("policy", {})), dict(train_cfg.get("algorithm", {}))
        if pol.get("class_name", "ActorCritic") != "ActorCritic" or alg.get("class_name", "PPO") != "PPO":
            raise NotImplementedError(f"stand-in runner: ActorCritic + PPO only, got {pol.get('class_name')} / {alg.get('class_name')}")
        if pol.get("activation", "elu") != "elu" or pol.get("actor_obs_normalization") or pol.get("critic_obs_normalization"):
            raise NotImplementedError("stand-in runner: ELU networks without observation normalisation (what every robot_lab agent cfg uses)")
        if alg.get("symmetry_cfg") or alg.get("rnd_cfg"):
            raise NotImplementedError("stand-in runner: symmetry augmentation / RND are not wired into the update")
        # options this stand-in does not implement are refused, never ignored (a silently different learner is worse than none)
        groups = dict(train_cfg.get("obs_groups") or {})
        actor_set = groups.get("policy", groups.get("actor", ["policy"]))
        if list(actor_set) != ["policy"] or list(groups.get("critic", ["critic"])) != ["critic"]:
            raise NotImplementedError(f"stand-in runner: obs_groups {groups} - only policy <- ['policy'], critic <- ['critic'] (what every "
                                      f"robot_lab velocity cfg but ANYmal-D's uses) is wired to the fused inference kernels")
        if pol.get("noise_std_type", "scalar") != "scalar":
            raise NotImplementedError(f"stand-in runner: noise_std_type={pol.get('noise_std_type')!r}; the sampling kernel reads a scalar-type std vector")
        if train_cfg.get("empirical_normalization") or alg.get("normalize_advantage_per_mini_batch"):
            raise NotImplementedError("stand-in runner: empirical_normalization / normalize_advantage_per_mini_batch are not implemented")
        # one process per GPU (README.md:323-337, train.py:143-150): rsl_rl's multi-GPU contract lives in robot_lab_amd/dist.py.  A world of
        # N ranks whose learners were not tied together would be N unrelated trainings writing over each other's logs - never run that.
        from robot_lab_amd.dist import LearnerGroup

        self.group = LearnerGroup(device)
        keys = ("value_loss_coef", "use_clipped_value_loss", "clip_param", "entropy_coef", "num_learning_epochs", "num_mini_batches",
                "learning_rate", "schedule", "desired_kl", "max_grad_norm")
        self.trainer = Trainer(env.unwrapped, num_steps_per_env=int(train_cfg.get("num_steps_per_env", 24)), gamma=float(alg.get("gamma", 0.99)),
                               lam=float(alg.get("lam", 0.95)), seed=int(train_cfg.get("seed", 42)), actor_hidden=pol.get("actor_hidden_dims", (512, 256, 128)),
                               critic_hidden=pol.get("critic_hidden_dims", (512, 256, 128)), init_noise_std=float(pol.get("init_noise_std", 1.0)),
                               clip_actions=getattr(env, "clip_actions", None), group=self.group if self.group.enabled else None,
                               **{k: alg[k] for k in keys if k in alg})
        self.alg = self.trainer.alg
        self.alg.policy.reset = lambda dones=None: None  # feed-forward policy: nothing to reset (play.py:246)
        self.save_interval = int(train_cfg.get("save_interval", 50))
        self.current_learning_iteration = 0
        self.git_status_repos = []
        self.last_episode_log = None  # rank 0: extras["log"] with the means over every rank's envs (robot_lab_amd/dist.py reduce_episode_log)

    def add_git_repo_to_log(self, path):
        self.git_status_repos.append(path)

    def learn(self, num_learning_iterations: int, init_at_random_ep_len: bool = False):
        env = self.env.unwrapped
        if init_at_random_ep_len:
            env.episode_length_buf = torch.randint(0, int(env.max_episode_length), (env.num_envs,))
        main = self.group.is_main  # rank 0 logs and writes checkpoints, the other ranks train silently (rsl_rl `disable_logs`)
        if self.log_dir and main:
            os.makedirs(self.log_dir, exist_ok=True)
        start = self.current_learning_iteration
        t0 = time.time()
        for it in range(start, start + num_learning_iterations):
            out = self.trainer.iterate()
            self.current_learning_iteration = it + 1
            steps = self.trainer.storage.num_transitions_per_env * env.num_envs * self.group.world_size
            # the episode log of the JOB, not of rank 0's envs: the packed episode-metric vector SUM all-reduced over the ranks on a side
            # stream (every rank takes part in the collective, rank 0 alone reads the result)
            # Only at the log interval (`self.log_interval`, default every iteration as rsl_rl prints every iteration), and every rank
            # keeps its handle until the next collective and resolves it then (`vector()`: work.wait() + stream order) - a rank must not
            # drop an async Work handle while its tensor is in flight.
            from robot_lab_amd.dist import reduce_episode_log

            if getattr(self, "_log_in_flight", None) is not None:
                self._log_in_flight.vector()
                self._log_in_flight = None
            if (it + 1 - start) % max(1, int(getattr(self, "log_interval", 1))) != 0 and it + 1 != start + num_learning_iterations:
                continue
            episode_log = reduce_episode_log(env)
            if not main:
                self._log_in_flight = episode_log
                continue
            self.last_episode_log = ep = episode_log.result()
            print(f"[rsl_rl stand-in] iteration {it + 1}/{start + num_learning_iterations}  mean reward/step {out['mean_reward']:+.4f}  value loss {out['value_loss']:.4f}  "
                  f"surrogate {out['surrogate_loss']:+.4f}  std {out['action_std']:.3f}  lr {out['learning_rate']:.1e}  "
                  f"{steps * (it + 1 - start) / max(time.time() - t0, 1e-9):.0f} steps/s  "
                  f"episode log over {int(ep['num_envs'])} envs ({int(ep['episodes'])} episodes ended)"
                  + (f"  terrain level {float(ep['Curriculum/terrain_levels']):.2f}" if "Curriculum/terrain_levels" in ep else ""), flush=True)
            if self.log_dir and (it + 1) % self.save_interval == 0:
                self.save(os.path.join(self.log_dir, f"model_{it + 1}.pt"))
        if getattr(self, "_log_in_flight", None) is not None:
            self._log_in_flight.vector()
            self._log_in_flight = None
        if self.log_dir and main:
            self.save(os.path.join(self.log_dir, f"model_{self.current_learning_iteration}.pt"))

    def save(self, path: str, infos=None):
        torch.save({"model_state_dict": self.alg.policy.state_dict(), "optimizer_state_dict": self.alg.optimizer.state_dict(),
                    "iter": self.current_learning_iteration, "infos": infos}, path)

    def load(self, path: str, load_optimizer: bool = True, map_location=None):
        d = torch.load(path, map_location=map_location or self.trainer.device, weights_only=False)
        self.alg.policy.load_state_dict(d["model_state_dict"])
        if load_optimizer and d.get("optimizer_state_dict"):
            self.alg.optimizer.load_state_dict(d["optimizer_state_dict"])
        self.current_learning_iteration = int(d.get("iter", 0))
        self.trainer.push_parameters()
        return d.get("infos")

    def get_inference_policy(self, device=None):
        actor = self.trainer.actor  # the fused HIP inference kernel (csrc/rl_policy.hip), fed by push_parameters()

        def policy(obs):
            return actor(obs if torch.is_tensor(obs) else obs["policy"])

        return policy


class DistillationRunner:
    def __init__(self, *a, **k):
        raise NotImplementedError("stand-in for rsl_rl: DistillationRunner is not provided (no robot_lab velocity task uses it)")
