#!/usr/bin/env python
"""End-to-end check that the simulator is a LEARNABLE environment: PPO on a robot_lab velocity task, all of it on the MI355X.

    collect: 24 x (actor + critic [rl_policy.hip] -> sample [rl_rollout.hip] -> env.step [rl_env.hip]) + GAE, ONE hipGraph launch
    update:  robot_lab_amd/ppo.py (torch autograd; rsl_rl's PPO.update with the reference's hyper-parameters, rsl_rl_ppo_cfg.py:10-37)
    push:    rl_mlp_set_weights (the inference kernels keep their buffers: the captured graph stays valid)

What the reference runs as `python scripts/reinforcement_learning/rsl_rl/train.py --task ... --headless` (train.py:177-224) with
rsl-rl-lib, which is not installable here.  A robot that learns to follow velocity commands within a few hundred iterations is
evidence that observations, actions, actuators, contacts, rewards, resets and curricula hang together - something no per-step
parity test against our own oracle can show.
    python tools/train_demo.py [--task ID] [--num-envs N] [--iterations K] [--out DIR]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robot_lab_amd.env import ManagerBasedRLEnv  # noqa: E402
from robot_lab_amd.ppo import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--iterations", type=int, default=300)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out"))
    ap.add_argument("--print-every", type=int, default=10)
    ap.add_argument("--also-terminate-on", default="", help="DIAGNOSTIC, not the reference's cfg: regex of body names added to the illegal-contact termination")
    a = ap.parse_args()
    if a.also_terminate_on:
        import re

        from robot_lab_amd.scene import load_bundle

        desc, extra = load_bundle(a.task)
        hit = [i for i, n in enumerate(desc.body_names) if re.fullmatch(a.also_terminate_on, n)]
        for i in hit:
            desc.task.illegal_body_mask |= 1 << i
        desc.task.term_illegal_contact = 1
        print(f"DIAGNOSTIC run: illegal-contact termination extended to {[desc.body_names[i] for i in hit]}")
        env = ManagerBasedRLEnv(desc=desc, extra=extra, num_envs=a.num_envs, seed=a.seed, device="cuda:0")
    else:
        env = ManagerBasedRLEnv(a.task, num_envs=a.num_envs, seed=a.seed, device="cuda:0")
    print(env, flush=True)  # (names the step kernel: specialised on the task, or the interpreter)
    tr = Trainer(env, seed=a.seed)
    # init_at_random_ep_len=True (train.py:224): the first time-outs are spread over an episode length
    env.episode_length_buf = torch.randint(0, env.max_episode_length, (a.num_envs,), generator=torch.Generator().manual_seed(a.seed))
    names = list(env.desc.reward_names)
    track = [n for n in names if n.startswith("track_lin_vel_xy")][0]
    log, t_collect, t_update = [], 0.0, 0.0
    t_start = time.perf_counter()
    for it in range(a.iterations):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.collector.collect()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st = tr.storage
        row = dict(iteration=it, mean_step_reward=float(st.rewards.mean()), done_rate=float(st.dones.float().mean()))
        row.update(tr.alg.update(st, tr.gen))
        tr.push_parameters()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t_collect += t1 - t0
        t_update += t2 - t1
        ex = env.extras.get("log", {})
        for k in ("Episode_Reward/" + track, "Metrics/base_velocity/error_vel_xy", "Metrics/base_velocity/error_vel_yaw", "Episode_Termination/time_out",
                  "Curriculum/terrain_levels"):
            if k in ex:
                row[k] = float(ex[k])
        row["action_std"] = float(tr.policy.std.detach().mean())
        if it % a.print_every == 0 or it == a.iterations - 1:  # posture of the batch (an export of the state: only when a line is printed)
            d = env.scene["robot"].data
            qw = d.root_quat_w
            row["root_height"] = float((d.root_pos_w[:, 2] - env.scene.env_origins[:, 2]).mean())
            row["upright"] = float((1.0 - 2.0 * (qw[:, 1] ** 2 + qw[:, 2] ** 2)).mean())  # z component of the body's up axis: 1 = upright, 0 = on a side
        log.append(row)
        if it % a.print_every == 0 or it == a.iterations - 1:
            print(f"it {it:4d}  reward/step {row['mean_step_reward']:+.4f}  {track} {row.get('Episode_Reward/' + track, float('nan')):.3f}  "
                  f"err_xy {row.get('Metrics/base_velocity/error_vel_xy', float('nan')):.3f}  done/step {row['done_rate']:.4f}  std {row['action_std']:.3f}  "
                  f"lr {row['learning_rate']:.1e}  kl {row['kl']:.4f}  v_loss {row['value_loss']:.4f}  height {row.get('root_height', float('nan')):.3f}  upright {row.get('upright', float('nan')):+.3f}", flush=True)
    wall = time.perf_counter() - t_start
    n_steps = a.iterations * st.num_transitions_per_env * a.num_envs
    summary = dict(task=a.task, step_kernel=env.step_kernel, num_envs=a.num_envs, iterations=a.iterations, wall_s=wall, env_steps=n_steps, env_steps_per_s=n_steps / wall,
                   collect_ms_per_iteration=1e3 * t_collect / a.iterations, update_ms_per_iteration=1e3 * t_update / a.iterations,
                   first=log[0], last=log[-1])
    print(json.dumps({k: v for k, v in summary.items() if k not in ("first", "last")}))
    os.makedirs(a.out, exist_ok=True)
    tag = a.task.replace("RobotLab-Isaac-Velocity-", "").replace("-v0", "")
    with open(os.path.join(a.out, f"train_demo_{tag}.json"), "w") as f:
        json.dump(dict(summary=summary, log=log), f, indent=1)
    torch.save(tr.policy.state_dict(), os.path.join(a.out, f"train_demo_{tag}.pt"))
    env.close()


if __name__ == "__main__":
    main()
