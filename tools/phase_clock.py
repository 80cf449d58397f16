#!/usr/bin/env python
"""Phase breakdown of the env-step kernel from in-kernel shader-clock stamps.

Needs a `-DRL_PHASE_CLOCK` build of the HIP library (csrc/env_step.h RL_PHASE): lane 0 of every wavefront accumulates the
ticks it spends in each phase into float row [wavefront][phase id] behind the reward-term rows.

    bash tools/build_variant.sh clock 34 - -DRL_PHASE_CLOCK [-DRL_ENV_SPEC_ONLY=1 -DRL_ENV_SPEC_SUB=4]   (the library's own flags, incl. -mllvm -amdgpu-remove-redundant-endcf=false)
    RL_ENV_LIB=robot_lab_amd/csrc/variants/clock_34.so python tools/phase_clock.py [task] [num_envs]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robot_lab_amd.env import ManagerBasedRLEnv, _DevView  # noqa: E402

PHASES = ['load', 'action', 'sub.actuators+kinematics', 'sub.contact_fetch', 'sub.link_records', 'sub.contact_pass1', 'sub.leg_sum', 'sub.crba', 'sub.schur', 'sub.aba', 'sub.cross_leg_sum', 'sub.trunk_solve', 'sub.back_subst', 'sub.contact_pass2', 'sub.sensor+integrate', 'terminations', 'rewards', 'rewards.terms', 'rewards.writeback', 'resets+commands+push', 'observations', 'obs.policy_done', 'obs.flush', 'store', 'end', 'rewards.term_body', 'reset.uniforms', 'reset.state', 'reset.log', 'commands', 'push', 'obs.kinematics']
ROW0, SLOTS = 24, 32

# --reset-env0: the first env of EVERY wavefront times out on every counted step (its episode clock is set to the last step; the clock is
# kept by the wavefront's lane 0, which belongs to that env): the table then shows what a reset costs the wavefront that carries it, phase
# by phase, against a run without the flag
RESET0 = any(a.startswith("--reset-env0") for a in sys.argv)
# --reset-env0=K: only every K-th wavefront (few resets per step: no contention on the log's atomics, as in a steady-state launch); the table
# then has a second column, the wavefronts that did not reset
STRIDE = max([int(a.split("=")[1]) for a in sys.argv if a.startswith("--reset-env0=")] + [1])
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
task = argv[0] if len(argv) > 0 else "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"
N = int(argv[1]) if len(argv) > 1 else 4096
STEPS = 100
env = ManagerBasedRLEnv(task, num_envs=N, seed=42, device="cuda:0")
env.reset()
A = env.num_actions
g = torch.Generator(device="cuda:0").manual_seed(1)
nwave = -(-N // int(env._native.envs_per_wavefront()))  # one row of SLOTS per wavefront (the lane mapping decides how many envs it holds)
rt = env._bufs["REWARD_TERMS"]
Np = rt.shape[1]
assert nwave * SLOTS <= (40 - ROW0) * Np, "the borrowed reward-term rows are too short for this many wavefronts"
buf = torch.as_tensor(_DevView(rt.data_ptr() + ROW0 * Np * 4, (nwave, SLOTS), np.float32, env), device="cuda:0")
for s in range(50):
    env.step(torch.rand(N, A, device="cuda:0", generator=g) * 2 - 1)
torch.cuda.synchronize()
buf.zero_()
ept = int(env._native.envs_per_wavefront())
ep0 = torch.zeros(N, dtype=torch.int64)
ep0[::ept * STRIDE] = int(env.max_episode_length) - 1
for s in range(STEPS):
    if RESET0:
        env.episode_length_buf = ep0
    env.step(torch.rand(N, A, device="cuda:0", generator=g) * 2 - 1)
torch.cuda.synchronize()
rows = buf.cpu().numpy().astype(np.float64) / STEPS
others = rows[np.arange(nwave) % STRIDE != 0].mean(axis=0) if RESET0 and STRIDE > 1 else None
acc = rows[::STRIDE].mean(axis=0) if RESET0 else rows.mean(axis=0)
bogus = acc > 1e9  # a row that holds a raw time stamp instead of a sum of intervals (seen in round 4's `load` row): left out, and said so
acc[bogus] = 0.0
tot = acc.sum()
print(f"spec id {env._native.spec_id()}, {int(env._native.envs_per_wavefront())} envs per wavefront{', env 0 of every wavefront reset on every step' if RESET0 else ''}" + (f"; rows left out as bogus: {[PHASES[i] for i in np.nonzero(bogus)[0]]}" if bogus.any() else ""))
print(f"{task} N={N}: mean shader-clock ticks per wavefront per step by phase (sub.* = the 4 substeps together), total {tot:.0f}")
for i, (n, v) in enumerate(zip(PHASES, acc)):
    if v > 0:
        print(f"  {n:28s} {v:10.0f}  {100 * v / tot:5.1f} %" + (f"   | wavefronts without a reset {others[i]:10.0f}   difference {v - others[i]:+9.0f}" if others is not None else ""))
if others is not None:
    print(f"  {'total':28s} {tot:10.0f}            | {others[others < 1e9].sum():10.0f}   difference {tot - others[others < 1e9].sum():+9.0f}   (every {STRIDE}th wavefront resets its env 0)")
