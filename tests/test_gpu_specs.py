"""`-m gpu`: the step kernels SPECIALISED on a task (robot_lab_amd/csrc/env_spec.h) against the interpreter's kernels of the same
library, on the GPU, in every lane mapping a specialised kernel is built for: both envs take every step from the SAME state (the
interpreter env adopts the specialised env's state through the C-ABI exchange before each step), so what is compared is one step of
the two kernels - per-term rewards, the reward, dones, both observation groups, the state they leave.  The oracle tiers
(tests/test_gpu_teacher_forced.py, test_gpu_parity.py, test_gpu_canary.py) run the specialised kernels for these tasks by default."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

A1, GO2, GO2W, G1 = (f"RobotLab-Isaac-Velocity-Rough-Unitree-{r}-v0" for r in ("A1", "Go2", "Go2W", "G1"))
# (task, spec id, RL_ENV_SUB, RL_ENV_WG)
CASES = [
    (A1, 1, "4", ""), (A1, 1, "4", "-4"), (A1, 1, "2", "-4"), (A1, 1, "1", "-4"), (A1, 1, "1", ""),
    (GO2, 2, "4", "-4"), (GO2, 2, "2", ""), (GO2, 2, "1", "-4"),
    (GO2W, 3, "4", "-4"), (GO2W, 3, "2", "-4"), (GO2W, 3, "1", ""),
    (G1, 4, "8", "-4"), (G1, 4, "8", ""),
    # round 6: the Flat twins (BASELINE config 1 = A1 Flat) in the mapping a 4096-env (G1: 2048) launch runs + one more each
    (A1.replace("Rough", "Flat"), 5, "4", "-4"), (A1.replace("Rough", "Flat"), 5, "1", ""),
    (GO2.replace("Rough", "Flat"), 6, "4", "-4"), (GO2W.replace("Rough", "Flat"), 7, "4", "-4"), (GO2W.replace("Rough", "Flat"), 7, "2", ""),
    (G1.replace("Rough", "Flat"), 8, "8", "-4"),
]


# Tasks the library has NO built-in Spec for, specialised at run time (robot_lab_amd/jit.py: Spec source from the library, hipcc on the
# box, plugin registered, env created again on it): a quadruped of another make, a wheeled one on the merged instance, a humanoid
JIT_CASES = [
    ("RobotLab-Isaac-Velocity-Rough-Unitree-B2-v0", "jit", "4", "-4"),
    ("RobotLab-Isaac-Velocity-Rough-Deeprobotics-M20-v0", "jit", "4", ""),
    ("RobotLab-Isaac-Velocity-Rough-RobotEra-Xbot-v0", "jit", "8", "-4"),
    # the reward kinds no built-in Spec uses: base_height_l2 (its 3 x 3 ray caster), wheel_vel_penalty, feet_distance_y_exp (Tita); the hand-stand terms
    ("RobotLab-Isaac-Velocity-Rough-DDTRobot-Tita-v0", "jit", "4", "-4"),
    ("RobotLab-Isaac-Velocity-Rough-DDTRobot-Tita-v0", "jit", "1", ""),
    ("RobotLab-Isaac-Velocity-Rough-HandStand-Unitree-A1-v0", "jit", "4", "-4"),
    ("RobotLab-Isaac-Velocity-Flat-HandStand-Unitree-A1-v0", "jit", "2", ""),
]


@pytest.mark.parametrize("task,sid,sub,wg", CASES + JIT_CASES)
def test_specialised_kernel_equals_interpreter_kernel(task, sid, sub, wg, monkeypatch, tmp_path):
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    N, steps = 512, 24
    monkeypatch.setenv("RL_ENV_SUB", sub)
    if wg:
        monkeypatch.setenv("RL_ENV_WG", wg)
    monkeypatch.setenv("RL_ENV_SPEC", "1")
    if sid == "jit":
        monkeypatch.setenv("RL_ENV_JIT_CACHE", str(tmp_path))
        a = ManagerBasedRLEnv(task, num_envs=N, seed=11, device="cuda:0", specialise=True)
        assert a._native.spec_id() >= 1000 and "specialised" in repr(a), repr(a)
        sid = a._native.spec_id()
    else:
        a = ManagerBasedRLEnv(task, num_envs=N, seed=11, device="cuda:0")
    monkeypatch.setenv("RL_ENV_SPEC", "0")
    b = ManagerBasedRLEnv(task, num_envs=N, seed=11, device="cuda:0")
    assert a._native.spec_id() == sid and b._native.spec_id() == 0
    assert a._native.envs_per_wavefront() == 16 // int(sub)
    a.reset(); b.reset()
    ep = torch.randint(0, a.max_episode_length, (N,), generator=torch.Generator().manual_seed(5))
    ep[::9] = a.max_episode_length - 1 - (torch.arange(len(ep[::9])) % steps)  # time-out resets all along the run
    a.episode_length_buf = ep
    g = torch.Generator(device="cuda").manual_seed(3)
    n_terms = a.reward_terms().shape[0]
    seen = np.zeros(n_terms, bool)
    worst = dict(obs=0.0, term_rel=0.0, state=0.0, bit_different_obs_entries=0)
    obs_err = []
    for s in range(steps):
        b.load_state(a.read_state())  # one step of each kernel from the same state
        act = torch.rand(N, a.num_actions, device="cuda", generator=g) * 2 - 1
        if s % 5 == 3:
            act.zero_()
        oa, ra, ta, toa, _ = a.step(act)
        ob, rb, tb, tob, _ = b.step(act)
        assert torch.equal(ta, tb) and torch.equal(toa, tob), (task, s)
        for grp in ("policy", "critic"):
            d = (oa[grp] - ob[grp]).abs() / ob[grp].abs().clamp(min=1.0)
            obs_err.append(d.flatten().cpu().numpy())
            worst["obs"] = max(worst["obs"], float(d.max()))
            worst["bit_different_obs_entries"] += int((d != 0).sum())
        xa, xb = a.reward_terms()[:, :N].double().cpu().numpy(), b.reward_terms()[:, :N].double().cpu().numpy()
        seen |= (xb != 0).any(axis=1)
        err = np.abs(xa - xb) / (np.abs(xb) + 1e-7)
        worst["term_rel"] = max(worst["term_rel"], float(err.max()))
        # (a term is a sum of <= 30 fp32 products in another order - and, since round 6, on a state that differs by one step's round-off where
        # the Spec composes axis-aligned joint rotations in their sparse form; joint_acc_l2 squares a finite difference of the velocities)
        # (floor: 0.05 % of a typical step reward.  It was 1e-6 until the kernels were compiled with -fno-signed-zeros -ffinite-math-only: the two
        # template instantiations fold different zero terms, and one entry of 12 x 512 x 24 - HandStand A1, term 1, step 10 - came out 3.8e-6 apart)
        assert np.all(np.abs(xa - xb) <= 5e-4 * np.abs(xb) + 5e-6), (task, s, np.abs(xa - xb).max(axis=1))
        # the reward is a sum of terms of both signs: its error is bounded by the terms' magnitudes, not by its own
        assert np.all(np.abs(ra.double().cpu().numpy() - rb.double().cpu().numpy()) <= 5e-4 * np.abs(xb).sum(axis=0) + 1e-5), (task, s)
        sa, sb = a.read_state(), b.read_state()
        for k2 in ("root_state", "joint_pos", "joint_vel", "task_state"):
            xs, ys = np.asarray(sa[k2], dtype=np.float64), np.asarray(sb[k2], dtype=np.float64)
            worst["state"] = max(worst["state"], float((np.abs(xs - ys) / np.maximum(np.abs(ys), 1.0)).max()))  # (relative: wheel speeds are tens of rad/s)
        # the contact timers are DISCRETE in what moves them (|F| against the 1 N threshold per substep): a force that sits on the threshold
        # flips with the last bit and the two timers then differ by whole substeps (round 6, A1 Rough sub2 after -ffinite-math-only: one entry of
        # 24 steps x 512 envs x 17 bodies x 4 timers, 0.025 s apart).  Counted, not bounded by the state tolerance; re-synced with the state.
        tflip = np.abs(np.asarray(sa["contact_timers"], dtype=np.float64) - np.asarray(sb["contact_timers"], dtype=np.float64)) > 1e-6
        worst["timer_entries_apart"] = worst.get("timer_entries_apart", 0) + int(tflip.sum())
        assert np.array_equal(sa["episode_length"], sb["episode_length"])
    # the physics of the two kernels differ by the kinematics' two forms where a Spec joint is axis-aligned (round-off of ONE step, amplified by
    # a stiff contact in a few entries), else not at all: nearly every entry bit equal, the 99th percentile at 2e-5, nothing past 2e-3
    worst["obs_p99"] = float(np.quantile(np.concatenate(obs_err), 0.99))
    assert worst["obs"] <= 2e-3 and worst["state"] <= 2e-3 and worst["obs_p99"] <= 2e-5, worst
    assert worst["timer_entries_apart"] <= 8, worst  # (of ~1e6 compared: at most a couple of threshold flips, each up to four timer words)
    assert seen.sum() >= len(seen) - 2, f"only {seen.sum()} of {len(seen)} terms ever non-zero"
    print("\n[spec-vs-interpreter]", json.dumps(dict(task=task, sub=sub, wg=wg, **worst)))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "spec_vs_interpreter.jsonl"), "a") as f:
            f.write(json.dumps(dict(task=task, sub=sub, wg=wg, **worst)) + "\n")
    a.close(); b.close()
