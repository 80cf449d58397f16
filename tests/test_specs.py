"""Task-specialised term stacks (robot_lab_amd/csrc/env_spec.h), CPU tier: the generated constants are what the generator writes today,
rl_env_create picks a Spec exactly when the env's tables equal its constants, and the specialised lane program gives the interpreter's
results (same source as the HIP kernels, run by the CPU lane emulator: every Spec with one lane per limb and in the lane mapping
its kernel runs at the BASELINE size).  The HIP kernels themselves: tests/test_gpu_specs.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import emu_load_state, emu_read_state, host_view
from robot_lab_amd.capi import NativeEnv
from robot_lab_amd.scene import build_world, load_bundle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPECS = {  # id -> task (tools/gen_specs.py SPECS)
    1: "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0",
    2: "RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0",
    3: "RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0",
    4: "RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0",
    5: "RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0",
    6: "RobotLab-Isaac-Velocity-Flat-Unitree-Go2-v0",
    7: "RobotLab-Isaac-Velocity-Flat-Unitree-Go2W-v0",
    8: "RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0",
}


def make(task, N, seed, lib, mutate=None):
    desc, extra = load_bundle(task)
    if mutate is not None:
        mutate(desc)
    h, to, eo = build_world(desc, extra, N, 0)
    return NativeEnv(desc, h, to, eo, N, seed, 0, lib)


def test_generated_specs_are_current(emu_lib):
    """csrc/spec/* is what tools/gen_specs.py writes from the committed descriptor bundles today."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_specs.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr


def test_spec_is_picked_only_for_identical_tables(emu_lib, monkeypatch):
    monkeypatch.delenv("RL_ENV_SPEC", raising=False)
    for sid, task in SPECS.items():
        assert make(task, 4, 1, emu_lib).spec_id() == sid, task
    # another robot has other terms: the interpreter
    assert make("RobotLab-Isaac-Velocity-Rough-Unitree-B2-v0", 4, 1, emu_lib).spec_id() == 0

    def play(desc):  # play.py:126-139: observation corruption, pushes, curricula off - the task's own Spec still fits
        desc.task.policy_corrupt = 0
        desc.task.ev_push = 0
        desc.terrain.curriculum = 0

    for sid, task in SPECS.items():
        assert make(task, 4, 1, emu_lib, play).spec_id() == sid, task

    def corrupt_critic(desc):  # ... but a group the Spec runs clean may not corrupt
        desc.task.critic_corrupt = 1

    assert make(SPECS[1], 4, 1, emu_lib, corrupt_critic).spec_id() == 0

    def heavier(desc):  # one edited weight: not the task the kernel was compiled for
        desc.task.rewards[0].weight = desc.task.rewards[0].weight * 2.0

    def other_noise(desc):
        desc.task.policy[0].noise_hi = 0.3

    def other_mask(desc):
        desc.task.rewards[2].joint_mask = 0x7

    for mutate in (heavier, other_noise, other_mask):
        assert make(SPECS[1], 4, 1, emu_lib, mutate).spec_id() == 0, mutate.__name__
    monkeypatch.setenv("RL_ENV_SPEC", "0")
    assert make(SPECS[1], 4, 1, emu_lib).spec_id() == 0


def spec_vs_interpreter(task, N, steps, lib, monkeypatch, sid, mutate=None):
    """Both lane programs take every step from the SAME state (the interpreter env adopts the specialised env's state through the C-ABI
    exchange before each step) with the same actions.  Since round 6 a Spec also composes axis-aligned joint rotations in their sparse
    form (env_spec.h spec_axis_kind): the physics of the two programs agree to fp32 round-off of one step, not bit for bit."""
    monkeypatch.setenv("RL_ENV_SPEC", "1")
    a = make(task, N, 5, lib, mutate)
    monkeypatch.setenv("RL_ENV_SPEC", "0")
    b = make(task, N, 5, lib, mutate)
    assert a.spec_id() == sid and b.spec_id() == 0
    a.reset(); b.reset()
    rng = np.random.default_rng(0)
    n_terms = host_view(a, "REWARD_TERMS").shape[0]
    seen = np.zeros(n_terms, bool)
    obs_err = []
    for s in range(steps):
        act = (rng.random((N, a.num_actions), dtype=np.float32) * 2 - 1).astype(np.float32)
        if s % 5 == 3:
            act[:] = 0.0  # (stand_still / feet_contact_without_cmd style gates need quiet joints now and then)
        emu_load_state(b, emu_read_state(a))
        a.step(act.ctypes.data); b.step(act.ctypes.data)
        for name in ("TERMINATED", "TIME_OUT"):
            assert np.array_equal(host_view(a, name), host_view(b, name)), (task, s, name)
        for name in ("OBS_POLICY", "OBS_CRITIC"):  # one step from a shared state: round-off of the kinematics' two forms (bit equal where no Spec joint
            xa, xb = host_view(a, name).astype(np.float64), host_view(b, name).astype(np.float64)  # is axis-aligned), amplified by a stiff contact in a few
            err = np.abs(xa - xb) / np.maximum(np.abs(xb), 1.0)                                     # entries: 92 - 98 % are bit equal, the worst ~2e-4
            assert err.max() <= 2e-3, (task, s, name, err.max())
            obs_err.append(err.ravel())
        ta, tb = host_view(a, "REWARD_TERMS")[:, :N].astype(np.float64), host_view(b, "REWARD_TERMS")[:, :N].astype(np.float64)
        seen |= (tb != 0).any(axis=1)
        # a term is a sum of <= 30 fp32 products in another order on a state that differs by one step's round-off (joint_acc_l2 squares a
        # finite difference of the velocities: 2 / dt = 400 x their error): relative 5e-4 of the term, absolute floor 1e-6
        assert np.all(np.abs(ta - tb) <= 5e-4 * np.abs(tb) + 1e-6), (task, s, np.abs(ta - tb).max(axis=1))  # (floor: 0.01 % of a typical step reward)
        ra, rb = host_view(a, "REWARD").astype(np.float64), host_view(b, "REWARD").astype(np.float64)
        assert np.all(np.abs(ra - rb) <= 5e-4 * np.abs(tb).sum(axis=0) + 2e-6), (task, s)
    assert np.quantile(np.concatenate(obs_err), 0.99) <= 2e-5, (task, np.quantile(np.concatenate(obs_err), [0.5, 0.9, 0.99, 1.0]))
    return seen


@pytest.mark.parametrize("sid", sorted(SPECS))
def test_specialised_program_equals_interpreter_one_lane_per_limb(sid, emu_lib, monkeypatch):
    monkeypatch.setenv("RL_EMU_FIBERS", "1")  # deterministic: the trunk + limbs instance adds into shared words (thread order otherwise)
    monkeypatch.delenv("RL_EMU_SUB", raising=False)
    seen = spec_vs_interpreter(SPECS[sid], 8, 30, emu_lib, monkeypatch, sid)
    assert seen.sum() >= len(seen) - 3, f"only {seen.sum()} of {len(seen)} terms were ever non-zero: the comparison is too quiet"


@pytest.mark.parametrize("sid,sub", [(1, 4), (1, 2), (2, 4), (3, 4), (4, 8), (5, 4), (6, 4), (7, 4), (8, 8)])
def test_specialised_program_equals_interpreter_in_its_baseline_mapping(sid, sub, emu_lib, monkeypatch):
    monkeypatch.setenv("RL_EMU_FIBERS", "1")
    monkeypatch.setenv("RL_EMU_SUB", str(sub))
    spec_vs_interpreter(SPECS[sid], 4, 14, emu_lib, monkeypatch, sid)


@pytest.mark.parametrize("sid,sub", [(1, 4), (5, 4), (3, 4)])
def test_play_variant_runs_its_tasks_spec(sid, sub, emu_lib, monkeypatch):
    """play.py's edits (no observation corruption): the specialised observation stage reads the flag at run time and equals the interpreter."""
    monkeypatch.setenv("RL_EMU_FIBERS", "1")
    monkeypatch.setenv("RL_EMU_SUB", str(sub))

    def play(desc):
        desc.task.policy_corrupt = 0
        desc.task.ev_push = 0

    spec_vs_interpreter(SPECS[sid], 4, 10, emu_lib, monkeypatch, sid, play)
