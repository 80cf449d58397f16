"""Body of tests/test_fp64_lane_program.py (its own process: RL_ABI_REAL=f64 switches the ctypes mirror of the C-ABI to 8-byte reals).

The lane program retyped to double (tests/emu/make_f64.py) and the fp64 oracle take ONE step from a shared state that the lane
program itself reached after K random-action steps, made eventful the way tests/test_teacher_forced.py does (time-out resets, interval push,
command resampling, a terrain-level promotion, an out-of-bounds env).  Both sides now compute in double precision, so whatever separates
them is not round-off: a lagged term, another linearisation point, a different constant.  Prints one JSON report.

usage: RL_ABI_REAL=f64 python tests/fp64_lane_program.py <task> <num_envs> <warmup steps | -1: the on-the-ground scenario> [sub]"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE)]

from helpers import emu_load_state, emu_read_state, host_view, make_pair, rel_err, switch_mask  # noqa: E402

RL_TS_CMD_TIME_LEFT, RL_TS_PUSH_TIME_LEFT = 4, 7


def outputs(nat, N):
    got = emu_read_state(nat)
    got.update(reward=host_view(nat, "REWARD").copy(), reward_terms=host_view(nat, "REWARD_TERMS")[:, :N].copy(),
               done=host_view(nat, "TERMINATED").astype(bool) | host_view(nat, "TIME_OUT").astype(bool),
               obs_policy=host_view(nat, "OBS_POLICY").copy(), obs_critic=host_view(nat, "OBS_CRITIC").copy())
    return got


def compare(desc, ora, nat, state, a, N, rep):
    """One step of both sides from `state` with action `a`; fills rep["fields"] / the exact flags with the worst case so far."""
    nat.step(a.ctypes.data)
    got = outputs(nat, N)
    ora.load_state(state)
    ora.phys.margins = {}
    o = ora.step(a)
    want = ora.read_state()
    want.update(reward=ora.reward.copy(), reward_terms=ora.reward_terms.copy(), done=(ora.terminated | ora.time_outs).copy(),
                obs_policy=o[0].copy(), obs_critic=o[1].copy())
    mask = switch_mask(ora.phys.margins)
    ok = ~mask
    rep["masked"] = max(rep.get("masked", 0), int(mask.sum()))
    rep["done_count"] = rep.get("done_count", 0) + int(want["done"].sum())
    for f in ("root_state", "joint_pos", "joint_vel", "task_state", "gains", "contact_timers", "episode_sums", "obs_policy", "obs_critic", "reward"):
        g, w = np.asarray(got[f], np.float64), np.asarray(want[f], np.float64)
        if f == "episode_sums":
            g, w = g.T, w.T
        if f == "reward":
            g, w = g[:, None], w[:, None]
        err = rel_err(g, w, 1.0)
        old = rep["fields"].get(f, dict(max_err=0.0))
        if float(err[ok].max()) >= old["max_err"]:
            rep["fields"][f] = dict(max_err=float(err[ok].max()), p50=float(np.median(err[ok])), worst_env=int(np.argmax(np.where(ok, err, -1.0))),
                                    max_err_masked_too=float(err.max()))
    et = np.abs(np.asarray(got["reward_terms"], np.float64) - want["reward_terms"])
    if float(et[:, ok].max()) >= rep["fields"].get("reward_terms", dict(max_err=0.0))["max_err"]:
        rep["fields"]["reward_terms"] = dict(max_err=float(et[:, ok].max()), worst_term=int(np.argmax(et[:, ok].max(axis=1))))
    cf = np.abs(np.asarray(host_view(nat, "CONTACT_FORCE"), np.float64) - ora.contact_force).reshape(N, -1).max(axis=1)
    rep["fields"]["contact_force_abs"] = dict(max_err=max(float(cf[ok].max()), rep["fields"].get("contact_force_abs", dict(max_err=0.0))["max_err"]))
    rep["done_equal"] = rep.get("done_equal", True) and bool(np.array_equal(np.asarray(got["done"])[ok], want["done"][ok]))
    rep["episode_length_equal"] = rep.get("episode_length_equal", True) and bool(np.array_equal(np.asarray(got["episode_length"])[ok], np.asarray(want["episode_length"])[ok]))
    rep["terrain_level_equal"] = rep.get("terrain_level_equal", True) and bool(np.array_equal(np.asarray(got["terrain_level"])[ok], np.asarray(want["terrain_level"])[ok]))
    return got, want


def on_the_ground(task, N, sub):
    """Robots thrown on their backs and on their faces, pressed 1 - 20 cm into the ground, terminations off so that nobody resets and
    the state AFTER the physics is what is compared: the trunk links' collision spheres carry the robot - contacts a random-action
    warm-up rarely reaches and a termination term hides (the env is reset in the step that makes them).  Three consecutive steps, each
    from the lane program's own state.  (Round 6: this is where the lane program of the trunk + limbs instances and the oracle differed by
    ~1 % - csrc/env_step.h substep_aba_trunk carried limb link 0's rotational inertia onto a trunk link whose spheres touched the ground.)"""
    from robot_lab_amd.desc import REAL_NP

    def no_terminations(d):
        d.task.term_illegal_contact = 0

    lib = os.path.join(HERE, "emu", "librl_env_emu_f64.so")
    desc, ora, nat = make_pair(task, N, 5, lib, mutate=no_terminations)
    nat.reset()
    rng = np.random.default_rng(3)
    for _ in range(2):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(REAL_NP)
        nat.step(a.ctypes.data)
    state = emu_read_state(nat)
    rs = state["root_state"].copy()
    ground = ora.phys.terrain.sample(rs[:, 0].astype(np.float64), rs[:, 1].astype(np.float64))[0]
    M = 3 * N // 4
    rs[:M, 2] = (ground + np.linspace(0.015, 0.2, N))[:M]
    rs[: M // 2, 3:7] = [0.0, 1.0, 0.0, 0.0]                                    # on its back
    rs[M // 2: M, 3:7] = [0.7071067811865476, 0.0, 0.7071067811865476, 0.0]     # pitched 90 degrees: face down
    rs[:M, 7:13] *= 0.1
    state["root_state"] = rs
    emu_load_state(nat, state)
    rep = dict(task=task, n=N, scenario="on_the_ground", lanes_per_limb=sub, spec_id=nat.spec_id(), real_bytes=int(np.dtype(REAL_NP).itemsize), fields={})
    m = desc.model
    trunk_links = {0} | {int(m.trunk_link[i]) for i in range(m.num_trunk)}
    trunk_bodies = [b for b in range(m.num_bodies) if int(m.body_link[b]) in trunk_links]  # bodies on the base / trunk links
    trunk_loaded = 0
    for _ in range(3):
        state = emu_read_state(nat)
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(REAL_NP)
        compare(desc, ora, nat, state, a, N, rep)
        trunk_loaded = max(trunk_loaded, int((np.linalg.norm(ora.contact_force[:, trunk_bodies], axis=2) > 1.0).any(axis=1).sum()))
    rep["envs_with_a_trunk_body_loaded"] = trunk_loaded
    nat.close()
    print("FP64_REPORT " + json.dumps(rep))


def main(task, N, K, sub):
    from robot_lab_amd.desc import REAL_F64, REAL_NP

    assert REAL_F64, "run with RL_ABI_REAL=f64"
    os.environ["RL_EMU_SUB"] = str(sub)
    os.environ["RL_EMU_FIBERS"] = "1"
    os.environ.setdefault("RL_ENV_SPEC", "0")  # (RL_ENV_SPEC=1 from outside: the task-specialised lane program, where the retyped tables still equal its constants)
    if K < 0:
        return on_the_ground(task, N, sub)
    lib = os.path.join(HERE, "emu", "librl_env_emu_f64.so")
    desc, ora, nat = make_pair(task, N, 42, lib)
    nat.reset()
    rng = np.random.default_rng(1)
    ep = rng.integers(0, nat.max_episode_length, N)
    ep[::5] = nat.max_episode_length - 1 - (np.arange(len(ep[::5])) % (K + 2))
    ep[1:4] = nat.max_episode_length - 1 - K
    host_view(nat, "EPISODE_LENGTH")[:] = ep
    for _ in range(K):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(REAL_NP)
        nat.step(a.ctypes.data)
    state = emu_read_state(nat)
    ts = state["task_state"].copy()
    ts[2::7, RL_TS_PUSH_TIME_LEFT] = 0.015
    ts[4::9, RL_TS_CMD_TIME_LEFT] = 0.015
    state["task_state"] = ts
    td, tk = desc.terrain, desc.task
    rs = state["root_state"].copy()
    up, oob = 1, 6
    if not td.is_plane and N > 6:
        rs[up, 0] += 4.5
        rs[oob, 0] = 0.5 * (td.num_rows * td.tile_size + 2 * td.border) - tk.oob_buffer + 0.5
        for i in (up, oob):
            rs[i, 2] = ora.phys.terrain.sample(rs[i, 0:1].astype(np.float64), rs[i, 1:2].astype(np.float64))[0][0] + 0.45
            rs[i, 3:7] = [1.0, 0.0, 0.0, 0.0]
    state["root_state"] = rs
    emu_load_state(nat, state)
    state = emu_read_state(nat)
    a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(REAL_NP)
    rep = dict(task=task, n=N, warmup=K, scenario="eventful_step", lanes_per_limb=sub, spec_id=nat.spec_id(), real_bytes=int(np.dtype(REAL_NP).itemsize), fields={})
    compare(desc, ora, nat, state, a, N, rep)
    nat.close()
    print("FP64_REPORT " + json.dumps(rep))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 1)
