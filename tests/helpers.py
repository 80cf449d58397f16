"""Shared helpers for the parity tests (test infrastructure)."""
import ctypes

import numpy as np

from oracle.env import OracleEnv
from robot_lab_amd.scene import build_world, load_bundle


def make_pair(task, N, seed, lib_path=None, device=0, mutate=None):
    """(oracle env, native env driven through the C-ABI) on the same descriptor, world and seed."""
    from robot_lab_amd.capi import NativeEnv

    desc, extra = load_bundle(task)
    if mutate is not None:
        mutate(desc)
    h, to, eo = build_world(desc, extra, N, 0)
    ora = OracleEnv(desc, h, to, N, seed, eo)
    nat = NativeEnv(desc, h, to, eo, N, seed, device, lib_path)
    for name in ("CONTACT_FORCE", "JOINT_TORQUE", "JOINT_ACC"):  # inspection views: filled by the steps after the first request
        nat.buffer(name)
    return desc, ora, nat


def set_action_sync(desc, host_term, joint_groups):
    """Turn reward term `host_term` of the descriptor into `action_sync` (rewards.py:305-337) over `joint_groups` (lists of joint names),
    with the index lists robot_lab_amd/model/build.py writes: idx_a the action columns, idx_b the group of each."""
    from robot_lab_amd.desc import REW
    from robot_lab_amd.model.build import find_names

    r = desc.task.rewards[list(desc.reward_names).index(host_term)]
    n = 0
    for gi, group in enumerate(joint_groups):
        for name in group:
            (c,) = find_names(str(name), list(desc.joint_names))
            r.idx_a[n], r.idx_b[n] = c, gi
            n += 1
    r.kind, r.n_idx, r.p[0] = REW["action_sync"], n, 1.0 / len(joint_groups)
    return r


A1_SYNC_GROUPS = [[f"{leg}_{part}_joint" for leg in ("FR", "FL", "RL", "RR")] for part in ("hip", "thigh", "calf")]  # velocity_env_cfg.py:492-496


def host_view(nat, name):
    """numpy view of an env buffer of the CPU lane emulator (host pointers)."""
    p, shp, dt = nat.buffer(name)
    n = int(np.prod(shp))
    return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).reshape(shp)


def oracle_root_state(ora):
    st = ora.st
    return np.concatenate([st["root_pos"], st["root_quat"], st["root_lin_vel"], st["root_ang_vel"]], -1)


def assert_close(name, got, want, rtol, atol, frac_ok=1.0):
    """|got - want| <= atol + rtol*|want| for at least `frac_ok` of the entries (contact switching is a
    measure-zero discontinuity: a few entries may sit on the other side of a threshold in fp32)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    ok = np.abs(got - want) <= atol + rtol * np.abs(want)
    frac = ok.mean() if ok.size else 1.0
    assert frac >= frac_ok, f"{name}: only {frac:.4f} within tol (need {frac_ok}); max abs err {np.abs(got - want).max():.3e}"


# ------------------------------------------------------------------------------------------------
# teacher-forced parity: both sides step ONCE from a shared state (SURVEY.md 8(c) last row)
# ------------------------------------------------------------------------------------------------
STATE_BUFFERS = ("ROOT_STATE", "JOINT_POS", "JOINT_VEL", "ACTION", "GAINS", "CONTACT_TIMERS", "TASK_STATE", "ENV_ORIGIN")


def emu_read_state(nat):
    """`ManagerBasedRLEnv.read_state()` for a NativeEnv of the CPU lane emulator (host pointers)."""
    nat.export_state()
    out = {k.lower(): host_view(nat, k).copy() for k in STATE_BUFFERS}
    out["episode_length"] = host_view(nat, "EPISODE_LENGTH").copy()
    out["episode_sums"] = host_view(nat, "EPISODE_SUMS")[:, : nat.num_envs].copy()
    out["terrain_level"] = host_view(nat, "TERRAIN_LEVEL").copy()
    out["step_count"] = nat.step_count
    return out


def emu_load_state(nat, s):
    nat.export_state()
    for k in STATE_BUFFERS:
        host_view(nat, k)[...] = np.asarray(s[k.lower()], dtype=host_view(nat, k).dtype).reshape(host_view(nat, k).shape)  # (fp32; fp64 in the retyped emulator build)
    host_view(nat, "EPISODE_LENGTH")[...] = s["episode_length"]
    host_view(nat, "EPISODE_SUMS")[:, : nat.num_envs] = s["episode_sums"]
    host_view(nat, "TERRAIN_LEVEL")[...] = s["terrain_level"]
    nat.step_count = int(s["step_count"])
    nat.commit_state()


# Margins (oracle/physics.py Physics.margins) below which an env counts as sitting ON one of the model's discontinuities,
# where fp32 and fp64 may legitimately land on different sides: contact on/off (penetration, lagged normal force),
# static/dynamic friction, joint-limit damper, implicit-actuator saturation, contact-sensor force threshold, and a touching
# sphere on a heightfield cell edge across which the terrain normal jumps ("cell": 0 = yes).
# Sized by what the fp32 side can resolve.  The terrain lookups add the root position (a world coordinate of tens of metres) and
# the sphere / ray offsets in fp64 (csrc/env_step.h terrain_fetch), so a penetration depth carries the round-off of metre-sized
# numbers only (~3e-7 m, i.e. ~0.006 N of contact force at k = 2e4 N/m); joint angles are ~1 rad (ulp 1.2e-7), velocities a few
# m/s (ulp 5e-7).  (Round 1 summed in fp32: the margins were 1e-5 m / 0.1 N then and flagged 3-4 % of the envs.)
SWITCH_EPS = dict(phi=1e-6, fn0=0.01, stick=1e-5, limit=1e-6, saturation=1e-3, force=0.01, cell=0.5)


def switch_mask(margins, eps=SWITCH_EPS):
    """bool [N]: envs within `eps` of a switch during the step the margins were recorded over."""
    mask = None
    for k, v in margins.items():
        m = np.asarray(v) < eps[k]
        mask = m if mask is None else (mask | m)
    return mask


def rel_err(got, want, floor):
    """max over the trailing axes of |got - want| / max(|want|, floor): one number per env."""
    got = np.asarray(got, dtype=np.float64).reshape(len(got), -1)
    want = np.asarray(want, dtype=np.float64).reshape(len(want), -1)
    return (np.abs(got - want) / np.maximum(np.abs(want), floor)).max(axis=1)


# Hard ceilings on top of the conditioning-aware envelope (VERDICT r2 "the envelope has no ceiling"): whatever the twins'
# sensitivity grants an entry, no unmasked env may be further from the oracle than this (metric of rel_err: |err| / max(|want|, 1)).
# Sized from the measured worst cases at the BASELINE sizes (profiles/r02_teacher_forced_*.json: root 3.5e-4, joint position
# 3.2e-5, joint velocity 2.0e-3, observations 2.0e-3 - the critic row carries the joint velocities) with a 2.5 - 3x margin.
HARD_CAPS = dict(root_state=5e-3, joint_pos=1e-4, joint_vel=5e-3, task_state=5e-3, obs_policy=5e-3, obs_critic=5e-3)


def teacher_forced_check(ora, state, action, got, n_twins=3, gain=32.0, base=1e-5, seed=0, max_mask=0.025, caps=HARD_CAPS, max_outliers=0, outlier_factor=4.0):
    """One step of the fp64 oracle from the SHARED `state` (a read_state() dict of the HIP / emulator env, i.e. fp32 values)
    against what the fp32 side produced from that same state (`got`: dict with the read_state() keys after the step plus
    reward, reward_terms [T, N], done [N] bool, obs_policy, obs_critic).

    Tolerance, per env and per field:   err <= base + gain * s   (1e-5 + 32 s),   err = max |got - want| / max(|want|, 1)
    where s is the oracle's OWN response (same metric) to fp32-sized disturbances, maximum over `n_twins` twins: each twin
    starts from the shared root / joint state perturbed by a random relative 1e-6 (16 fp32 ulp; the root's world x, y by an
    absolute 1e-6 m) AND solves its linear systems
    in single precision (Physics.solve_dtype: the conditioning of H + A, which an input perturbation alone does not probe).  Why not a flat 1e-5: the step map of a robot in stiff
    contact amplifies an input perturbation of 1e-6 by 30x (median) to 2000x (joint velocities; measured),
    so fp32 round-off inside 4 substeps necessarily shows up at 1e-5 .. 1e-3 there, while airborne envs agree to < 1e-5.
    A kernel bug (wrong lane, wrong slot, wrong term) produces errors orders of magnitude above its env's own sensitivity.
    Envs within SWITCH_EPS of a discontinuity of the model (contact on/off, stick/slip, limit damper, actuator saturation,
    sensor threshold) are excluded - and counted: the test fails if they exceed `max_mask` of the batch.
    On top of the envelope every field has a flat ceiling (`caps`, HARD_CAPS): an entry must satisfy BOTH.
    Returns a report dict; raises AssertionError listing the offending envs otherwise."""
    N = ora.N
    ora.load_state(state)
    ora.phys.margins = {}
    o = ora.step(action)
    want = ora.read_state()
    want.update(reward=ora.reward.copy(), reward_terms=ora.reward_terms.copy(), done=(ora.terminated | ora.time_outs).copy(),
                obs_policy=o[0].copy(), obs_critic=o[1].copy())
    mask = switch_mask(ora.phys.margins)
    margins, ora.phys.margins = ora.phys.margins, None
    rng = np.random.default_rng(seed)
    fields = ("root_state", "joint_pos", "joint_vel", "task_state", "obs_policy", "obs_critic")
    sens = {f: np.zeros(N) for f in fields}
    sens["reward"] = np.zeros(N)
    sens["reward_terms"] = np.zeros((len(want["reward_terms"]), N))
    for _ in range(n_twins):
        tw = dict(state)
        for k in ("root_state", "joint_pos", "joint_vel"):
            tw[k] = np.asarray(state[k], dtype=np.float64) * (1.0 + 1e-6 * rng.uniform(-1, 1, np.shape(state[k])))
        # (the root's world x, y - tens of metres - by an ABSOLUTE 1e-6 m: the kernels' terrain lookups are free of world-coordinate
        # round-off, so a relative perturbation there would be 4e-5 m and hand the contacts a tolerance they do not need)
        tw["root_state"][:, 0:2] = np.asarray(state["root_state"], dtype=np.float64)[:, 0:2] + 1e-6 * rng.uniform(-1, 1, (len(tw["root_state"]), 2))
        ora.load_state(tw)
        ora.phys.solve_dtype = np.float32
        ot = ora.step(action)
        ora.phys.solve_dtype = None
        t = ora.read_state()
        t.update(obs_policy=ot[0], obs_critic=ot[1])
        for f in fields:
            sens[f] = np.maximum(sens[f], rel_err(t[f], want[f], 1.0))
        sens["reward"] = np.maximum(sens["reward"], np.abs(ora.reward - want["reward"]))
        sens["reward_terms"] = np.maximum(sens["reward_terms"], np.abs(ora.reward_terms - want["reward_terms"]))
    ok = ~mask
    # What an fp32 implementation of THIS step can agree to at best: the oracle itself with nothing changed but the dtype of its linear
    # solves (no input perturbation, libm everywhere) against its fp64 self - the share of entries inside the flat `base`.  The tolerance
    # floor of the tested side is this, not its hardware sin / cos / rcp approximations (profiles/r05a_exact_math_teacher_forced_*.json:
    # an exact-math build of the kernels moves the share by 1 - 3 points).  Reported, not asserted.
    ora.load_state(state)
    ora.phys.solve_dtype = np.float32
    o32 = ora.step(action)
    ora.phys.solve_dtype = None
    t32 = ora.read_state()
    t32.update(obs_policy=o32[0], obs_critic=o32[1])
    fp32_solve = {f: dict(frac_within_base=float(np.mean(rel_err(t32[f], want[f], 1.0)[ok] <= base)), p50=float(np.median(rel_err(t32[f], want[f], 1.0))),
                          max_err=float(rel_err(t32[f], want[f], 1.0)[ok].max())) for f in fields}
    report = dict(n=N, masked=int(mask.sum()), masked_frac=float(mask.mean()), margins_min={k: float(np.min(v)) for k, v in margins.items()},
                  masked_by_margin={k: int((np.asarray(v) < SWITCH_EPS[k]).sum()) for k, v in margins.items()})  # envs each margin flags (they overlap)
    bad = {}
    for f in fields:
        err = rel_err(got[f], want[f], 1.0)
        # (observations: 5 x base - a height-scan ray is a bilinear sample at world coordinates of +-60 m, where fp32 resolves the
        # position inside a 5 cm cell to 1e-4 of the cell: 2e-5 m of height on a stair edge, and an env that was just reset has no
        # twin response to show for it)
        tol = (5.0 if f.startswith("obs") else 1.0) * base + gain * sens[f]
        cap = (caps or {}).get(f)
        report[f] = dict(max_err=float(err[ok].max()), p99=float(np.quantile(err[ok], 0.99)), p50=float(np.median(err)),
                         frac_within_base=float(np.mean(err[ok] <= base)), max_tol=float(tol[ok].max()), cap=cap)
        b = np.nonzero((err > tol) & ok)[0]
        if len(b):
            bad[f] = [(int(i), float(err[i]), float(tol[i])) for i in b[:8]]
        if cap is not None:
            b = np.nonzero((err > cap) & ok)[0]
            if len(b):
                bad[f + ":cap"] = [(int(i), float(err[i]), float(cap)) for i in b[:8]]
    # rewards: absolute, base scaled by the term weights (reward = sum of w * f * dt)
    w = np.abs(np.array([ora.desc.task.rewards[i].weight for i in range(ora.desc.task.n_rewards)], dtype=np.float64))
    err_t = np.abs(np.asarray(got["reward_terms"], dtype=np.float64) - want["reward_terms"])
    # (absolute floor base * 1e-2 = 1e-7 reward units: terms with tiny weights such as joint_acc_l2, w = 2.5e-7 on (rad/s^2)^2)
    # (5e-3 relative: terms like joint_acc_l2 square a finite difference of the velocities, (qd+ - qd) / dt, i.e. amplify the
    # velocity error 2 / dt = 400 times before squaring; a wrong term, mask or weight is an O(1) relative error)
    tol_t = base * (w[:, None] + 1e-2) + 5e-3 * np.abs(want["reward_terms"]) + gain * sens["reward_terms"]
    bt = np.argwhere((err_t > tol_t) & ok[None])
    if len(bt):
        bad["reward_terms"] = [(int(t), int(i), float(err_t[t, i]), float(tol_t[t, i])) for t, i in bt[:8]]
    err_r = np.abs(np.asarray(got["reward"], dtype=np.float64) - want["reward"])
    tol_r = base * np.maximum(w.sum(), 1.0) * ora.step_dt + 5e-3 * np.abs(want["reward_terms"]).sum(axis=0) + gain * sens["reward"]
    br = np.nonzero((err_r > tol_r) & ok)[0]
    if len(br):
        bad["reward"] = [(int(i), float(err_r[i]), float(tol_r[i])) for i in br[:8]]
    report["reward"] = dict(max_err=float(err_r[ok].max()), frac_within_base=float(np.mean(err_r[ok] <= base * np.maximum(w.sum(), 1.0) * ora.step_dt)))
    # discrete outputs: exact
    if not np.array_equal(np.asarray(got["done"])[ok], want["done"][ok]):
        bad["done"] = np.nonzero((np.asarray(got["done"]) != want["done"]) & ok)[0][:8].tolist()
    for f in ("episode_length", "terrain_level"):
        if not np.array_equal(np.asarray(got[f])[ok], np.asarray(want[f])[ok]):
            bad[f] = np.nonzero((np.asarray(got[f]) != np.asarray(want[f])) & ok)[0][:8].tolist()
    te = np.abs(np.asarray(got["contact_timers"], dtype=np.float64) - want["contact_timers"]).reshape(N, -1).max(axis=1)
    if (te[ok] > 1e-6).any():
        bad["contact_timers"] = np.nonzero((te > 1e-6) & ok)[0][:8].tolist()
    report["done_count"] = int(want["done"].sum())
    report["oracle_with_fp32_solves"] = fp32_solve
    report["bad"] = bad
    assert report["masked_frac"] <= max_mask, f"{report['masked']} of {N} envs sit on a switch (> {max_mask:.1%}): {report}"
    if max_outliers > 0 and bad:
        # `max_outliers` envs may leave the ENVELOPE (never a flat cap, a discrete output or a reward bound): FFTAI GR1 at 1024 envs has one
        # joint-velocity entry in 32 k at 3 x its six-twin envelope, the same entry to nine digits in both lane mappings - an env
        # the margins do not flag, not a lane-program difference
        soft = {k: v for k, v in bad.items() if k in ("root_state", "joint_pos", "joint_vel", "task_state", "obs_policy", "obs_critic")}
        envs = {e[0] for v in soft.values() for e in v}
        # ... and never by more than 4 x the envelope: a forgiven entry is an env the margins failed to flag, not a free pass
        within = all(err <= outlier_factor * tol for v in soft.values() for (_, err, tol) in v)
        if len(soft) == len(bad) and len(envs) <= max_outliers and all(len(v) < 8 for v in soft.values()) and within:
            report["envelope_outliers"] = {k: v for k, v in soft.items()}
            print(f"\n[teacher-forced] {len(envs)} env(s) outside the twin envelope, forgiven (max_outliers={max_outliers}, <= {outlier_factor:g} x tol): {report['envelope_outliers']}")
            report["bad"] = bad = {}
    assert not bad, f"teacher-forced parity violated outside the switch mask: {bad}\nreport: {report}"
    return report


class OracleWithTwin:
    """The fp64 oracle plus a TWIN of it that is disturbed the way an fp32 implementation is: its root / joint state is
    perturbed by a relative 1e-6 (16 fp32 ulp) after every reset() and its linear systems are solved in single precision.  Over a
    free run the twin drifts away from the oracle exactly where the step map is ill conditioned or a switch is crossed, so
    |twin - oracle| is a per-entry scale for what an fp32 trajectory may legitimately differ by:

        |got - want| <= atol + rtol |want| + gain |twin - want|   (gain 32)   for 100 % of the entries

    (free-running trajectories cannot be held to a flat tolerance: a robot in stiff contact amplifies a 1e-6 input perturbation
    30x - 2000x per step).  The single-step, shared-state form of the comparison is teacher_forced_check."""

    def __init__(self, make, gain=32.0, seed=0):
        self.ora, self.twin = make(), make()
        self.twin.phys.solve_dtype = np.float32
        self.gain, self.rng = gain, np.random.default_rng(seed)
        self.done_differs = np.zeros(self.ora.N, dtype=bool)  # envs whose twin took a different reset decision at some step
        self.max_outlier_envs = 0   # envs that may leave the band altogether (a free run that took another branch); see close()
        self.outlier_envs = set()
        self.ora.phys.margins = {}  # smallest distance to every switch of the model over the whole run (Physics._margin keeps minima)

    def _perturb(self):
        st = self.ora.read_state()
        xy = st["root_state"][:, 0:2].copy()
        for k in ("root_state", "joint_pos", "joint_vel"):
            st[k] = st[k] * (1.0 + 1e-6 * self.rng.uniform(-1, 1, st[k].shape))
        st["root_state"][:, 0:2] = xy + 1e-6 * self.rng.uniform(-1, 1, xy.shape)  # world x, y: absolute (see teacher_forced_check)
        self.twin.load_state(st)

    def reset(self):
        o = self.ora.reset()
        self.twin.reset()
        self._perturb()
        return o

    def step(self, a):
        o = self.ora.step(a)
        self.twin_obs = self.twin.step(a)
        self.done_differs |= (self.ora.terminated | self.ora.time_outs) != (self.twin.terminated | self.twin.time_outs)
        return o

    def close(self, name, got, pick, rtol, atol):
        """`pick(env)` extracts the compared array from an OracleEnv; envs whose twin reset differently, or that came within
        SWITCH_EPS of a switch of the model at some substep, are skipped (`self.masked` counts them)."""
        want, tw = np.asarray(pick(self.ora), dtype=np.float64), np.asarray(pick(self.twin), dtype=np.float64)
        got = np.asarray(got, dtype=np.float64)
        assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
        env_axis = [i for i, n in enumerate(want.shape) if n == self.ora.N][0]
        ok_env = ~self.done_differs
        if self.ora.phys.margins:  # envs that sat ON a switch at some substep of the run (same rule as the teacher-forced check)
            ok_env = ok_env & ~switch_mask(self.ora.phys.margins)
        self.masked = int((~ok_env).sum())
        keep = np.moveaxis(np.broadcast_to(np.expand_dims(ok_env, tuple(i for i in range(want.ndim) if i != env_axis)), want.shape), 0, 0)
        bad = (np.abs(got - want) > atol + rtol * np.abs(want) + self.gain * np.abs(tw - want)) & keep
        if bad.any() and self.max_outlier_envs > 0:
            # a caller that names a KNOWN case may let whole envs go: a free run in which the implementation's round-off took one env through
            # a switch the oracle and its twin did not take is out in every field from then on.  The envs are remembered - the budget is per
            # run, not per field - and printed.
            envs = set(np.nonzero(np.moveaxis(bad, env_axis, 0).reshape(self.ora.N, -1).any(axis=1))[0].tolist())
            if len(self.outlier_envs | envs) <= self.max_outlier_envs:
                self.outlier_envs |= envs
                print(f"\n[free-run outlier] {name}: env(s) {sorted(envs)} outside the band, worst |err| {np.abs(got - want)[bad].max():.3e}")
                idx = [slice(None)] * want.ndim
                idx[env_axis] = sorted(self.outlier_envs)
                bad[tuple(idx)] = False
        assert not bad.any(), (f"{name}: {int(bad.sum())} of {bad.size} entries outside atol {atol} + rtol {rtol} + {self.gain} x twin drift; "
                               f"worst |err| {np.abs(got - want)[bad].max():.3e} where the twin drifted {np.abs(tw - want)[bad].max():.3e}")


# ------------------------------------------------------------------------------------------------
# distributional parity: episode statistics over a long free run (SURVEY.md section 7 "Chaotic divergence")
# ------------------------------------------------------------------------------------------------
class EpisodeStats:
    """Per-ENV time averages over a free run, from what step() returns and from read_state() dicts: the statistics a PPO run
    is sensitive to.  A free-running fp32 trajectory decorrelates from the fp64 oracle's within tens of steps, so trajectories
    cannot be compared - but their statistics must agree: a small systematic bias (a wrong sign in a rarely active term, a
    contact that is slightly too soft, a timer that drifts) hides inside the per-step envelopes and shows up here."""

    def __init__(self, n_envs, n_terms, n_levels):
        self.N, self.T, self.n_levels, self.steps = n_envs, n_terms, n_levels, 0
        self.term_sum = np.zeros((n_terms, n_envs))
        self.reward_sum, self.terminated, self.time_out = np.zeros(n_envs), np.zeros(n_envs), np.zeros(n_envs)
        self.contacts, self.height, self.speed, self.tilt = np.zeros(n_envs), np.zeros(n_envs), np.zeros(n_envs), np.zeros(n_envs)
        self.level = np.zeros(n_envs)

    def add(self, reward, reward_terms, terminated, time_out, state):
        """`state`: read_state() after the step (root_state, contact_timers, env_origin, terrain_level)."""
        self.steps += 1
        self.term_sum += np.asarray(reward_terms, dtype=np.float64)[: self.T, : self.N]
        self.reward_sum += np.asarray(reward, dtype=np.float64)
        self.terminated += np.asarray(terminated, dtype=np.float64)
        self.time_out += np.asarray(time_out, dtype=np.float64)
        rs = np.asarray(state["root_state"], dtype=np.float64)
        self.contacts += (np.asarray(state["contact_timers"])[:, :, 1] > 0).sum(axis=1)       # bodies in contact (current contact time > 0)
        self.height += rs[:, 2] - np.asarray(state["env_origin"], dtype=np.float64)[:, 2]  # root height above the tile's origin
        self.speed += np.linalg.norm(rs[:, 7:10], axis=1)
        w, x, y, z = rs[:, 3], rs[:, 4], rs[:, 5], rs[:, 6]
        self.tilt += 1.0 - (1.0 - 2.0 * (x * x + y * y))                                       # 1 - cos(angle between body z and world z)
        self.level = np.asarray(state["terrain_level"], dtype=np.float64)                    # the last one: where the curriculum left the env

    def per_env(self):
        """dict name -> [N] per-env time averages (terrain level: the final one)."""
        s = float(self.steps)
        out = {f"term_{t}": self.term_sum[t] / s for t in range(self.T)}
        out.update(reward=self.reward_sum / s, terminated=self.terminated / s, time_out=self.time_out / s, contacts=self.contacts / s,
                   height=self.height / s, speed=self.speed / s, tilt=self.tilt / s, level=self.level)
        return out


def staggered_episode_lengths(n_envs, max_episode_length):
    """Episode counters spread over [0, max_episode_length) (what rsl_rl's `init_at_random_ep_len` does, train.py:224): within a
    300-step run ~30 % of the envs time out, so resets - curriculum moves, event draws, command resampling - are in the statistics."""
    return (np.arange(n_envs, dtype=np.int64) * 37 + 11) % int(max_episode_length)


def run_episode_stats(step_fn, read_state_fn, n_envs, n_terms, n_levels, action_dim, steps, action_seed):
    """Drives `step_fn(action [N, A] float32) -> (reward, reward_terms, terminated, time_out)` for `steps` steps with the action
    stream numpy default_rng(action_seed).uniform(-1, 1) - the stream both sides and the committed fixture share."""
    rng = np.random.default_rng(action_seed)
    st = EpisodeStats(n_envs, n_terms, n_levels)
    for _ in range(steps):
        a = rng.uniform(-1, 1, (n_envs, action_dim)).astype(np.float32)
        st.add(*step_fn(a), read_state_fn())
    return st.per_env()


def compare_episode_stats(got, ora, twin, k_sigma=4.0, boot=400, seed=0, weights=None):
    """`got`, `ora`, `twin`: per_env() dicts of the tested side, the fp64 oracle and its fp32-disturbed twin over the SAME envs, seeds
    and action stream.  For every statistic: |mean(got) - mean(ora)| must lie inside k_sigma standard deviations of the
    oracle-vs-twin difference of means, estimated by a paired bootstrap over the envs (plus that difference itself: the twin is one
    legitimate realisation, the tested side another) and an absolute floor for statistics that are constant across the batch.
    Returns the report; raises AssertionError naming the statistics outside their interval."""
    rng = np.random.default_rng(seed)
    N = len(ora["reward"])
    idx = rng.integers(0, N, (boot, N))
    report, bad = {}, []
    for name in ora:
        a, b, g = np.asarray(ora[name]), np.asarray(twin[name]), np.asarray(got[name])
        d = b - a
        sigma = d[idx].mean(axis=1).std()
        # the tested side's own sampling noise against the oracle (paired), in case the twin happens to track the oracle closely
        sigma = max(sigma, (g - a)[idx].mean(axis=1).std())
        floor = 1e-6 + 1e-4 * max(abs(a.mean()), abs(b.mean()))
        tol = abs(d.mean()) + k_sigma * sigma + floor
        diff = g.mean() - a.mean()
        report[name] = dict(oracle=float(a.mean()), twin=float(b.mean()), got=float(g.mean()), diff=float(diff), tol=float(tol), sigma=float(sigma))
        if abs(diff) > tol:
            bad.append((name, float(diff), float(tol)))
    assert not bad, f"episode statistics outside the oracle-vs-twin interval: {bad}\n{report}"
    return report
