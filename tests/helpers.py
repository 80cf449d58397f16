"""Shared helpers for the parity tests (test infrastructure)."""
import ctypes

import numpy as np

from oracle.env import OracleEnv
from robot_lab_amd.scene import build_world, load_bundle


def make_pair(task, N, seed, lib_path=None, device=0):
    """(oracle env, native env driven through the C-ABI) on the same descriptor, world and seed."""
    from robot_lab_amd.capi import NativeEnv

    desc, extra = load_bundle(task)
    h, to, eo = build_world(desc, extra, N, 0)
    ora = OracleEnv(desc, h, to, N, seed, eo)
    nat = NativeEnv(desc, h, to, eo, N, seed, device, lib_path)
    for name in ("CONTACT_FORCE", "JOINT_TORQUE", "JOINT_ACC"):  # inspection views: filled by the steps after the first request
        nat.buffer(name)
    return desc, ora, nat


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
