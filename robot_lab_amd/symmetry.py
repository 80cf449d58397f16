"""Symmetry data augmentation on MI355X (`include/rl_rollout.h`: `rl_symmetry_*`, `csrc/rl_rollout.hip`).

Host-side mirror of the reference's `compute_symmetric_states(env, obs, actions)`
(`source/robot_lab/robot_lab/tasks/manager_based/locomotion/velocity/mdp/symmetry/anymal.py:27-87`), the function an
rsl_rl PPO configured with `RslRlSymmetryCfg(use_data_augmentation=True, data_augmentation_func=...)` calls on every
mini-batch (`.../config/quadruped/anymal_d/agents/rsl_rl_ppo_cfg.py:100-105`).  Each of the four copies (identity,
left-right, front-back, diagonal) is a signed column permutation; the tables are built here from the joint names and the
observation layout, the copies are produced by one HIP kernel launch.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import re

import numpy as np

from .rollout import RlRolloutError, load_rollout_library

SYMMETRY_EXPORTS = ["rl_symmetry_create", "rl_symmetry_apply", "rl_symmetry_destroy"]
# ANYmal articulation order (anymal.py:216-229)
ANYMAL_JOINTS = ["LF_HAA", "LH_HAA", "RF_HAA", "RH_HAA", "LF_HFE", "LH_HFE", "RF_HFE", "RH_HFE", "LF_KFE", "LH_KFE", "RF_KFE", "RH_KFE"]
# what a mirror does to the components of a base-frame quantity: (left-right, front-back)
_VEC = dict(lin=((1, -1, 1), (-1, 1, 1)),       # polar vectors: linear velocity, projected gravity
            ang=((-1, 1, -1), (1, -1, -1)),     # axial vectors: angular velocity
            cmd=((1, -1, -1), (-1, 1, -1)))     # (v_x, v_y, omega_z) velocity command


def joint_tables(joint_names, pattern=r"(?P<side>[LR])(?P<end>[FH])_(?P<kind>\w+)", roll_kinds=("HAA",)):
    """(perm, sign) of the left-right and the front-back mirror on a per-joint vector: out[j] = sign[j] * in[perm[j]].
    Left-right swaps the L and R legs and flips the roll (abduction) joints; front-back swaps the F and H legs and flips
    the pitch joints (everything that is not a roll joint)."""
    parsed = []
    for n in joint_names:
        m = re.fullmatch(pattern, n)
        if m is None:
            raise ValueError(f"joint name {n!r} does not match {pattern!r}")
        parsed.append((m["side"], m["end"], m["kind"]))
    index = {p: i for i, p in enumerate(parsed)}
    lr_p, lr_s, fb_p, fb_s = [], [], [], []
    for side, end, kind in parsed:
        lr_p.append(index[("R" if side == "L" else "L", end, kind)])
        lr_s.append(-1.0 if kind in roll_kinds else 1.0)
        fb_p.append(index[(side, "H" if end == "F" else "F", kind)])
        fb_s.append(1.0 if kind in roll_kinds else -1.0)
    return (np.array(lr_p), np.array(lr_s)), (np.array(fb_p), np.array(fb_s))


def layout_tables(layout, joint_names=ANYMAL_JOINTS, **kw):
    """perm [4, dim] (int32) and sign [4, dim] (float32) of the copies (identity, left-right, front-back, diagonal =
    front-back of left-right) for a row made of `layout` blocks: "lin" | "ang" | "cmd" (3 columns) or "joint" (one
    column per joint)."""
    (lrp, lrs), (fbp, fbs) = joint_tables(joint_names, **kw)
    perms, signs = [[], []], [[], []]
    off = 0
    for blk in layout:
        for m, (jp, js) in enumerate(((lrp, lrs), (fbp, fbs))):
            if blk == "joint":
                perms[m] += list(off + jp)
                signs[m] += list(js)
            else:
                perms[m] += [off, off + 1, off + 2]
                signs[m] += list(_VEC[blk][m])
        off += len(joint_names) if blk == "joint" else 3
    ident = np.arange(off)
    lr_p, fb_p = np.array(perms[0]), np.array(perms[1])
    lr_s, fb_s = np.array(signs[0], dtype=np.float64), np.array(signs[1], dtype=np.float64)
    # diagonal: y = FB(LR(x)) -> y[c] = fb_s[c] * lr_s[fb_p[c]] * x[lr_p[fb_p[c]]]
    perm = np.stack([ident, lr_p, fb_p, lr_p[fb_p]]).astype(np.int32)
    sign = np.stack([np.ones(off), lr_s, fb_s, fb_s * lr_s[fb_p]]).astype(np.float32)
    return perm, sign


# the policy observation of the velocity tasks (anymal.py:110-125): angular velocity, projected gravity, command, q, qd, last action
POLICY_LAYOUT = ("ang", "lin", "cmd", "joint", "joint", "joint")
ACTION_LAYOUT = ("joint",)


class SymmetryAugmentation:
    """`aug(x)`: [n, dim] device tensor -> [n_sym * n, dim] (copy s in rows s * n .. (s + 1) * n)."""

    def __init__(self, perm, sign, device: str = "cuda:0", lib_path: str | None = None):
        import torch

        self._torch = torch
        self.lib = load_rollout_library(lib_path)
        self.lib.rl_symmetry_create.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_void_p)]
        self.lib.rl_symmetry_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        self.lib.rl_symmetry_destroy.argtypes = [C.c_void_p]
        self.device = torch.device(device)
        perm = np.ascontiguousarray(perm, dtype=np.int32)
        sign = np.ascontiguousarray(sign, dtype=np.float32)
        if perm.shape != sign.shape or perm.ndim != 2:
            raise ValueError("perm and sign must both be [n_sym, dim]")
        self.n_sym, self.dim = perm.shape
        self.handle = C.c_void_p()
        if self.lib.rl_symmetry_create(self.n_sym, self.dim, perm.ctypes.data_as(C.POINTER(C.c_int32)), sign.ctypes.data_as(C.POINTER(C.c_float)),
                                       self.device.index or 0, C.byref(self.handle)) != 0:
            raise RlRolloutError((self.lib.rl_rollout_last_error() or b"").decode())

    def __call__(self, x):
        torch = self._torch
        if x.dtype != torch.float32 or not x.is_contiguous() or x.ndim != 2 or x.shape[1] != self.dim or x.device != self.device:
            raise RlRolloutError(f"expected a contiguous fp32 [n, {self.dim}] tensor on {self.device}")
        out = torch.empty((self.n_sym * x.shape[0], self.dim), dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if self.lib.rl_symmetry_apply(self.handle, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.shape[0], stream) != 0:
            raise RlRolloutError((self.lib.rl_rollout_last_error() or b"").decode())
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.rl_symmetry_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_cache: dict = {}


def compute_symmetric_states(env, obs=None, actions=None):
    """Drop-in for the reference function (same signature and return value): `obs` is a dict / TensorDict with a
    "policy" entry (other groups are replicated unchanged, anymal.py:52), `actions` a [n, 12] tensor."""
    import torch

    def aug(x, layout):
        key = (layout, str(x.device))
        if key not in _cache:
            _cache[key] = SymmetryAugmentation(*layout_tables(layout), device=str(x.device))
        return _cache[key](x.contiguous().float())

    obs_aug = act_aug = None
    if obs is not None:
        obs_aug = type(obs)({k: (aug(v, POLICY_LAYOUT) if k == "policy" else v.repeat(4, *([1] * (v.ndim - 1)))) for k, v in obs.items()})
    if actions is not None:
        act_aug = aug(actions, ACTION_LAYOUT)
    return obs_aug, act_aug
