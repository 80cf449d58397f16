"""ctypes binding of the C-ABI in ``include/rl_env.h``.

The product library is ``robot_lab_amd/csrc/librl_env_hip.so`` (hand-written HIP for gfx950, built by
``__graft_entry__.build()``).  There is no CPU fallback: if the library is missing or cannot be
loaded, :func:`load_library` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .desc import BUF, REAL_C, REAL_NP, EnvDesc, RL_LOG_SIZE

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(_HERE, "csrc", "librl_env_hip.so")

EXPORTS = [
    "rl_env_create", "rl_env_reset", "rl_env_step", "rl_env_step_record", "rl_env_get_buffer", "rl_env_export_state", "rl_env_commit_state",
    "rl_env_import_state", "rl_env_read_log", "rl_env_log_slot", "rl_env_obs_slot", "rl_env_step_count", "rl_env_set_step_count", "rl_env_num_envs", "rl_env_num_actions", "rl_env_obs_dim", "rl_env_max_episode_length", "rl_env_envs_per_wavefront", "rl_env_spec_id", "rl_env_plan",
    "rl_env_spec_source", "rl_env_register_spec_plugin", "rl_env_spec_plugin_count", "rl_env_abi_stamp",
    "rl_env_destroy", "rl_env_last_error", "rl_env_desc_size", "rl_env_graph_begin", "rl_env_graph_end", "rl_env_graph_launching",
]

_libs: dict[str, C.CDLL] = {}


class RlEnvError(RuntimeError):
    pass


def load_library(path: str | None = None) -> C.CDLL:
    path = path or os.environ.get("RL_ENV_LIB") or HIP_LIB  # RL_ENV_LIB: alternative HIP build (kernel ablations)
    if path in _libs:
        return _libs[path]
    if not os.path.isfile(path):
        raise RlEnvError(
            f"{path} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "robot_lab_amd has no CPU fallback.")
    lib = C.CDLL(path)
    fp = C.POINTER(REAL_C)
    lib.rl_env_create.argtypes = [C.POINTER(EnvDesc), fp, fp, fp, C.c_int32, C.c_uint64, C.c_int32, C.POINTER(C.c_void_p)]
    lib.rl_env_reset.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_void_p]
    lib.rl_env_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rl_env_step_record.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, REAL_C, C.c_void_p]
    lib.rl_env_get_buffer.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.rl_env_export_state.argtypes = [C.c_void_p, C.c_void_p]
    lib.rl_env_commit_state.argtypes = [C.c_void_p, C.c_void_p]
    lib.rl_env_obs_slot.argtypes = [C.c_void_p]
    lib.rl_env_obs_slot.restype = C.c_int32
    lib.rl_env_step_count.argtypes = [C.c_void_p]
    lib.rl_env_step_count.restype = C.c_int64
    lib.rl_env_set_step_count.argtypes = [C.c_void_p, C.c_int64]
    lib.rl_env_import_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    for n in ("rl_env_graph_begin", "rl_env_graph_end", "rl_env_graph_launching"):
        getattr(lib, n).argtypes = [C.c_void_p, C.c_void_p]
    lib.rl_env_read_log.argtypes = [C.c_void_p, fp, C.c_void_p]
    lib.rl_env_log_slot.argtypes = [C.c_void_p]
    lib.rl_env_log_slot.restype = C.c_int32
    for n in ("rl_env_num_envs", "rl_env_num_actions", "rl_env_max_episode_length", "rl_env_envs_per_wavefront", "rl_env_spec_id"):
        getattr(lib, n).argtypes = [C.c_void_p]
        getattr(lib, n).restype = C.c_int32
    lib.rl_env_plan.argtypes = [C.POINTER(EnvDesc), C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.rl_env_spec_source.argtypes = [C.POINTER(EnvDesc), C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    lib.rl_env_spec_source.restype = C.c_int
    lib.rl_env_register_spec_plugin.argtypes = [C.c_char_p]
    lib.rl_env_spec_plugin_count.restype = C.c_int32
    lib.rl_env_abi_stamp.restype = C.c_char_p
    lib.rl_env_obs_dim.argtypes = [C.c_void_p, C.c_int32]
    lib.rl_env_obs_dim.restype = C.c_int32
    lib.rl_env_destroy.argtypes = [C.c_void_p]
    lib.rl_env_last_error.restype = C.c_char_p
    lib.rl_env_desc_size.restype = C.c_uint64
    if lib.rl_env_desc_size() != C.sizeof(EnvDesc):
        raise RlEnvError(f"ABI mismatch: library rl_env_desc is {lib.rl_env_desc_size()} bytes, Python mirror {C.sizeof(EnvDesc)}")
    _libs[path] = lib
    return lib


def _fptr(a):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=REAL_NP)
    return a, a.ctypes.data_as(C.POINTER(REAL_C))


def plan(desc: EnvDesc, num_envs: int, n_cu: int = 256, lib_path: str | None = None) -> dict:
    """`rl_env_plan`: the launch geometry `rl_env_create` would pick (no device needed)."""
    lib = load_library(lib_path)
    out = (C.c_int32 * 4)()
    if lib.rl_env_plan(C.byref(desc), num_envs, n_cu, out) != 0:
        raise RlEnvError((lib.rl_env_last_error() or b"").decode())
    return dict(lanes_per_limb=out[0], wavefronts_per_workgroup=out[1], lds_bytes_per_wavefront=out[2], lds_bytes_per_workgroup=out[3])


class NativeEnv:
    """Thin owner of an ``rl_env*``.  Pointers in/out are raw integers (device addresses for the HIP
    library, host addresses for the CPU lane emulator used by the tests)."""

    def __init__(self, desc: EnvDesc, heights, terrain_origins, env_origins, num_envs: int, seed: int, device: int = 0,
                 lib_path: str | None = None):
        self.lib = load_library(lib_path)
        self.handle = C.c_void_p()
        keep = []
        ptrs = []
        for a in (heights, terrain_origins, env_origins):
            r = _fptr(a)
            keep.append(r)
            ptrs.append(None if r is None else r[1])
        rc = self.lib.rl_env_create(C.byref(desc), ptrs[0], ptrs[1], ptrs[2], num_envs, C.c_uint64(seed & (2**64 - 1)), device, C.byref(self.handle))
        if rc != 0:
            raise RlEnvError(self.error())
        self.num_envs = num_envs
        self.num_actions = self.lib.rl_env_num_actions(self.handle)
        self.max_episode_length = self.lib.rl_env_max_episode_length(self.handle)

    def envs_per_wavefront(self) -> int:
        return int(self.lib.rl_env_envs_per_wavefront(self.handle))

    def spec_id(self) -> int:
        """0: the term-stack interpreter; > 0: the step kernel specialised on this task (csrc/env_spec.h)."""
        return int(self.lib.rl_env_spec_id(self.handle))

    def error(self) -> str:
        return (self.lib.rl_env_last_error() or b"").decode()

    def _check(self, rc):
        if rc != 0:
            raise RlEnvError(self.error())

    def reset(self, env_ids=None, stream: int = 0):
        if env_ids is None:
            self._check(self.lib.rl_env_reset(self.handle, None, 0, C.c_void_p(stream)))
        else:
            ids = np.ascontiguousarray(env_ids, dtype=np.int32)
            self._check(self.lib.rl_env_reset(self.handle, ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids), C.c_void_p(stream)))

    def step(self, action_ptr: int, stream: int = 0):
        self._check(self.lib.rl_env_step(self.handle, C.c_void_p(action_ptr), C.c_void_p(stream)))

    def step_record(self, action_ptr: int, values_ptr: int, rewards_ptr: int, dones_ptr: int, gamma: float, stream: int = 0):
        self._check(self.lib.rl_env_step_record(self.handle, C.c_void_p(action_ptr), C.c_void_p(values_ptr), C.c_void_p(rewards_ptr),
                                                C.c_void_p(dones_ptr), gamma, C.c_void_p(stream)))

    def buffer(self, name: str):
        """-> (address, shape tuple, numpy dtype)"""
        ptr = C.c_void_p()
        shape = (C.c_int64 * 3)()
        nd, es = C.c_int32(), C.c_int32()
        self._check(self.lib.rl_env_get_buffer(self.handle, BUF[name], C.byref(ptr), shape, C.byref(nd), C.byref(es)))
        shp = tuple(int(shape[i]) for i in range(nd.value))
        dt = {"TERMINATED": np.uint8, "TIME_OUT": np.uint8, "EPISODE_LENGTH": np.int64, "TERRAIN_LEVEL": np.int32}.get(name, REAL_NP)
        if np.dtype(dt).itemsize != es.value:
            raise RlEnvError(f"buffer {name}: the library reports {es.value}-byte elements, the binding expects {np.dtype(dt)}")
        return ptr.value, shp, dt

    def export_state(self, stream: int = 0):
        self._check(self.lib.rl_env_export_state(self.handle, C.c_void_p(stream)))

    def commit_state(self, stream: int = 0):
        """AoS state buffers (ROOT_STATE, JOINT_*, ACTION, GAINS, CONTACT_TIMERS, TASK_STATE, ENV_ORIGIN) -> simulator state."""
        self._check(self.lib.rl_env_commit_state(self.handle, C.c_void_p(stream)))

    def obs_slot(self) -> int:
        return int(self.lib.rl_env_obs_slot(self.handle))

    @property
    def step_count(self) -> int:
        return int(self.lib.rl_env_step_count(self.handle))

    @step_count.setter
    def step_count(self, n: int):
        self._check(self.lib.rl_env_set_step_count(self.handle, int(n)))

    # hipGraph capture of a loop around step() (include/rl_env.h)
    def graph_begin(self, stream: int = 0):
        self._check(self.lib.rl_env_graph_begin(self.handle, C.c_void_p(stream)))

    def graph_end(self, stream: int = 0) -> int:
        n = self.lib.rl_env_graph_end(self.handle, C.c_void_p(stream))
        if n < 0:
            raise RlEnvError(self.error())
        return n

    def graph_launching(self, stream: int = 0):
        self._check(self.lib.rl_env_graph_launching(self.handle, C.c_void_p(stream)))

    def import_state(self, root_ptr=0, qpos_ptr=0, qvel_ptr=0, stream: int = 0):
        self._check(self.lib.rl_env_import_state(self.handle, C.c_void_p(root_ptr), C.c_void_p(qpos_ptr), C.c_void_p(qvel_ptr), C.c_void_p(stream)))

    def read_log(self, stream: int = 0) -> np.ndarray:
        out = np.zeros(RL_LOG_SIZE, dtype=REAL_NP)
        self._check(self.lib.rl_env_read_log(self.handle, out.ctypes.data_as(C.POINTER(REAL_C)), C.c_void_p(stream)))
        return out

    def log_slot(self) -> int:
        return int(self.lib.rl_env_log_slot(self.handle))

    def close(self):
        if self.handle:
            self.lib.rl_env_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
