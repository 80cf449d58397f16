"""On-policy rollout storage on MI355X (`include/rl_rollout.h`, `csrc/rl_rollout.hip`).

The host-side mirror of what the reference drives through `runner.learn(...)`
(`scripts/reinforcement_learning/rsl_rl/train.py:224`): rsl_rl's `PPO.act` / `PPO.process_env_step` /
`PPO.compute_returns` and the `RolloutStorage` they fill (same attribute names: `observations`,
`privileged_observations`, `actions`, `mu`, `sigma`, `actions_log_prob`, `values`, `rewards`, `dones`, `returns`,
`advantages`, each `[num_transitions_per_env, num_envs, ...]`).  Every call is one or three hand-written HIP kernel
launches on the current torch stream; the storage lives in HBM and torch adopts it without copies.  No CPU
fallback: a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROLLOUT_LIB = os.path.join(_HERE, "csrc", "librl_rollout_hip.so")
ROLLOUT_EXPORTS = ["rl_rollout_create", "rl_rollout_act", "rl_rollout_act_epilogue", "rl_rollout_act_done", "rl_rollout_record", "rl_rollout_record_slots", "rl_rollout_values_slot", "rl_rollout_store_critic_obs", "rl_rollout_compute_returns", "rl_rollout_clear",
                   "rl_rollout_get_buffer", "rl_rollout_step", "rl_rollout_destroy", "rl_rollout_last_error",
                   "rl_rollout_graph_begin", "rl_rollout_graph_end", "rl_rollout_graph_launching"]
# name -> (rl_rollout_buffer id, dtype, has a trailing feature dim)
BUFFERS = dict(observations=(0, np.float32, True), privileged_observations=(1, np.float32, True), actions=(2, np.float32, True),
               mu=(3, np.float32, True), sigma=(4, np.float32, True), actions_log_prob=(5, np.float32, False),
               values=(6, np.float32, False), rewards=(7, np.float32, False), dones=(8, np.uint8, False),
               returns=(9, np.float32, False), advantages=(10, np.float32, False))
_lib = None


class ActEpilogue(C.Structure):
    """include/rl_act.h `rl_act_epilogue`: the step's sampling / log-prob / slot addresses for the actor launch's epilogue."""
    _fields_ = [("actions_out", C.c_void_p), ("s_obs", C.c_void_p), ("s_critic_obs", C.c_void_p), ("s_actions", C.c_void_p), ("s_mu", C.c_void_p),
                ("s_sigma", C.c_void_p), ("s_logp", C.c_void_p), ("s_values", C.c_void_p), ("std", C.c_void_p), ("counter_base", C.c_void_p),
                ("seed", C.c_uint64), ("counter", C.c_uint32), ("num_envs", C.c_int32), ("obs_dim", C.c_int32), ("critic_dim", C.c_int32),
                ("act_dim", C.c_int32), ("clip", C.c_float)]


class RlRolloutError(RuntimeError):
    pass


def load_rollout_library(path: str | None = None) -> C.CDLL:
    global _lib
    path = path or os.environ.get("RL_ROLLOUT_LIB") or ROLLOUT_LIB  # RL_ROLLOUT_LIB: alternative build (kernel analysis)
    if _lib is not None and path == ROLLOUT_LIB:
        return _lib
    if not os.path.isfile(path):
        raise RlRolloutError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback)")
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.rl_rollout_create.argtypes = [C.c_int32] * 5 + [C.c_uint64, C.c_int32, C.POINTER(vp)]
    lib.rl_rollout_act.argtypes = [vp] * 8
    lib.rl_rollout_act_epilogue.argtypes = [vp, vp, vp, C.c_float, C.POINTER(ActEpilogue)]
    lib.rl_rollout_act_done.argtypes = [vp]
    lib.rl_rollout_record.argtypes = [vp, vp, vp, vp, C.c_float, vp]
    lib.rl_rollout_record_slots.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.rl_rollout_values_slot.argtypes = [vp, C.POINTER(vp)]
    lib.rl_rollout_store_critic_obs.argtypes = [vp, vp, vp]
    lib.rl_rollout_compute_returns.argtypes = [vp, vp, C.c_float, C.c_float, C.c_int32, vp]
    lib.rl_rollout_clear.argtypes = [vp]
    for n in ("rl_rollout_graph_begin", "rl_rollout_graph_end", "rl_rollout_graph_launching"):
        getattr(lib, n).argtypes = [vp, vp]
    lib.rl_rollout_get_buffer.argtypes = [vp, C.c_int32, C.POINTER(vp), C.POINTER(C.c_int64)]
    lib.rl_rollout_step.argtypes = [vp]
    lib.rl_rollout_step.restype = C.c_int32
    lib.rl_rollout_destroy.argtypes = [vp]
    lib.rl_rollout_last_error.restype = C.c_char_p
    if path == ROLLOUT_LIB:
        _lib = lib
    return lib


class _DevView:
    def __init__(self, ptr, shape, dtype, owner):
        self._owner = owner
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=np.dtype(dtype).str, data=(int(ptr), False), version=2, strides=None)


class RolloutStorage:
    """`RolloutStorage(num_envs, num_transitions_per_env, obs_dim, privileged_obs_dim, act_dim)` plus the three PPO
    calls that fill it.  Tensors passed in must be contiguous fp32 (uint8 / bool for the done flags) on `device`."""

    def __init__(self, num_envs: int, num_transitions_per_env: int, obs_dim: int, privileged_obs_dim: int, act_dim: int,
                 seed: int = 0, device: str = "cuda:0", lib_path: str | None = None):
        import torch

        self._torch = torch
        self.lib = load_rollout_library(lib_path)
        self.device = torch.device(device)
        self.num_envs, self.num_transitions_per_env = num_envs, num_transitions_per_env
        self.handle = C.c_void_p()
        dev = self.device.index or 0
        if self.lib.rl_rollout_create(num_envs, num_transitions_per_env, obs_dim, privileged_obs_dim, act_dim, C.c_uint64(seed & (2**64 - 1)), dev,
                                      C.byref(self.handle)) != 0:
            raise RlRolloutError(self._err())
        dims = dict(observations=obs_dim, privileged_observations=privileged_obs_dim, actions=act_dim, mu=act_dim, sigma=act_dim)
        for name, (which, dt, feat) in BUFFERS.items():
            ptr, cnt = C.c_void_p(), C.c_int64()
            if self.lib.rl_rollout_get_buffer(self.handle, which, C.byref(ptr), C.byref(cnt)) != 0:
                raise RlRolloutError(self._err())
            shape = (num_transitions_per_env, num_envs) + ((dims[name],) if feat else ())
            t = torch.as_tensor(_DevView(ptr.value, shape, dt, self), device=self.device)
            setattr(self, name, t.view(torch.bool) if name == "dones" else t)
        self._actions = torch.empty((num_envs, act_dim), dtype=torch.float32, device=self.device)

    def _err(self):
        return (self.lib.rl_rollout_last_error() or b"").decode()

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def _f32(self, t, shape):
        if t.dtype != self._torch.float32 or not t.is_contiguous() or tuple(t.shape) != tuple(shape) or t.device != self.device:
            raise RlRolloutError(f"expected a contiguous fp32 tensor of shape {tuple(shape)} on {self.device}, got {t.dtype} {tuple(t.shape)} on {t.device}")
        return self._p(t)

    @property
    def step(self) -> int:
        return int(self.lib.rl_rollout_step(self.handle))

    # -- PPO.act: actions = mean + std * eps; log-prob; first half of add_transitions
    def act(self, obs, privileged_obs, action_mean, action_std, values):
        """`privileged_obs` / `values` None: the critic's half of the slot is written on the critic's own stream
        (`critic_half`; include/rl_rollout.h)."""
        N, A = self.num_envs, self.actions.shape[-1]
        args = [self._f32(obs, (N, self.observations.shape[-1])),
                None if privileged_obs is None else self._f32(privileged_obs, (N, self.privileged_observations.shape[-1])),
                self._f32(action_mean, (N, A)), self._f32(action_std, (A,)), None if values is None else self._f32(values.view(-1), (N,)), self._p(self._actions)]
        if self.lib.rl_rollout_act(self.handle, *args, self._stream()) != 0:
            raise RlRolloutError(self._err())
        return self._actions

    def act_fused(self, actor, critic, obs, privileged_obs, action_std, clip_actions=None):
        """`act` inside the actor / critic launch (include/rl_act.h): sampling, log-prob and the slot's first half in the epilogue of
        `rl_mlp_forward_pair_act`, V straight into the values slot - no launch between the networks and env.step.  Falls back to the
        pair launch + `act` when the networks' kernel for this size has no epilogue.  Returns the actions env.step consumes (clamped to
        +-clip_actions when given; the storage keeps the sample).  Same numbers as `act`, bit for bit."""
        N, A = self.num_envs, self.actions.shape[-1]
        ep = ActEpilogue()
        if self.lib.rl_rollout_act_epilogue(self.handle, self._f32(action_std, (A,)), self._p(self._actions), -1.0 if clip_actions is None else float(clip_actions),
                                            C.byref(ep)) != 0:
            raise RlRolloutError(self._err())
        rc = actor.forward_pair_act(self._f32(obs, (N, self.observations.shape[-1])), critic, self._f32(privileged_obs, (N, self.privileged_observations.shape[-1])), ep)
        if rc == 1:  # this size / precision runs a kernel without the epilogue
            mean, values = actor.forward_pair(obs, critic, privileged_obs)
            actions = self.act(obs, privileged_obs, mean, action_std, values)
            return actions if clip_actions is None else actions.clamp(-clip_actions, clip_actions)
        if self.lib.rl_rollout_act_done(self.handle) != 0:
            raise RlRolloutError(self._err())
        return self._actions

    def critic_half(self, critic, privileged_obs, small: bool = False):
        """The critic's half of the current step on the CURRENT torch stream (robot_lab_amd/collect.py calls it under a side stream):
        the privileged observations go into the slot, `critic` (an MlpPolicy with one output) writes V straight into the values slot."""
        N = self.num_envs
        v = C.c_void_p()
        if self.lib.rl_rollout_values_slot(self.handle, C.byref(v)) != 0:
            raise RlRolloutError(self._err())
        if self.lib.rl_rollout_store_critic_obs(self.handle, self._f32(privileged_obs, (N, self.privileged_observations.shape[-1])), self._stream()) != 0:
            raise RlRolloutError(self._err())
        critic.forward_into(privileged_obs, v.value, small=small)

    # -- PPO.process_env_step: time-out bootstrapping, dones; second half of add_transitions
    def process_env_step(self, rewards, terminated, time_outs, gamma: float):
        N = self.num_envs
        for f in (terminated, time_outs):
            if f.dtype not in (self._torch.uint8, self._torch.bool) or not f.is_contiguous() or f.numel() != N:
                raise RlRolloutError("terminated / time_outs must be contiguous uint8 or bool tensors of num_envs entries")
        if self.lib.rl_rollout_record(self.handle, self._f32(rewards.view(-1), (N,)), self._p(terminated), self._p(time_outs), gamma, self._stream()) != 0:
            raise RlRolloutError(self._err())

    def record_slots(self):
        """(values, rewards, dones) device addresses of the current step's slots, closing the step: for a producer that writes
        the transition's second half itself (`ManagerBasedRLEnv.step(actions, rollout=storage)` -> `rl_env_step_record`)."""
        v, r, d = C.c_void_p(), C.c_void_p(), C.c_void_p()
        if self.lib.rl_rollout_record_slots(self.handle, C.byref(v), C.byref(r), C.byref(d)) != 0:
            raise RlRolloutError(self._err())
        return v.value, r.value, d.value

    # -- PPO.compute_returns / RolloutStorage.compute_returns
    def compute_returns(self, last_values, gamma: float, lam: float, normalize_advantage: bool = True):
        if self.lib.rl_rollout_compute_returns(self.handle, self._f32(last_values.view(-1), (self.num_envs,)), gamma, lam, int(normalize_advantage),
                                               self._stream()) != 0:
            raise RlRolloutError(self._err())

    def clear(self):
        if self.lib.rl_rollout_clear(self.handle) != 0:
            raise RlRolloutError(self._err())

    # -- hipGraph capture of a collection iteration (include/rl_rollout.h; robot_lab_amd/collect.py drives it)
    def graph_begin(self):
        if self.lib.rl_rollout_graph_begin(self.handle, self._stream()) != 0:
            raise RlRolloutError(self._err())

    def graph_end(self) -> int:
        n = self.lib.rl_rollout_graph_end(self.handle, self._stream())
        if n < 0:
            raise RlRolloutError(self._err())
        return n

    def graph_launching(self):
        if self.lib.rl_rollout_graph_launching(self.handle, self._stream()) != 0:
            raise RlRolloutError(self._err())

    def close(self):
        if self.handle:
            for name in BUFFERS:
                if hasattr(self, name):
                    delattr(self, name)
            self.lib.rl_rollout_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
