"""Fused actor / critic MLP inference on MI355X (`include/rl_policy.h`, `csrc/rl_policy.hip`).

What the reference obtains from rsl_rl as `policy = runner.get_inference_policy(device=...)` and calls as
`actions = policy(obs)` every step (`scripts/reinforcement_learning/rsl_rl/play.py:207,246`): the MLP
`RslRlPpoActorCriticCfg(actor_hidden_dims=[512, 256, 128], activation="elu")`
(`.../unitree_a1/agents/rsl_rl_ppo_cfg.py:15-22`), evaluated here by one hand-written HIP kernel (exact-fp32
MFMA) per call.  Weights come from an rsl_rl checkpoint `state_dict` (`actor.{0,2,4,6}.weight/bias`) or from any
`torch.nn.Sequential` of Linear + activation layers.  There is no CPU fallback: a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
POLICY_LIB = os.path.join(_HERE, "csrc", "librl_policy_hip.so")
POLICY_EXPORTS = ["rl_mlp_create", "rl_mlp_set_weights", "rl_mlp_forward", "rl_mlp_forward_small", "rl_mlp_forward_pair", "rl_mlp_forward_pair_act", "rl_mlp_in_dim", "rl_mlp_out_dim", "rl_mlp_destroy", "rl_mlp_last_error"]
ACTIVATIONS = {"elu": 0, "relu": 1, "tanh": 2}
_lib = None


class RlPolicyError(RuntimeError):
    pass


def load_policy_library(path: str | None = None) -> C.CDLL:
    global _lib
    path = path or os.environ.get("RL_POLICY_LIB") or POLICY_LIB  # RL_POLICY_LIB: alternative build (kernel ablations)
    if _lib is not None and path == POLICY_LIB:
        return _lib
    if not os.path.isfile(path):
        raise RlPolicyError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback)")
    lib = C.CDLL(path)
    fpp = C.POINTER(C.POINTER(C.c_float))
    lib.rl_mlp_create.argtypes = [C.POINTER(C.c_int32), C.c_int32, C.c_int32, fpp, fpp, C.c_int32, C.POINTER(C.c_void_p)]
    lib.rl_mlp_set_weights.argtypes = [C.c_void_p, fpp, fpp, C.c_void_p]
    lib.rl_mlp_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rl_mlp_forward_small.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rl_mlp_forward_pair.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.rl_mlp_forward_pair_act.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.rl_mlp_in_dim.argtypes = [C.c_void_p]
    lib.rl_mlp_out_dim.argtypes = [C.c_void_p]
    lib.rl_mlp_destroy.argtypes = [C.c_void_p]
    lib.rl_mlp_last_error.restype = C.c_char_p
    if path == POLICY_LIB:
        _lib = lib
    return lib


class MlpPolicy:
    """y = MLP(x) on the GPU.  `weights[l]`: [out, in] (nn.Linear layout), `biases[l]`: [out]."""

    def __init__(self, weights, biases, activation: str = "elu", device: str = "cuda:0", lib_path: str | None = None):
        import torch

        self._torch = torch
        self.lib = load_policy_library(lib_path)
        self.device = torch.device(device)
        ws = [np.ascontiguousarray(np.asarray(w, dtype=np.float32)) for w in weights]
        bs = [np.ascontiguousarray(np.asarray(b, dtype=np.float32)) for b in biases]
        n = len(ws)
        dims = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        for l in range(n):
            if ws[l].shape != (dims[l + 1], dims[l]) or bs[l].shape != (dims[l + 1],):
                raise ValueError(f"layer {l}: weight {ws[l].shape} / bias {bs[l].shape} do not chain")
        fp = C.POINTER(C.c_float)
        wp = (fp * n)(*[w.ctypes.data_as(fp) for w in ws])
        bp = (fp * n)(*[b.ctypes.data_as(fp) for b in bs])
        self.handle = C.c_void_p()
        rc = self.lib.rl_mlp_create((C.c_int32 * (n + 1))(*dims), n, ACTIVATIONS[activation], wp, bp, self.device.index or 0, C.byref(self.handle))
        if rc != 0:
            raise RlPolicyError((self.lib.rl_mlp_last_error() or b"").decode())
        self.in_dim, self.out_dim = dims[0], dims[-1]
        self._shapes = [w.shape for w in ws]
        self._out = None

    @classmethod
    def from_state_dict(cls, sd, prefix: str = "actor", activation: str = "elu", **kw):
        """rsl_rl `ActorCritic.state_dict()` layout: `<prefix>.<2 l>.weight`, `<prefix>.<2 l>.bias`."""
        idx = sorted({int(k.split(".")[1]) for k in sd if k.startswith(prefix + ".") and k.endswith(".weight")})
        to_np = lambda t: t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)  # noqa: E731
        return cls([to_np(sd[f"{prefix}.{i}.weight"]) for i in idx], [to_np(sd[f"{prefix}.{i}.bias"]) for i in idx], activation, **kw)

    def set_weights(self, weights, biases):
        """New parameters in place (`rl_mlp_set_weights`): after an optimiser step of a training loop.  Same shapes as at creation;
        the device images keep their addresses (captured graphs that launch this network stay valid)."""
        to_np = lambda t: np.ascontiguousarray((t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)).astype(np.float32, copy=False))  # noqa: E731
        ws, bs = [to_np(w) for w in weights], [to_np(b) for b in biases]
        if [w.shape for w in ws] != self._shapes or [b.shape for b in bs] != [(s[0],) for s in self._shapes]:
            raise ValueError("set_weights: layer shapes differ from the network's")
        fp = C.POINTER(C.c_float)
        n = len(ws)
        wp = (fp * n)(*[w.ctypes.data_as(fp) for w in ws])
        bp = (fp * n)(*[b.ctypes.data_as(fp) for b in bs])
        stream = self._torch.cuda.current_stream(self.device).cuda_stream
        if self.lib.rl_mlp_set_weights(self.handle, wp, bp, C.c_void_p(stream)) != 0:
            raise RlPolicyError((self.lib.rl_mlp_last_error() or b"").decode())

    def load_linear_layers(self, module):
        """`set_weights` from the nn.Linear layers of a torch module (an `nn.Sequential` of Linear / activation pairs, rsl_rl's actor / critic)."""
        lin = [m for m in module.modules() if isinstance(m, self._torch.nn.Linear)]
        self.set_weights([m.weight for m in lin], [m.bias for m in lin])

    def _prepare(self, obs):
        torch = self._torch
        if obs.device != self.device or obs.dtype != torch.float32 or not obs.is_contiguous():
            obs = obs.to(device=self.device, dtype=torch.float32).contiguous()
        if obs.ndim != 2 or obs.shape[1] != self.in_dim:
            raise ValueError(f"obs shape {tuple(obs.shape)} != (N, {self.in_dim})")
        n = obs.shape[0]
        if self._out is None or self._out.shape[0] != n:
            self._out = torch.empty(n, self.out_dim, device=self.device, dtype=torch.float32)
        return obs, n

    def forward_pair(self, obs, other: "MlpPolicy", other_obs):
        """(self(obs), other(other_obs)) from ONE kernel launch - the actor and the critic of a rollout step
        (rsl_rl PPO.act: `policy.act(obs)` then `policy.evaluate(privileged_obs)`); both batches have the same row count."""
        obs, n = self._prepare(obs)
        other_obs, n2 = other._prepare(other_obs)
        if n != n2 or other.device != self.device:
            raise ValueError("forward_pair needs two batches of the same row count on the same device")
        stream = self._torch.cuda.current_stream(self.device).cuda_stream
        if self.lib.rl_mlp_forward_pair(self.handle, C.c_void_p(obs.data_ptr()), C.c_void_p(self._out.data_ptr()), other.handle,
                                        C.c_void_p(other_obs.data_ptr()), C.c_void_p(other._out.data_ptr()), n, C.c_void_p(stream)) != 0:
            raise RlPolicyError((self.lib.rl_mlp_last_error() or b"").decode())
        return self._out, other._out

    def forward_pair_act(self, obs_ptr, other: "MlpPolicy", other_obs_ptr, ep) -> int:
        """The actor (self) / critic (other) launch with the rollout step's stochastic head in its epilogue (include/rl_policy.h
        rl_mlp_forward_pair_act; `ep`: a filled `rollout.ActEpilogue`, whose `s_values` receives V).  0: done; 1: not available for this size."""
        n = int(ep.num_envs)
        if self._out is None or self._out.shape[0] != n:
            self._out = self._torch.empty(n, self.out_dim, device=self.device, dtype=self._torch.float32)
        stream = self._torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.rl_mlp_forward_pair_act(self.handle, obs_ptr, C.c_void_p(self._out.data_ptr()), other.handle, other_obs_ptr, C.c_void_p(ep.s_values), n,
                                              C.byref(ep), C.c_void_p(stream))
        if rc < 0:
            raise RlPolicyError((self.lib.rl_mlp_last_error() or b"").decode())
        return rc

    def __call__(self, obs):
        """obs: float32 device tensor [N, in_dim] (or a dict / TensorDict with a "policy" entry, as rsl_rl passes)."""
        torch = self._torch
        if not torch.is_tensor(obs):
            obs = obs["policy"]
        obs, n = self._prepare(obs)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if self.lib.rl_mlp_forward(self.handle, C.c_void_p(obs.data_ptr()), C.c_void_p(self._out.data_ptr()), n, C.c_void_p(stream)) != 0:
            raise RlPolicyError((self.lib.rl_mlp_last_error() or b"").decode())
        return self._out

    def forward_into(self, obs, out_ptr: int, small: bool = False):
        """self(obs) written to the device address `out_ptr` ([N, out_dim] fp32, e.g. the values slot of a rollout storage).
        `small`: the small-footprint launch (`rl_mlp_forward_small`), which fits on a CU beside the env-step kernel's workgroup."""
        obs, n = self._prepare(obs)
        stream = self._torch.cuda.current_stream(self.device).cuda_stream
        if (self.lib.rl_mlp_forward_small if small else self.lib.rl_mlp_forward)(self.handle, C.c_void_p(obs.data_ptr()), C.c_void_p(out_ptr), n, C.c_void_p(stream)) != 0:
            raise RlPolicyError((self.lib.rl_mlp_last_error() or b"").decode())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.rl_mlp_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
