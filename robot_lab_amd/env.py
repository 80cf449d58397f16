"""`ManagerBasedRLEnv` - the drop-in boundary.

Same construction and call surface as the class the reference registers as its gym entry point
(`isaaclab.envs:ManagerBasedRLEnv`; `VEL/config/quadruped/unitree_a1/__init__.py:12-32`,
`scripts/reinforcement_learning/rsl_rl/train.py:177`, `scripts/tools/zero_agent.py:56-73`):

    env = gym.make("RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0", cfg=env_cfg)
    obs, extras = env.reset()
    obs, rew, terminated, time_outs, extras = env.step(actions)      # device tensors in, device tensors out

but `step()` is ONE hand-written HIP kernel launch on the MI355X (through the C-ABI of
`include/rl_env.h`) instead of PhysX + a few hundred torch kernels.  PyTorch is only the owner of
device memory handed in (actions) and the view type handed out; every returned tensor is a zero-copy
view of an env-owned HBM buffer.  Rewards, dones and the inspection views are rewritten in place by the
next `step()` (the reference's `reward_buf` / `reset_buf` are too).  The observation groups are NOT: the
reference builds fresh observation tensors on every step (`ObservationManager.compute` -> `torch.cat`) and
rsl_rl's `PPO.act` keeps a reference to them across the following `env.step()`, so the kernel alternates
between two HBM buffers per group - the tensors returned by step t stay intact until step t + 2 is launched
(no copy, no extra traffic; `include/rl_env.h` "Ownership").

There is no CPU path: without `librl_env_hip.so` or without a GPU the constructor raises.
"""
from __future__ import annotations

import collections
import os
import weakref

import math

import numpy as np
import torch

from .capi import NativeEnv, RlEnvError
from .desc import RL_LOG_RING, RL_LOG_SIZE, EnvDesc
from .scene import build_world, load_bundle

try:  # gymnasium (or the shim in robot_lab_amd/shims) supplies Env / spaces when available
    import gymnasium as gym
    _EnvBase = gym.Env
except Exception:  # noqa: BLE001
    gym = None
    _EnvBase = object


class _DevView:
    """`__cuda_array_interface__` holder: lets torch adopt an env-owned device buffer without copying."""

    def __init__(self, ptr, shape, dtype, owner):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=np.dtype(dtype).str, data=(int(ptr), False), version=2, strides=None)
        self._owner = owner


_INSPECTION = ("CONTACT_FORCE", "JOINT_TORQUE", "JOINT_ACC")


class _Buffers(dict):
    """name -> torch view of an env-owned device buffer, adopted on first use.  The three inspection views
    (`data.applied_torque`, `data.joint_acc`, contact forces) are only allocated - and from then on written
    by every step - when somebody asks for them (`include/rl_env.h: rl_env_get_buffer`)."""

    def __init__(self, env):
        super().__init__()
        self._env = env

    def __missing__(self, name):
        e = self._env
        ptr, shape, dt = e._native.buffer(name)
        t = torch.as_tensor(_DevView(ptr, shape, dt, e._native), device=e.device)
        self[name] = t
        return t


class _LazyLog(dict):
    """`extras["log"]`: the episode log of the most recent step that reset an environment [UPSTREAM B1/B2] - the
    reference rebuilds `extras["log"]` inside `_reset_idx` only, so on a step without resets a caller still sees the
    previous one (rsl_rl appends it once per step and averages).  Backed by the device-side ring (the kernel resolves
    an empty slot to its predecessor, `csrc/env_terms.h step()`); entries are 0-dim device tensors materialised on
    first access, so a training loop that only reads them at log time never forces a host sync inside `step()`."""

    def __init__(self, env, slot, prev, step):
        super().__init__()
        self._env, self._slot, self._prev, self._step, self._done = env, slot, prev, step, False

    def _fill(self):
        if self._done:
            return
        e = self._env
        # the kernel logs step k into ring slot k % RL_LOG_RING and clears the slot of step k + 1 (include/rl_env.h):
        # this step's slot is untouched until RL_LOG_RING - 2 further steps have been launched, its PREDECESSOR's slot (read when
        # this step reset nobody) one step less: step k + RL_LOG_RING - 2 clears slot k - 1.  The env materialises a log that is
        # still referenced before that happens (ManagerBasedRLEnv._retire_logs), so this only fires for a log detached from its env
        if e.common_step_counter - self._step > RL_LOG_RING - 3:
            raise RuntimeError(f'extras["log"] of step {self._step} was read {e.common_step_counter - self._step} steps later: '
                               f"the device keeps the last {RL_LOG_RING - 3} steps")
        self._done = True
        # this step's slot if it reset an env, else its predecessor (which the kernel has already resolved the same way)
        # (a slot is RL_LOG_PARTS partial rows: include/rl_env.h RL_BUF_LOG)
        cur, prev = self._slot.sum(0), self._prev.sum(0)
        s = torch.where(cur[0] > 0, cur, prev)
        self._slot = self._prev = None
        cnt = torch.clamp(s[0], min=1.0)
        for i, name in enumerate(e.desc.reward_names):
            dict.__setitem__(self, "Episode_Reward/" + name, s[8 + i] / cnt / e.max_episode_length_s)
        dict.__setitem__(self, "Metrics/base_velocity/error_vel_xy", s[4] / cnt)
        dict.__setitem__(self, "Metrics/base_velocity/error_vel_yaw", s[5] / cnt)
        dict.__setitem__(self, "Episode_Termination/time_out", s[1])
        dict.__setitem__(self, "Episode_Termination/terrain_out_of_bounds", s[2])
        if e.desc.task.term_illegal_contact:
            dict.__setitem__(self, "Episode_Termination/illegal_contact", s[3])
        if e.desc.terrain.curriculum and not e.desc.terrain.is_plane:
            dict.__setitem__(self, "Curriculum/terrain_levels", e.terrain_levels.float().mean())
        if e.desc.task.cur_cmd_lin:  # what the term functions return (curriculums.py:62, 96): the live upper bounds
            dict.__setitem__(self, "Curriculum/command_levels_lin_vel", e.command_levels[1])
        if e.desc.task.cur_cmd_ang:
            dict.__setitem__(self, "Curriculum/command_levels_ang_vel", e.command_levels[5])

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def __contains__(self, k):
        self._fill()
        return dict.__contains__(self, k)


class _ArticulationData:
    """`env.scene["robot"].data` view (camera follow `rl_utils.py:12-13`, user scripts): refreshed lazily
    from the SoA simulator state by `rl_env_export_state`."""

    def __init__(self, env):
        self._env = env

    def _root(self):
        self._env._export()
        return self._env._bufs["ROOT_STATE"]

    root_pos_w = property(lambda self: self._root()[:, 0:3])
    root_quat_w = property(lambda self: self._root()[:, 3:7])
    root_lin_vel_w = property(lambda self: self._root()[:, 7:10])
    root_ang_vel_w = property(lambda self: self._root()[:, 10:13])
    root_state_w = property(lambda self: self._root())

    @property
    def joint_pos(self):
        self._env._export()
        return self._env._bufs["JOINT_POS"]

    @property
    def joint_vel(self):
        self._env._export()
        return self._env._bufs["JOINT_VEL"]

    @property
    def default_joint_pos(self):
        e = self._env
        return torch.tensor(list(e.desc.model.default_joint_pos)[: e.num_actions], device=e.device).repeat(e.num_envs, 1)

    applied_torque = property(lambda self: self._env._bufs["JOINT_TORQUE"])
    joint_acc = property(lambda self: self._env._bufs["JOINT_ACC"])


class _Articulation:
    def __init__(self, env):
        self.data = _ArticulationData(env)
        self.joint_names = list(env.desc.joint_names)
        self.body_names = list(env.desc.body_names)
        self.num_joints = len(self.joint_names)
        self.num_bodies = len(self.body_names)

    def find_joints(self, name_keys, preserve_order=False):
        from .model.build import find_names
        ids = find_names(name_keys, self.joint_names, preserve_order)
        return ids, [self.joint_names[i] for i in ids]

    def find_bodies(self, name_keys, preserve_order=False):
        from .model.build import find_names
        ids = find_names(name_keys, self.body_names, preserve_order)
        return ids, [self.body_names[i] for i in ids]


class _Scene(dict):
    def __init__(self, env):
        super().__init__(robot=_Articulation(env))
        self._env = env
        self.sensors = {}

    num_envs = property(lambda self: self._env.num_envs)
    env_origins = property(lambda self: (self._env._export(), self._env._bufs["ENV_ORIGIN"])[1])


class _CommandManager:
    def __init__(self, env):
        self._env = env

    def get_command(self, name):
        return self._env._bufs["COMMAND"]


class _RewardManager:
    def __init__(self, env):
        self._env = env
        self.active_terms = list(env.desc.reward_names)

    @property
    def _episode_sums(self):  # read by VEL/mdp/curriculums.py:44
        e = self._env
        return {n: e._bufs["EPISODE_SUMS"][i, : e.num_envs] for i, n in enumerate(self.active_terms)}


class ManagerBasedRLEnv(_EnvBase):
    """MI355X-native vectorised velocity-tracking environment.  See module docstring."""

    is_vector_env = True
    metadata = {"render_modes": [None]}

    def __init__(self, cfg=None, render_mode=None, *, desc: EnvDesc | None = None, extra: dict | None = None,
                 num_envs: int | None = None, seed: int | None = None, device: str | None = None, terrain_seed: int = 0,
                 lib_path: str | None = None, inspection: bool = False, specialise: bool | None = None, **kwargs):
        self._task_name = cfg if isinstance(cfg, str) else (type(cfg).__name__ if cfg is not None else "task")
        if isinstance(cfg, str):  # a compiled descriptor bundle id / path (robot_lab_amd/data)
            desc, extra = load_bundle(cfg)
            cfg = None
        if cfg is not None:
            from .model.cfg_compile import compile_cfg
            desc, spec = compile_cfg(cfg)
            extra = dict(terrain_generator=spec.get("terrain_generator"), env_spacing=spec.get("env_spacing"))
            num_envs = num_envs or cfg.scene.num_envs
            device = device or getattr(cfg.sim, "device", None)
            seed = seed if seed is not None else getattr(cfg, "seed", None)
        if desc is None:
            raise ValueError("ManagerBasedRLEnv needs a task cfg, a descriptor bundle id, or desc=")
        self.cfg = cfg
        self.desc = desc
        self.render_mode = render_mode
        self.num_envs = int(num_envs or 4096)
        self._seed = 42 if seed is None else int(seed)
        from .dist import physical_device

        dev = physical_device(device or "cuda:0")  # identity, except under the RL_SHARE_GPU=1 self-test aid (robot_lab_amd/dist.py)
        if dev.type != "cuda":
            raise RlEnvError(f"robot_lab_amd runs on MI355X only (device={dev}); there is no CPU path")
        if not torch.cuda.is_available():
            raise RlEnvError("no HIP device visible: robot_lab_amd has no CPU path")
        if dev.index is not None and dev.index >= torch.cuda.device_count():
            raise RlEnvError(f"device {dev} requested but {torch.cuda.device_count()} HIP device(s) are visible (one process per GPU: "
                             f"launch as many ranks as there are devices)")
        self.device = str(dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device()))
        self._dev_index = torch.device(self.device).index
        heights, terrain_origins, env_origins = build_world(desc, extra or {}, self.num_envs, terrain_seed)
        with torch.cuda.device(self._dev_index):
            self._native = NativeEnv(desc, heights, terrain_origins, env_origins, self.num_envs, self._seed, self._dev_index, lib_path)
            # `specialise` (None: RL_ENV_JIT, default on): a task the library has no specialised step kernel for gets one compiled now - or loaded from
            # the cache - and the env is created again on it (robot_lab_amd/jit.py; any failure there: one log line and this interpreter env)
            from . import jit

            if jit.enabled(specialise) and self._native.spec_id() == 0 and os.environ.get("RL_ENV_SPEC", "1") != "0":
                if jit.specialise(self._native.lib, desc, os.path.basename(str(self._task_name)), 16 // max(1, self._native.envs_per_wavefront())):
                    self._native.close()
                    self._native = NativeEnv(desc, heights, terrain_origins, env_origins, self.num_envs, self._seed, self._dev_index, lib_path)
        self.num_actions = self._native.num_actions
        self.max_episode_length = self._native.max_episode_length
        self.max_episode_length_s = float(desc.task.episode_length_s)
        self.physics_dt = float(desc.sim.dt)
        self.step_dt = float(desc.sim.dt) * int(desc.sim.decimation)
        self.common_step_counter = 0
        self._live_logs = collections.deque()
        self._bufs = _Buffers(self)
        for name in ("OBS_POLICY_RING", "OBS_CRITIC_RING", "REWARD", "TERMINATED", "TIME_OUT", "EPISODE_LENGTH", "ROOT_STATE", "JOINT_POS",
                     "JOINT_VEL", "REWARD_TERMS", "EPISODE_SUMS", "COMMAND", "CONTACT_TIMERS", "LOG", "ACTION", "ENV_ORIGIN", "TERRAIN_LEVEL",
                     "TASK_STATE", "GAINS"):
            self._bufs[name]
        if inspection:  # applied torque / joint acceleration / contact force views, filled by every step from now on
            for name in _INSPECTION:
                self._bufs[name]
        self._terminated = self._bufs["TERMINATED"].view(torch.bool)
        self._time_outs = self._bufs["TIME_OUT"].view(torch.bool)
        # the two observation groups alternate between two HBM buffers: views of both, picked after every step / reset
        n = self.num_envs
        self._obs_slots = [{"policy": self._bufs["OBS_POLICY_RING"][s, :n], "critic": self._bufs["OBS_CRITIC_RING"][s, :n]} for s in (0, 1)]
        self._obs = self._obs_slots[self._native.obs_slot()]
        self._export_stamp = -1
        self.scene = _Scene(self)
        self.command_manager = _CommandManager(self)
        self.reward_manager = _RewardManager(self)
        self.extras: dict = {}
        self.log_episodes = True
        g = gym
        if g is None:  # gymnasium (or its shim) may have become importable after this module was first imported
            try:
                import gymnasium as g
            except Exception:  # noqa: BLE001
                g = None
        if g is not None:
            sp = g.spaces
            inf = float("inf")
            self.single_observation_space = sp.Dict({k: sp.Box(-inf, inf, (v.shape[1],)) for k, v in self._obs.items()})
            self.single_action_space = sp.Box(-inf, inf, (self.num_actions,))
            self.observation_space = sp.Dict({k: sp.Box(-inf, inf, tuple(v.shape)) for k, v in self._obs.items()})
            self.action_space = sp.Box(-inf, inf, (self.num_envs, self.num_actions))

    # ------------------------------------------------------------------ gym surface
    @property
    def unwrapped(self):
        return self

    def _stream(self) -> int:
        return torch.cuda.current_stream(self._dev_index).cuda_stream

    def seed(self, seed: int = -1) -> int:
        return self._seed

    def reset(self, seed: int | None = None, options=None, env_ids=None):
        ids = None if env_ids is None else torch.as_tensor(env_ids).cpu().numpy()
        self._native.reset(ids, self._stream())
        self._export_stamp = -1  # state changed without a step: exported AoS views are stale
        self._obs = self._obs_slots[self._native.obs_slot()]
        self.extras = {}
        return self._obs, self.extras

    def step(self, action: torch.Tensor, rollout=None, gamma: float = 0.99, defer_bootstrap: bool = False):
        """`rollout`: a `robot_lab_amd.rollout.RolloutStorage` whose `act()` produced `action` - the env kernel then also
        writes the transition's rewards (+ time-out bootstrap) and dones into its current slot (`rl_env_step_record`), i.e.
        `rollout.process_env_step(...)` without a launch of its own.  `defer_bootstrap`: the values of this step are not
        needed here (the critic may still be running on another stream); `rollout.compute_returns` bootstraps the time outs."""
        if action.device != self._bufs["REWARD"].device or action.dtype != torch.float32 or not action.is_contiguous():
            action = action.to(device=self.device, dtype=torch.float32).contiguous()
        if action.shape != (self.num_envs, self.num_actions):
            raise ValueError(f"action shape {tuple(action.shape)} != {(self.num_envs, self.num_actions)}")
        if rollout is None:
            self._native.step(action.data_ptr(), self._stream())
        else:
            v, r, d = rollout.record_slots()
            self._native.step_record(action.data_ptr(), 0 if defer_bootstrap else v, r, d, gamma, self._stream())
        self.common_step_counter += 1
        self._obs = self._obs_slots[self._native.obs_slot()]
        if self.log_episodes:  # no snapshot, no memset: views of this step's ring slot and its predecessor, read only if somebody asks
            k = self._native.log_slot()
            log = _LazyLog(self, self._bufs["LOG"][k], self._bufs["LOG"][(k - 1) % RL_LOG_RING], self.common_step_counter)
            self.extras = {"log": log}
            self._retire_logs(log)
        else:
            self.extras = {}
        return self._obs, self._bufs["REWARD"], self._terminated, self._time_outs, self.extras

    def _retire_logs(self, new_log=None):
        """A log dict somebody still holds is materialised (device-side copies, no host sync) a few steps before its ring slot is
        reused, so `extras["log"]` stays readable however late it is read; logs nobody kept cost nothing."""
        q = self._live_logs
        if new_log is not None:
            q.append((self.common_step_counter, weakref.ref(new_log)))
        while q and self.common_step_counter - q[0][0] >= RL_LOG_RING - 4:
            _, ref = q.popleft()
            log = ref()
            if log is not None:
                log._fill()

    def get_observations(self):
        return self._obs

    # -- hipGraph capture of a loop around step() (include/rl_env.h; robot_lab_amd/collect.py drives it) -------------------
    def _fill_live_logs(self):
        for _, ref in self._live_logs:
            log = ref()
            if log is not None:
                log._fill()
        self._live_logs.clear()

    def graph_begin(self):
        self._fill_live_logs()  # before the capture: a _fill() inside it would be RECORDED into the graph instead of executed
        self._native.graph_begin(self._stream())
        self._graph_snap = self.common_step_counter

    def graph_end(self) -> int:
        n = self._native.graph_end(self._stream())
        self.common_step_counter = self._graph_snap  # capturing ran nothing
        self._obs = self._obs_slots[self._native.obs_slot()]
        return n

    def graph_launching(self, n: int):
        """Call right before replaying a captured loop of `n` steps: accounts them on the host side."""
        self._fill_live_logs()  # the replay advances the ring by n slots: what is still held is materialised now (ahead of it on the stream)
        self._native.graph_launching(self._stream())
        self.common_step_counter += n
        self._export_stamp = -1
        self._obs = self._obs_slots[self._native.obs_slot()]
        if self.log_episodes:  # the log of the replay's last step: tracked like a step()'s, so that later replays materialise it in time
            k = self._native.log_slot()
            log = _LazyLog(self, self._bufs["LOG"][k], self._bufs["LOG"][(k - 1) % RL_LOG_RING], self.common_step_counter)
            self.extras = {"log": log}
            self._live_logs.append((self.common_step_counter, weakref.ref(log)))

    @property
    def step_kernel(self) -> str:
        """Which step kernel this env launches: the one specialised on its task (csrc/env_spec.h; picked only when the env's compiled
        tables equal the Spec's constants bit for bit) or the term-stack interpreter (any other task, any edited cfg)."""
        sid = self._native.spec_id() if getattr(self, "_native", None) is not None else -1
        return f"specialised (spec_id {sid})" if sid > 0 else ("interpreter" if sid == 0 else "closed")

    def __repr__(self):
        lanes = 64 // max(1, self._native.envs_per_wavefront()) if getattr(self, "_native", None) is not None else 0
        return (f"<robot_lab_amd ManagerBasedRLEnv num_envs={self.num_envs} device={self.device} actions={self.num_actions} "
                f"step_kernel={self.step_kernel} lanes_per_env={lanes}>")

    def close(self):
        if getattr(self, "_native", None) is not None:
            torch.cuda.synchronize(self._dev_index)
            self._native.close()
            self._native = None

    def render(self, recompute=False):
        return None

    # ------------------------------------------------------------------ attributes callers touch
    @property
    def episode_length_buf(self) -> torch.Tensor:
        return self._bufs["EPISODE_LENGTH"]

    @episode_length_buf.setter
    def episode_length_buf(self, value):  # rsl_rl `init_at_random_ep_len` (train.py:224)
        self._bufs["EPISODE_LENGTH"].copy_(torch.as_tensor(value, device=self.device).to(torch.int64))

    @property
    def terrain_levels(self) -> torch.Tensor:
        return self._bufs["TERRAIN_LEVEL"]

    @property
    def command_levels(self) -> torch.Tensor:
        """float[16]: live command ranges of the command_levels_* curricula (lin_vel_x lo/hi, lin_vel_y lo/hi, ang_vel_z lo/hi, ...)."""
        return self._bufs["CMD_LEVELS"]

    @property
    def reward_buf(self):
        return self._bufs["REWARD"]

    @property
    def reset_buf(self):
        return self._terminated | self._time_outs

    def reward_terms(self) -> torch.Tensor:
        """[T, N] weighted per-term rewards of the last step."""
        return self._bufs["REWARD_TERMS"][:, : self.num_envs]

    def _export(self):
        if self._export_stamp != self.common_step_counter:
            self._native.export_state(self._stream())
            self._export_stamp = self.common_step_counter

    # the state step() carries between calls, as torch views of the AoS buffers `rl_env_export_state` fills
    STATE_BUFFERS = ("ROOT_STATE", "JOINT_POS", "JOINT_VEL", "ACTION", "GAINS", "CONTACT_TIMERS", "TASK_STATE", "ENV_ORIGIN")

    def read_state(self) -> dict:
        """Full carried state as a dict of host arrays: the AoS buffers above plus episode length, episode sums, terrain
        levels and the step counter (tests/test_gpu_teacher_forced.py; `rl_env_export_state`)."""
        self._export_stamp = -1
        self._export()
        out = {k.lower(): self._bufs[k].cpu().numpy().copy() for k in self.STATE_BUFFERS}
        out["episode_length"] = self._bufs["EPISODE_LENGTH"].cpu().numpy().copy()
        out["episode_sums"] = self._bufs["EPISODE_SUMS"][:, : self.num_envs].cpu().numpy().copy()
        out["terrain_level"] = self._bufs["TERRAIN_LEVEL"].cpu().numpy().copy()
        out["step_count"] = self._native.step_count
        return out

    def load_state(self, state: dict):
        """Inverse of `read_state` (`rl_env_commit_state`): any subset of its keys."""
        self._export_stamp = -1
        self._export()  # parts that are not given keep their current values
        for k in self.STATE_BUFFERS:
            if k.lower() in state:
                self._bufs[k].copy_(torch.as_tensor(np.asarray(state[k.lower()]), device=self.device).reshape(self._bufs[k].shape))
        if "episode_length" in state:
            self.episode_length_buf = state["episode_length"]
        if "episode_sums" in state:
            self._bufs["EPISODE_SUMS"][:, : self.num_envs].copy_(torch.as_tensor(np.asarray(state["episode_sums"]), device=self.device))
        if "terrain_level" in state:
            self._bufs["TERRAIN_LEVEL"].copy_(torch.as_tensor(np.asarray(state["terrain_level"]), device=self.device).to(torch.int32))
        if "step_count" in state:
            self._native.step_count = int(state["step_count"])
            self.common_step_counter = int(state["step_count"])
        self._native.commit_state(self._stream())
        self._export_stamp = -1

    def write_state(self, root_state=None, joint_pos=None, joint_vel=None):
        """Overwrite root / joint state from host arrays (`rl_env_import_state`: export, overwrite, commit)."""
        keep = []
        ptrs = []
        for a in (root_state, joint_pos, joint_vel):
            if a is None:
                ptrs.append(0)
            else:
                a = np.ascontiguousarray(a, dtype=np.float32)
                keep.append(a)
                ptrs.append(a.ctypes.data)
        self._native.import_state(ptrs[0], ptrs[1], ptrs[2], self._stream())
        self._export_stamp = -1
