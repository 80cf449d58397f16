"""ctypes mirror of ``include/rl_env.h`` (the env descriptor handed across the C-ABI) + JSON IO.

The descriptor is *data*: what the reference expresses as ``ArticulationCfg`` + URDF
(``SRC/robot_lab/assets/unitree.py``) and as the manager term lists of
``VEL/velocity_env_cfg.py`` / ``VEL/config/<robot>/rough_env_cfg.py``, flattened to fixed-size
arrays so that a native caller can fill it without Python.
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

RL_MAX_LINKS = 33
RL_MAX_DOF = 32
RL_MAX_BODIES = 48
RL_MAX_SPHERES = 96
RL_MAX_CAPSULES = 16
RL_MAX_SELF_PAIRS = 80
RL_MAX_REWARD_TERMS = 40
RL_MAX_OBS_TERMS = 12
RL_TERM_NPARAM = 8
RL_LOG_SIZE = 64
RL_LOG_RING = 64
RL_LOG_PARTS = 32  # partial rows of a ring slot (include/rl_env.h RL_BUF_LOG): a reader sums them

REWARD_KINDS = [
    "track_lin_vel_xy_exp", "track_ang_vel_z_exp", "lin_vel_z_l2", "ang_vel_xy_l2", "joint_torques_l2",
    "joint_acc_l2", "joint_pos_limits", "joint_power", "stand_still", "joint_pos_penalty", "joint_mirror",
    "action_rate_l2", "undesired_contacts", "contact_forces", "feet_contact_without_cmd", "feet_height_body",
    "upward", "feet_air_time", "feet_air_time_variance_penalty", "feet_slide", "GaitReward",
    "flat_orientation_l2", "is_terminated", "joint_deviation_l1", "joint_vel_l2", "feet_contact",
    "feet_stumble", "feet_height", "track_lin_vel_xy_yaw_frame_exp", "track_ang_vel_z_world_exp",
    "feet_air_time_positive_biped", "handstand_feet_height_exp", "handstand_feet_on_air", "handstand_feet_air_time",
    "handstand_orientation_l2", "base_height_l2", "wheel_vel_penalty", "feet_distance_y_exp",
    "feet_distance_xy_exp", "action_mirror", "action_sync",
]
REW = {n: i for i, n in enumerate(REWARD_KINDS)}
OBS_KINDS = [
    "base_lin_vel", "base_ang_vel", "projected_gravity", "generated_commands", "joint_pos_rel",
    "joint_vel_rel", "last_action", "height_scan", "joint_pos_rel_without_wheel",
]
OBS = {n: i for i, n in enumerate(OBS_KINDS)}

BUF = dict(
    OBS_POLICY=0, OBS_CRITIC=1, REWARD=2, TERMINATED=3, TIME_OUT=4, EPISODE_LENGTH=5, ROOT_STATE=6,
    JOINT_POS=7, JOINT_VEL=8, REWARD_TERMS=9, EPISODE_SUMS=10, COMMAND=11, CONTACT_FORCE=12,
    CONTACT_TIMERS=13, LOG=14, ACTION=15, JOINT_TORQUE=16, JOINT_ACC=17, ENV_ORIGIN=18, TERRAIN_LEVEL=19,
    TASK_STATE=20, GAINS=21, OBS_POLICY_RING=22, OBS_CRITIC_RING=23, CMD_LEVELS=24,
)
# fields of one RL_BUF_TASK_STATE row (include/rl_env.h rl_task_state_field)
TASK_STATE = dict(CMD=slice(0, 3), HEADING_TARGET=3, CMD_TIME_LEFT=4, METRIC_XY=5, METRIC_YAW=6, PUSH_TIME_LEFT=7,
                  IS_HEADING_ENV=8, IS_STANDING_ENV=9, EXT_FORCE=slice(10, 13), EXT_TORQUE=slice(13, 16))
RL_TASK_STATE_NF = 16

# The reals of the C-ABI are `float` (include/rl_env.h).  RL_ABI_REAL=f64 mirrors the header of the fp64 instantiation of the CPU lane
# emulator instead (tests/emu/make_f64.py retypes a copy of the sources: test infrastructure, its own process).  The product library's
# rl_env_desc_size() then disagrees with this mirror and capi.load_library refuses it.
REAL_F64 = os.environ.get("RL_ABI_REAL", "f32") == "f64"
REAL_C, REAL_NP = (C.c_double, np.float64) if REAL_F64 else (C.c_float, np.float32)
f32, i32, u32, u64 = REAL_C, C.c_int32, C.c_uint32, C.c_uint64


class RewardTerm(C.Structure):
    _fields_ = [
        ("kind", i32), ("weight", f32), ("p", f32 * RL_TERM_NPARAM), ("joint_mask", u32), ("body_mask", u64),
        ("idx_a", i32 * 16), ("idx_b", i32 * 16), ("n_idx", i32),
    ]


class ObsTerm(C.Structure):
    _fields_ = [
        ("kind", i32), ("scale", f32), ("clip_lo", f32), ("clip_hi", f32), ("noise_lo", f32), ("noise_hi", f32),
        ("has_noise", i32),
    ]


class ModelDesc(C.Structure):
    _fields_ = [
        ("num_links", i32), ("num_dof", i32), ("num_bodies", i32), ("num_spheres", i32), ("num_chains", i32),
        ("chain_len", i32),
        ("chain_link", (i32 * 8) * 4), ("chain_nj", i32 * 4), ("chain_attach", i32 * 4),
        ("num_trunk", i32), ("trunk_link", i32 * 8),
        ("link_parent", i32 * RL_MAX_LINKS),
        ("link_origin", (f32 * 3) * RL_MAX_LINKS),
        ("link_quat", (f32 * 4) * RL_MAX_LINKS),
        ("link_axis", (f32 * 3) * RL_MAX_LINKS),
        ("joint_lower", f32 * RL_MAX_DOF), ("joint_upper", f32 * RL_MAX_DOF),
        ("joint_vel_limit", f32 * RL_MAX_DOF),
        ("joint_armature", f32 * RL_MAX_DOF),
        ("default_joint_pos", f32 * RL_MAX_DOF), ("default_joint_vel", f32 * RL_MAX_DOF),
        ("soft_lower", f32 * RL_MAX_DOF), ("soft_upper", f32 * RL_MAX_DOF),
        ("body_link", i32 * RL_MAX_BODIES),
        ("body_pos", (f32 * 3) * RL_MAX_BODIES),
        ("body_mass", f32 * RL_MAX_BODIES),
        ("body_com", (f32 * 3) * RL_MAX_BODIES),
        ("body_inertia", (f32 * 6) * RL_MAX_BODIES),
        ("sphere_body", i32 * RL_MAX_SPHERES),
        ("sphere_center", (f32 * 3) * RL_MAX_SPHERES),
        ("sphere_radius", f32 * RL_MAX_SPHERES),
        ("default_root_pos", f32 * 3),
        ("default_root_quat", f32 * 4),
        ("act_implicit", i32 * RL_MAX_DOF),
        ("act_kp", f32 * RL_MAX_DOF), ("act_kd", f32 * RL_MAX_DOF),
        ("act_effort_limit", f32 * RL_MAX_DOF), ("act_saturation", f32 * RL_MAX_DOF),
        ("act_vel_limit", f32 * RL_MAX_DOF),
        ("action_is_vel", i32 * RL_MAX_DOF),
        ("action_scale", f32 * RL_MAX_DOF), ("action_offset", f32 * RL_MAX_DOF),
        ("action_clip_lo", f32 * RL_MAX_DOF), ("action_clip_hi", f32 * RL_MAX_DOF),
        ("self_collision", i32), ("num_capsules", i32), ("capsule_link", i32 * RL_MAX_CAPSULES),
        ("capsule_p0", (f32 * 3) * RL_MAX_CAPSULES), ("capsule_p1", (f32 * 3) * RL_MAX_CAPSULES), ("capsule_radius", f32 * RL_MAX_CAPSULES),
        ("num_self_pairs", i32), ("self_pair", (i32 * 2) * RL_MAX_SELF_PAIRS),
        ("chain_grp0", i32 * 4), ("trunk_parent", i32 * 8),
    ]


class SimDesc(C.Structure):
    _fields_ = [
        ("dt", f32), ("decimation", i32), ("gravity", f32), ("contact_k", f32), ("contact_c", f32),
        ("contact_phi_ref", f32), ("contact_ct", f32), ("contact_vdep", f32), ("contact_vstick", f32),
        ("limit_k", f32), ("limit_c", f32), ("force_threshold", f32), ("self_k", f32),
    ]


class TerrainDesc(C.Structure):
    _fields_ = [
        ("is_plane", i32), ("nx", i32), ("ny", i32), ("hscale", f32), ("x0", f32), ("y0", f32),
        ("num_rows", i32), ("num_cols", i32), ("tile_size", f32), ("border", f32), ("max_init_level", i32),
        ("curriculum", i32),
    ]


class TaskDesc(C.Structure):
    _fields_ = [
        ("episode_length_s", f32),
        ("cmd_range", (f32 * 2) * 4), ("cmd_resample", f32 * 2),
        ("cmd_rel_standing", f32), ("cmd_rel_heading", f32), ("cmd_heading_stiffness", f32),
        ("cmd_heading", i32), ("cmd_small_threshold", f32),
        ("n_policy", i32), ("n_critic", i32),
        ("policy", ObsTerm * RL_MAX_OBS_TERMS), ("critic", ObsTerm * RL_MAX_OBS_TERMS),
        ("policy_corrupt", i32), ("critic_corrupt", i32),
        ("scan_nx", i32), ("scan_ny", i32), ("scan_res", f32), ("scan_offset", f32), ("scan_body", i32),
        ("wheel_joint_mask", u32),
        ("n_rewards", i32),
        ("rewards", RewardTerm * RL_MAX_REWARD_TERMS),
        ("term_time_out", i32), ("term_out_of_bounds", i32), ("term_illegal_contact", i32),
        ("oob_buffer", f32), ("illegal_body_mask", u64), ("illegal_threshold", f32),
        ("ev_material", i32), ("ev_mass_base", i32), ("ev_mass_others", i32), ("ev_com", i32), ("ev_wrench", i32),
        ("ev_reset_joints", i32), ("ev_gains", i32), ("ev_reset_base", i32), ("ev_push", i32),
        ("friction_static", f32 * 2), ("friction_dynamic", f32 * 2), ("restitution", f32 * 2),
        ("friction_buckets", i32),
        ("mass_base_add", f32 * 2), ("mass_base_mask", u64),
        ("mass_scale", f32 * 2), ("mass_scale_mask", u64),
        ("com_range", (f32 * 2) * 3), ("com_mask", u64),
        ("wrench_force", f32 * 2), ("wrench_torque", f32 * 2),
        ("reset_joint_pos_scale", f32 * 2), ("reset_joint_vel_scale", f32 * 2),
        ("gain_kp_scale", f32 * 2), ("gain_kd_scale", f32 * 2),
        ("reset_pose", (f32 * 2) * 6), ("reset_vel", (f32 * 2) * 6),
        ("push_interval", f32 * 2), ("push_vel", (f32 * 2) * 6),
        ("base_body", i32),
        ("cur_cmd_lin", i32), ("cur_cmd_ang", i32), ("cur_cmd_lin_term", i32), ("cur_cmd_ang_term", i32),
        ("cur_cmd_lin_mult", f32 * 2), ("cur_cmd_ang_mult", f32 * 2),
    ]


class EnvDesc(C.Structure):
    _fields_ = [("model", ModelDesc), ("sim", SimDesc), ("terrain", TerrainDesc), ("task", TaskDesc)]

    # names are not part of the C-ABI; they ride along for the Python boundary (find_joints etc.)
    joint_names: list
    body_names: list
    reward_names: list

    def obs_dim(self, group: int) -> int:
        t = self.task
        terms, n = (t.policy, t.n_policy) if group == 0 else (t.critic, t.n_critic)
        D = self.model.num_dof
        dims = {0: 3, 1: 3, 2: 3, 3: 3, 4: D, 5: D, 6: D, 7: t.scan_nx * t.scan_ny, 8: D}
        return int(sum(dims[terms[i].kind] for i in range(n)))


def _to_py(obj):
    if isinstance(obj, C.Structure):
        return {n: _to_py(getattr(obj, n)) for n, _ in obj._fields_}
    if isinstance(obj, C.Array):
        return [_to_py(v) for v in obj]
    return obj


def _from_py(obj, val):
    """Fill ctypes object in place from nested python lists/dicts (arrays may be shorter)."""
    if isinstance(obj, C.Structure):
        for n, _ in obj._fields_:
            if n in val:
                sub = getattr(obj, n)
                if isinstance(sub, (C.Structure, C.Array)):
                    _from_py(sub, val[n])
                else:
                    setattr(obj, n, val[n])
    else:
        for i, v in enumerate(val):
            if isinstance(obj[i], (C.Structure, C.Array)):
                _from_py(obj[i], v)
            else:
                obj[i] = v


def desc_to_json(desc: EnvDesc) -> str:
    d = _to_py(desc)
    d["_names"] = dict(
        joints=list(getattr(desc, "joint_names", [])), bodies=list(getattr(desc, "body_names", [])),
        rewards=list(getattr(desc, "reward_names", [])),
    )
    return json.dumps(d)


def desc_from_json(s: str) -> EnvDesc:
    d = json.loads(s)
    desc = EnvDesc()
    _from_py(desc, d)
    names = d.get("_names", {})
    desc.joint_names = names.get("joints", [])
    desc.body_names = names.get("bodies", [])
    desc.reward_names = names.get("rewards", [])
    return desc


def arr(cobj, n=None) -> np.ndarray:
    """numpy view (no copy) of a ctypes array field, optionally truncated to the first n rows."""
    a = np.ctypeslib.as_array(cobj)
    return a if n is None else a[:n]


def set_arr(cobj, values):
    a = np.ctypeslib.as_array(cobj)
    v = np.asarray(values)
    a[tuple(slice(0, s) for s in v.shape)] = v


def mask_of(indices) -> int:
    m = 0
    for i in indices:
        m |= 1 << int(i)
    return m
