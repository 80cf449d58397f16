"""Reference cfg object (`LocomotionVelocityRoughEnvCfg` subclasses, `VEL/velocity_env_cfg.py:695`)
-> neutral spec dict -> `EnvDesc`.  This is where the reference's declarative task description
(L3 in SURVEY.md section 1) is turned into the data the HIP kernels consume.

Terms are recognised by `func.__name__`; `None` entries are deleted terms
(`.../unitree_a1/rough_env_cfg.py:44-45,153,158-159`); zero-weight rewards are already pruned by the
reference's `disable_zero_weight_rewards()` (`velocity_env_cfg.py:737-743`) and any that remain with
weight 0 are skipped exactly like the upstream RewardManager does.
"""
from __future__ import annotations

import os

from ..desc import OBS, REW, EnvDesc
from .build import build_desc
from .urdf import load_urdf


class UnsupportedTerm(ValueError):
    pass


def _terms(cfg_group):
    """(name, term_cfg) of a cfg container in declaration order, skipping deleted (None) terms."""
    out = []
    for k, v in (vars(cfg_group) if hasattr(cfg_group, "__dict__") else {}).items():
        if k.startswith("_") or v is None or not hasattr(v, "func"):
            continue
        out.append((k, v))
    return out


def _fname(term):
    f = term.func
    return getattr(f, "__name__", type(f).__name__)


def _names(entity, key):
    v = getattr(entity, key, None)
    if v is None:
        return None
    return [v] if isinstance(v, str) else list(v)


def compile_spec(cfg) -> tuple[dict, str]:
    """Returns (spec, urdf_path)."""
    robot = cfg.scene.robot
    urdf_path = robot.spawn.asset_path
    actuators = []
    for name, a in robot.actuators.items():
        kind = type(a).__name__
        if kind not in ("DCMotorCfg", "ImplicitActuatorCfg", "IdealPDActuatorCfg"):
            raise UnsupportedTerm(f"actuator model {kind}")
        implicit = kind == "ImplicitActuatorCfg"
        actuators.append(dict(
            type="implicit" if implicit else "dc",
            joint_names_expr=list(a.joint_names_expr),
            stiffness=a.stiffness, damping=a.damping,
            effort_limit=(a.effort_limit_sim if implicit and a.effort_limit_sim is not None else a.effort_limit),
            saturation_effort=(a.saturation_effort if kind == "DCMotorCfg" else None),
            velocity_limit=(a.velocity_limit_sim if implicit and a.velocity_limit_sim is not None else a.velocity_limit),
            armature=a.armature,
        ))
        if actuators[-1]["saturation_effort"] is None:
            actuators[-1]["saturation_effort"] = actuators[-1]["effort_limit"]
    spec = dict(robot=dict(
        init_pos=tuple(robot.init_state.pos), init_rot=tuple(robot.init_state.rot or (1.0, 0.0, 0.0, 0.0)),
        init_joint_pos=dict(robot.init_state.joint_pos), init_joint_vel=dict(robot.init_state.joint_vel or {".*": 0.0}),
        soft_joint_pos_limit_factor=robot.soft_joint_pos_limit_factor or 1.0, actuators=actuators))
    # joints no actuator group names keep the importer's drive gains; with the assets' `PDGainsCfg(stiffness=0, damping=0)`
    # (assets/booster.py:29-31) they are passive hinges whose action-vector entries have no effect
    gains = getattr(getattr(robot.spawn, "joint_drive", None), "gains", None)
    spec["robot"]["unactuated_passive"] = bool(gains is not None and gains.stiffness == 0 and gains.damping == 0)
    # ArticulationRootPropertiesCfg.enabled_self_collisions (assets/unitree.py:482 G1, assets/roboparty.py:33 ATOM01: True)
    spec["robot"]["self_collisions"] = bool(getattr(getattr(robot.spawn, "articulation_props", None), "enabled_self_collisions", False))
    # actions
    actions = []
    for name, a in vars(cfg.actions).items():
        if a is None or not hasattr(a, "joint_names"):
            continue
        kind = type(a).__name__
        if kind not in ("JointPositionActionCfg", "JointVelocityActionCfg"):
            raise UnsupportedTerm(f"action term {kind}")
        actions.append(dict(type="pos" if kind == "JointPositionActionCfg" else "vel", joint_names=list(a.joint_names),
                            preserve_order=bool(a.preserve_order), scale=a.scale, clip=a.clip,
                            use_default_offset=bool(a.use_default_offset)))
    spec["actions"] = actions
    spec["sim"] = dict(dt=cfg.sim.dt, decimation=cfg.decimation)
    # terrain
    # curricula: terrain_levels_vel is built in; the command_levels_* terms (mdp/curriculums.py:21-94) become the live range
    # table of the kernels (rl_task_desc.cur_cmd_*); every shipped cfg of the reference deletes them (e.g.
    # unitree_a1/rough_env_cfg.py:158-159), the base cfg (velocity_env_cfg.py:673-690) carries them
    cmd_levels = {}
    for name, t in _terms(getattr(cfg, "curriculum", None) or object()):
        fn = _fname(t)
        if fn in ("command_levels_lin_vel", "command_levels_ang_vel"):
            p = t.params or {}
            cmd_levels[fn[len("command_levels_"):]] = dict(reward_term_name=p["reward_term_name"],
                                                          range_multiplier=tuple(p.get("range_multiplier", (0.1, 1.0))))
        elif fn != "terrain_levels_vel":
            raise UnsupportedTerm(f"curriculum term {fn} ({name})")
    spec["command_levels"] = cmd_levels
    ter = cfg.scene.terrain
    if ter.terrain_type == "plane":
        spec["terrain"] = dict(is_plane=1)
        spec["terrain_generator"] = None
    else:
        g = ter.terrain_generator
        spec["terrain"] = dict(is_plane=0, num_rows=g.num_rows, num_cols=g.num_cols, tile_size=g.size[0], border=g.border_width,
                               max_init_level=ter.max_init_terrain_level if ter.max_init_terrain_level is not None else g.num_rows - 1,
                               curriculum=int(getattr(cfg.curriculum, "terrain_levels", None) is not None))
        spec["terrain_generator"] = dict(
            num_rows=g.num_rows, num_cols=g.num_cols, size=tuple(g.size), border_width=g.border_width,
            curriculum=bool(g.curriculum), difficulty_range=tuple(g.difficulty_range or (0.0, 1.0)),
            sub_terrains={k: {kk: vv for kk, vv in vars(v).items() if not callable(vv)} | {"kind": v.kind} for k, v in g.sub_terrains.items()})
    spec["env_spacing"] = cfg.scene.env_spacing
    # commands
    c = cfg.commands.base_velocity
    task = dict(episode_length_s=cfg.episode_length_s, command=dict(
        lin_vel_x=tuple(c.ranges.lin_vel_x), lin_vel_y=tuple(c.ranges.lin_vel_y), ang_vel_z=tuple(c.ranges.ang_vel_z),
        heading=tuple(c.ranges.heading or (0.0, 0.0)), resampling_time_range=tuple(c.resampling_time_range),
        rel_standing_envs=c.rel_standing_envs, rel_heading_envs=c.rel_heading_envs,
        heading_control_stiffness=c.heading_control_stiffness, heading_command=bool(c.heading_command),
        small_threshold=0.2 if type(c).__name__ == "UniformThresholdVelocityCommandCfg" else -1.0))
    # observations
    obs = {}
    for gname in ("policy", "critic"):
        grp = getattr(cfg.observations, gname, None)
        terms = []
        if grp is not None:
            for name, t in _terms(grp):
                fn = _fname(t)
                if fn not in OBS:
                    raise UnsupportedTerm(f"observation term {fn}")
                e = dict(func=fn, scale=t.scale, clip=t.clip, noise=(t.noise.n_min, t.noise.n_max) if t.noise is not None else None)
                if fn == "joint_pos_rel_without_wheel":
                    e["wheel_joint_names"] = _names(t.params["wheel_asset_cfg"], "joint_names")
                terms.append(e)
        obs[gname] = dict(terms=terms, enable_corruption=bool(getattr(grp, "enable_corruption", False)))
    task["observations"] = obs
    hs = getattr(cfg.scene, "height_scanner", None)
    if hs is not None:
        # the ray caster rides on the body its prim_path names (velocity_env_cfg.py:71; G1 moves it to
        # the torso, unitree_g1/rough_env_cfg.py:55)
        task["height_scan"] = dict(size=tuple(hs.pattern_cfg.size), resolution=hs.pattern_cfg.resolution, offset=0.5,
                                   body=str(hs.prim_path).rsplit("/", 1)[-1])
    # rewards
    rewards = []
    for name, t in _terms(cfg.rewards):
        if t.weight == 0:
            continue
        fn = _fname(t)
        if fn not in REW:
            raise UnsupportedTerm(f"reward term {fn} ({name})")
        p = dict(t.params or {})
        e = dict(name=name, func=fn, weight=float(t.weight), p=[])
        ac, sc = p.get("asset_cfg"), p.get("sensor_cfg")
        if ac is not None and _names(ac, "joint_names") is not None:
            e["joint_names"] = _names(ac, "joint_names")
        bn = (_names(sc, "body_names") if sc is not None else None) or (_names(ac, "body_names") if ac is not None else None)
        if bn is not None:
            e["body_names"] = bn
        if fn in ("track_lin_vel_xy_exp", "track_ang_vel_z_exp", "track_lin_vel_xy_yaw_frame_exp", "track_ang_vel_z_world_exp"):
            e["p"] = [p["std"] ** 2]
        elif fn == "stand_still":
            e["p"] = [p.get("command_threshold", 0.06)]
        elif fn == "joint_pos_penalty":
            e["p"] = [p["stand_still_scale"], p["velocity_threshold"], p["command_threshold"]]
        elif fn in ("joint_mirror", "action_mirror"):
            e["mirror_joints"] = [list(pair) for pair in p["mirror_joints"]]
        elif fn == "action_sync":
            e["joint_groups"] = [list(grp) for grp in p["joint_groups"]]
        elif fn == "wheel_vel_penalty":
            e["p"] = [p["velocity_threshold"], p["command_threshold"]]
        elif fn == "feet_distance_y_exp":
            e["p"] = [p["std"] ** 2, p["stance_width"]]
        elif fn == "feet_distance_xy_exp":
            e["p"] = [p["std"] ** 2, p["stance_width"], p["stance_length"]]
        elif fn == "handstand_feet_height_exp":
            e["p"] = [p["std"] ** 2, p["target_height"]]
        elif fn == "handstand_orientation_l2":
            e["p"] = list(p["target_gravity"])
        elif fn == "base_height_l2":
            sensor = p.get("sensor_cfg")
            if sensor is not None:  # the 3 x 3 `height_scanner_base` ray caster (velocity_env_cfg.py:78-85)
                hb = getattr(cfg.scene, getattr(sensor, "name", "height_scanner_base"), None)
                if hb is None or tuple(hb.pattern_cfg.size) != (0.1, 0.1) or hb.pattern_cfg.resolution != 0.05:
                    raise UnsupportedTerm("base_height_l2 with a ray caster other than the 0.1 x 0.1 @ 0.05 grid")
            e["p"] = [p["target_height"], 1.0 if sensor is not None else 0.0]
            e.pop("body_names", None)
        elif fn in ("undesired_contacts", "contact_forces", "feet_air_time", "feet_air_time_positive_biped", "handstand_feet_air_time"):
            e["p"] = [p["threshold"]]
        elif fn in ("feet_height_body", "feet_height"):
            e["p"] = [p["target_height"], p["tanh_mult"]]
        elif fn == "feet_contact":
            e["p"] = [p["expect_contact_num"]]
        elif fn == "GaitReward":
            e["p"] = [p["std"], p["max_err"], p["velocity_threshold"], p["command_threshold"]]
            e["synced_feet_pair_names"] = [list(x) for x in p["synced_feet_pair_names"]]
        rewards.append(e)
    task["rewards"] = rewards
    # terminations
    tm = {}
    for name, t in _terms(cfg.terminations):
        fn = _fname(t)
        if fn == "time_out":
            tm["time_out"] = True
        elif fn == "terrain_out_of_bounds":
            tm["terrain_out_of_bounds"] = dict(distance_buffer=t.params.get("distance_buffer", 3.0))
        elif fn == "illegal_contact":
            tm["illegal_contact"] = dict(body_names=_names(t.params["sensor_cfg"], "body_names"), threshold=t.params["threshold"])
        else:
            raise UnsupportedTerm(f"termination term {fn}")
    tm.setdefault("time_out", False)
    task["terminations"] = tm
    # events
    ev = {}
    base_name = None
    for name, t in _terms(cfg.events):
        fn, p = _fname(t), t.params or {}
        if fn == "randomize_rigid_body_material":
            ev["material"] = dict(static_friction_range=p["static_friction_range"], dynamic_friction_range=p["dynamic_friction_range"],
                                  restitution_range=p["restitution_range"], num_buckets=p["num_buckets"])
        elif fn == "randomize_rigid_body_mass":
            key = "mass_base" if p["operation"] == "add" else "mass_others"
            if p["operation"] not in ("add", "scale"):
                raise UnsupportedTerm("mass randomisation op " + p["operation"])
            ev[key] = dict(range=p["mass_distribution_params"], body_names=_names(p["asset_cfg"], "body_names"))
            if key == "mass_base":
                base_name = _names(p["asset_cfg"], "body_names")
        elif fn in ("randomize_rigid_body_com", "randomize_com_positions"):
            ev["com"] = dict(range=dict(p["com_range"]), body_names=_names(p["asset_cfg"], "body_names"))
        elif fn == "apply_external_force_torque":
            ev["wrench"] = dict(force_range=p["force_range"], torque_range=p["torque_range"])
            base_name = base_name or _names(p["asset_cfg"], "body_names")
        elif fn == "reset_joints_by_scale":
            ev["reset_joints"] = dict(position_range=p["position_range"], velocity_range=p["velocity_range"])
        elif fn == "randomize_actuator_gains":
            if p.get("operation", "scale") != "scale" or p.get("distribution", "uniform") != "uniform":
                raise UnsupportedTerm("actuator gain randomisation other than uniform scale")
            ev["gains"] = dict(stiffness=p["stiffness_distribution_params"], damping=p["damping_distribution_params"])
        elif fn == "reset_root_state_uniform":
            ev["reset_base"] = dict(pose_range=dict(p["pose_range"]), velocity_range=dict(p["velocity_range"]))
        elif fn == "push_by_setting_velocity":
            ev["push"] = dict(interval_range_s=t.interval_range_s, velocity_range=dict(p["velocity_range"]))
        else:
            raise UnsupportedTerm(f"event term {fn} ({name})")
    task["events"] = ev
    task["base_body_name"] = base_name or [getattr(cfg, "base_link_name", "base")]
    task["command_levels"] = spec.pop("command_levels")
    for k, c in task["command_levels"].items():
        if c["reward_term_name"] not in [r["name"] for r in task["rewards"]]:
            raise UnsupportedTerm(f"command_levels_{k}: reward term {c['reward_term_name']} is not active (weight 0 or removed)")
    spec["task"] = task
    is_regex = lambda n: any(c in n for c in "*()[]|?+^$")
    spec["joint_order"] = list(actions[0]["joint_names"]) if len(actions) == 1 and actions[0]["preserve_order"] else None
    if spec["joint_order"] is not None and any(is_regex(n) for n in spec["joint_order"]):
        spec["joint_order"] = getattr(cfg, "joint_names", None)  # G1: ".*" -> the importer's breadth-first order
    elif spec["joint_order"] is None:
        jo = []
        for a in actions:
            jo += [n for n in a["joint_names"]]
        spec["joint_order"] = jo if all("*" not in n and "(" not in n for n in jo) else getattr(cfg, "joint_names", None)
    return spec, urdf_path


def compile_cfg(cfg) -> tuple[EnvDesc, dict]:
    spec, urdf_path = compile_spec(cfg)
    if not os.path.isfile(urdf_path):
        raise FileNotFoundError(urdf_path)
    model = load_urdf(urdf_path, joint_order=spec["joint_order"])
    return build_desc(model, spec), spec
