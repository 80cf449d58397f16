#!/usr/bin/env python
"""Generate tests/golden/terms_<robot>.npz: outputs of the REFERENCE's own term functions
(`/root/reference/source/robot_lab/.../velocity/mdp/{rewards,commands,events}.py`, imported unchanged
through robot_lab_amd.shims) on a recorded simulator state.  The fixtures pin the oracle's
restatement of those terms (tests/test_terms_golden.py) and travel to the GPU box, where
/root/reference does not exist.

Run in the build container:  python tools/gen_golden_terms.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from robot_lab_amd import shims  # noqa: E402

shims.install(shims.REFERENCE_SOURCE)
import robot_lab.tasks  # noqa: E402,F401
from isaaclab.managers import SceneEntityCfg  # noqa: E402
from isaaclab.utils import math as mu  # noqa: E402
from isaaclab_tasks.utils import parse_env_cfg  # noqa: E402

from oracle.env import OracleEnv  # noqa: E402
from robot_lab_amd.model.build import find_names  # noqa: E402
from robot_lab_amd.model.cfg_compile import compile_cfg  # noqa: E402
from robot_lab_amd.scene import build_world, load_bundle  # noqa: E402

T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)  # noqa: E731


class Sensor:
    def __init__(self, ora, body_names):
        self.body_names = body_names
        self.data = types.SimpleNamespace(
            net_forces_w_history=T(ora.force_hist), net_forces_w=T(ora.contact_force),
            current_air_time=T(ora.timers[..., 0]), current_contact_time=T(ora.timers[..., 1]),
            last_air_time=T(ora.timers[..., 2]), last_contact_time=T(ora.timers[..., 3]))

    def compute_first_contact(self, dt, abs_tol=1.0e-8):  # [UPSTREAM ContactSensor]
        c = self.data.current_contact_time
        return (c > 0.0) & (c < dt + abs_tol)

    def compute_first_air(self, dt, abs_tol=1.0e-8):
        c = self.data.current_air_time
        return (c > 0.0) & (c < dt + abs_tol)

    def find_bodies(self, name_keys, preserve_order=False):
        ids = find_names(list(name_keys) if not isinstance(name_keys, str) else name_keys, self.body_names, preserve_order)
        return ids, [self.body_names[i] for i in ids]


def duck_env(ora, desc):
    """Duck-typed `env` over the oracle's recorded state; derived articulation data follows SURVEY.md B3
    using the shim's torch math (independent of oracle/spatial.py)."""
    st = ora.st
    quat, pos = T(st["root_quat"]), T(st["root_pos"])
    ang_w = T(st["root_ang_vel"])
    com_w = mu.quat_apply(quat, T(st["root_com"]))  # COM of the articulation root body
    lin_w = T(st["root_lin_vel"]) + torch.linalg.cross(ang_w, com_w)
    body_pos, body_vel = ora.phys.body_kinematics(st)
    fwd = mu.quat_apply(quat, torch.tensor([1.0, 0, 0], dtype=torch.float64).repeat(ora.N, 1))
    D = ora.D
    m = desc.model
    soft = torch.stack([T(list(m.soft_lower)[:D]), T(list(m.soft_upper)[:D])], -1).repeat(ora.N, 1, 1)
    data = types.SimpleNamespace(
        root_pos_w=pos, root_quat_w=quat, root_link_pos_w=pos, root_link_quat_w=quat, root_lin_vel_w=lin_w, root_ang_vel_w=ang_w,
        root_lin_vel_b=mu.quat_apply_inverse(quat, lin_w), root_ang_vel_b=mu.quat_apply_inverse(quat, ang_w),
        root_com_lin_vel_b=mu.quat_apply_inverse(quat, lin_w),
        projected_gravity_b=mu.quat_apply_inverse(quat, torch.tensor([0.0, 0, -1.0], dtype=torch.float64).repeat(ora.N, 1)),
        heading_w=torch.atan2(fwd[:, 1], fwd[:, 0]),
        joint_pos=T(st["q"]), joint_vel=T(st["qd"]), default_joint_pos=T(ora.q0).repeat(ora.N, 1),
        default_joint_vel=T(ora.qd0).repeat(ora.N, 1), applied_torque=T(ora.applied_torque), joint_acc=T(ora.joint_acc),
        soft_joint_pos_limits=soft, body_pos_w=T(body_pos), body_link_pos_w=T(body_pos), body_lin_vel_w=T(body_vel))
    asset = types.SimpleNamespace(data=data, joint_names=list(desc.joint_names), body_names=list(desc.body_names), device="cpu")
    asset.find_joints = lambda keys, preserve_order=False: (find_names(keys, asset.joint_names, preserve_order), None)
    asset.find_bodies = lambda keys, preserve_order=False: (find_names(keys, asset.body_names, preserve_order), None)
    sensor = Sensor(ora, list(desc.body_names))

    class Scene(dict):
        pass

    scene = Scene(robot=asset, contact_forces=sensor)
    scene.sensors = {"contact_forces": sensor}
    # `height_scanner_base` (velocity_env_cfg.py:78-85): 3 x 3 yaw-aligned rays, 0.05 m apart, under the root link.  The ray /
    # mesh intersection itself is upstream; the hit heights are the bilinear heightfield at the ray positions.
    gx, gy = torch.meshgrid(torch.tensor([-0.05, 0.0, 0.05], dtype=torch.float64), torch.tensor([-0.05, 0.0, 0.05], dtype=torch.float64), indexing="xy")
    rays = torch.stack([gx.reshape(-1), gy.reshape(-1), torch.zeros(9, dtype=torch.float64)], -1)
    hit = pos[:, None, :] + mu.quat_apply_yaw(quat[:, None, :].repeat(1, 9, 1), rays[None].repeat(ora.N, 1, 1))
    hz, _ = ora.phys.terrain.sample(hit[..., 0].numpy(), hit[..., 1].numpy())
    hit[..., 2] = T(hz)
    scene["height_scanner_base"] = types.SimpleNamespace(data=types.SimpleNamespace(ray_hits_w=hit))
    env = types.SimpleNamespace(
        scene=scene, num_envs=ora.N, device="cpu", step_dt=ora.step_dt, max_episode_length_s=ora.max_episode_length_s,
        command_manager=types.SimpleNamespace(get_command=lambda name: T(ora.vel_command_b)),
        action_manager=types.SimpleNamespace(action=T(ora.action), prev_action=T(ora.prev_action)),
        termination_manager=types.SimpleNamespace(terminated=torch.tensor(ora.terminated)))
    return env


def resolve(params, desc):
    out = {}
    for k, v in (params or {}).items():
        if isinstance(v, SceneEntityCfg):
            v = v.copy()
            if v.joint_names is not None:
                v.joint_ids = find_names(v.joint_names, list(desc.joint_names), v.preserve_order)
            if v.body_names is not None:
                v.body_ids = find_names(v.body_names, list(desc.body_names), v.preserve_order)
        out[k] = v
    return out


def observation_rows(cfg, env, ora, desc):
    """Rows of the two observation groups built the way ObservationManager.compute does [UPSTREAM B2] - terms in the
    DECLARATION ORDER of the reference's ObservationsCfg (VEL/velocity_env_cfg.py:134-254 and the robot's overrides), each
    through its own function (the reference's joint_pos_rel_without_wheel where the cfg names it; the shim's restatement of the
    upstream functions otherwise), then clip, then scale, concatenated.  Noise is left out (it is a random draw: pinned
    separately by the Philox tests); what the fixture pins is order, widths, clip, scale and the reference-owned function."""
    from isaaclab.envs import mdp as up

    R = mu.matrix_from_quat(T(ora.st["root_quat"])) if hasattr(mu, "matrix_from_quat") else None
    # height scanner data for the upstream height_scan(): ray hits of the yaw-aligned grid under the scanner body
    hs = ora.height_scan() + desc.task.scan_offset          # = sensor z - hit z
    sensor_z = T(ora.st["root_pos"][:, 2:3]) if desc.task.scan_body == 0 else None
    out = {}
    for gname in ("policy", "critic"):
        group = getattr(cfg.observations, gname)
        rows, names, widths = [], [], []
        for name, term in vars(group).items():
            if term is None or not hasattr(term, "func"):
                continue
            params = resolve(term.params, desc)
            if term.func is up.height_scan:
                v = T(hs) - 0.5  # height_scan(env, sensor_cfg, offset=0.5) = sensor z - hit z - offset [UPSTREAM B6]
            else:
                v = term.func(env, **params).to(torch.float64)
            if term.clip is not None:
                v = v.clip(term.clip[0], term.clip[1])
            if term.scale is not None:
                v = v * term.scale
            rows.append(v.numpy())
            names.append(name)
            widths.append(v.shape[1])
        out[f"obs_{gname}"] = np.concatenate(rows, -1)
        out[f"obs_{gname}_terms"] = np.array(names)
        out[f"obs_{gname}_widths"] = np.array(widths)
    return out


def snapshot(ora):
    keys = ["root_pos", "root_quat", "root_lin_vel", "root_ang_vel", "q", "qd", "base_com", "root_com"]
    snap = {"st_" + k: ora.st[k] for k in keys}
    for k in ("applied_torque", "joint_acc", "force_hist", "contact_force", "timers", "action", "prev_action", "vel_command_b", "terminated"):
        snap[k] = getattr(ora, k)
    return snap


def main():
    out_dir = os.environ.get("RL_GOLDEN_DIR", os.path.join(ROOT, "tests", "golden"))
    os.makedirs(out_dir, exist_ok=True)
    todo = [a for a in sys.argv[1:]] or ["A1", "Go2", "G1", "A1_HandStand", "Tita", "Go2W"]
    for robot, seed, task in (("A1", 3, None), ("Go2", 4, None), ("G1", 6, None),
                              # wheeled: its own term list (wheeled/unitree_go2w/rough_env_cfg.py:149-171) and the reference-owned
                              # observation function joint_pos_rel_without_wheel (VEL/mdp/observations.py:17-27)
                              ("Go2W", 9, "RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0"),
                              # the hand-stand terms (config/others/unitree_a1_handstand/env/rewards.py) and, on rough terrain,
                              # base_height_l2 with its ray caster, wheel_vel_penalty, feet_distance_y_exp
                              ("A1_HandStand", 7, "RobotLab-Isaac-Velocity-Flat-HandStand-Unitree-A1-v0"),
                              ("Tita", 8, "RobotLab-Isaac-Velocity-Rough-DDTRobot-Tita-v0")):
        if robot not in todo:
            continue
        task = task or f"RobotLab-Isaac-Velocity-Flat-Unitree-{robot}-v0"
        cfg = parse_env_cfg(task, device="cpu")
        desc, spec = compile_cfg(cfg)
        N = 48
        h, to, eo = build_world(desc, load_bundle(task)[1] if "Rough" in task else dict(env_spacing=2.5), N, 0)
        ora = OracleEnv(desc, h, to, N, seed, eo)
        ora.reset()
        rng = np.random.default_rng(seed)
        for s in range(37):  # long enough for contacts, air phases and a command resample of the standing envs
            ora.step(rng.uniform(-1, 1, (N, ora.D)))
        ora.vel_command_b[::5] *= 0.02  # exercise the small-command branches
        if robot in ("A1_HandStand", "Tita"):  # every body just left the ground: the first-air branches (all feet at once is rare)
            ora.timers[::7, :, 0] = ora.step_dt
            ora.timers[::7, :, 1] = 0.0
        snap = snapshot(ora)
        env = duck_env(ora, desc)
        expected = {}
        for name, term in vars(cfg.rewards).items():
            if term is None or not hasattr(term, "func") or term.weight == 0:
                continue
            f = term.func
            params = resolve(term.params, desc)
            if isinstance(f, type):  # class-based term (GaitReward)
                val = f(term, env)(env, **params)
            else:
                val = f(env, **params)
            expected[name] = val.detach().to(torch.float64).numpy()
        # command threshold rule: UniformThresholdVelocityCommand._resample_command (VEL/mdp/commands.py:43-47)
        from robot_lab.tasks.manager_based.locomotion.velocity.mdp.commands import UniformThresholdVelocityCommand

        obj = UniformThresholdVelocityCommand.__new__(UniformThresholdVelocityCommand)
        cmd_in = T(rng.uniform(-1, 1, (N, 3)) * rng.choice([0.1, 1.0], (N, 1)))
        obj.vel_command_b = cmd_in.clone()
        from isaaclab.envs.mdp import UniformVelocityCommand

        orig = UniformVelocityCommand._resample_command
        UniformVelocityCommand._resample_command = lambda self, ids: None  # parent draw is [UPSTREAM]; pin the subclass rule only
        obj._resample_command(torch.arange(N))
        UniformVelocityCommand._resample_command = orig
        # reset_root_state_uniform (VEL/mdp/events.py:205-271) with injected uniform samples
        from robot_lab.tasks.manager_based.locomotion.velocity.mdp import events as ref_events

        samples = [T(rng.uniform(-1, 1, (N, 6))), T(rng.uniform(-0.5, 0.5, (N, 6)))]
        calls = iter(samples)
        ref_events.math_utils.sample_uniform = lambda lo, hi, size, device: next(calls)
        written = {}
        asset = env.scene["robot"]
        droot = np.zeros((N, 13))
        droot[:, :3] = list(desc.model.default_root_pos)
        droot[:, 3:7] = list(desc.model.default_root_quat)
        asset.data.default_root_state = T(droot)
        asset.write_root_pose_to_sim = lambda pose, env_ids=None: written.__setitem__("pose", pose.numpy())
        asset.write_root_velocity_to_sim = lambda vel, env_ids=None: written.__setitem__("vel", vel.numpy())
        env.scene.env_origins = T(ora.env_origins)
        env.scene.terrain = None
        rp = cfg.events.randomize_reset_base.params
        ref_events.reset_root_state_uniform(env, torch.arange(N), rp["pose_range"], rp["velocity_range"])
        obs = observation_rows(cfg, env, ora, desc)
        np.savez_compressed(
            os.path.join(out_dir, f"terms_{robot.lower()}.npz"), seed=seed, N=N, task=task, **obs,
            term_names=np.array(list(expected.keys())), term_values=np.stack(list(expected.values())),
            cmd_in=cmd_in.numpy(), cmd_out=obj.vel_command_b.numpy(),
            reset_pose_samples=samples[0].numpy(), reset_vel_samples=samples[1].numpy(), reset_pose=written["pose"], reset_vel=written["vel"],
            env_origins=ora.env_origins, **snap)
        print(robot, "terms:", list(expected.keys()))


if __name__ == "__main__":
    main()
