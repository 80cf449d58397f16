"""[UPSTREAM isaaclab.envs.mdp] restated from SURVEY.md Appendix B / section 8(a): the one-line term
functions the reference's configs name (`VEL/velocity_env_cfg.py:138-664`) and its own terms call
(`mdp.joint_deviation_l1` in `VEL/mdp/rewards.py:101`), plus the command / action cfg classes the
reference subclasses (`VEL/mdp/commands.py:22,88`).

The MI355X env never calls these in `step()` - its HIP kernels evaluate the term stack; the cfg
compiler only reads `func.__name__`.  They are real torch functions (duck-typed `env`) so that the
reference's terms that call into them run on CPU for golden-vector generation, and so that user
code calling them on `env.unwrapped` keeps working.
"""
from __future__ import annotations

from dataclasses import MISSING

import torch

from isaaclab.managers import ActionTermCfg, CommandTerm, CommandTermCfg, SceneEntityCfg
from isaaclab.utils.configclass import configclass, named_stub
from isaaclab.utils.math import quat_apply_inverse, wrap_to_pi  # noqa: F401

_ROBOT = SceneEntityCfg("robot")


# ---------------------------------------------------------------- observations
def base_lin_vel(env, asset_cfg=_ROBOT):
    return env.scene[asset_cfg.name].data.root_lin_vel_b


def base_ang_vel(env, asset_cfg=_ROBOT):
    return env.scene[asset_cfg.name].data.root_ang_vel_b


def projected_gravity(env, asset_cfg=_ROBOT):
    return env.scene[asset_cfg.name].data.projected_gravity_b


def generated_commands(env, command_name):
    return env.command_manager.get_command(command_name)


def joint_pos_rel(env, asset_cfg=_ROBOT):
    a = env.scene[asset_cfg.name]
    return a.data.joint_pos[:, asset_cfg.joint_ids] - a.data.default_joint_pos[:, asset_cfg.joint_ids]


def joint_vel_rel(env, asset_cfg=_ROBOT):
    a = env.scene[asset_cfg.name]
    return a.data.joint_vel[:, asset_cfg.joint_ids] - a.data.default_joint_vel[:, asset_cfg.joint_ids]


def last_action(env, action_name=None):
    return env.action_manager.action


def height_scan(env, sensor_cfg, offset=0.5):
    sensor = env.scene.sensors[sensor_cfg.name]
    return sensor.data.pos_w[:, 2].unsqueeze(1) - sensor.data.ray_hits_w[..., 2] - offset


# ---------------------------------------------------------------- rewards
def is_terminated(env):
    return env.termination_manager.terminated.float()


def lin_vel_z_l2(env, asset_cfg=_ROBOT):
    return torch.square(env.scene[asset_cfg.name].data.root_lin_vel_b[:, 2])


def ang_vel_xy_l2(env, asset_cfg=_ROBOT):
    return torch.sum(torch.square(env.scene[asset_cfg.name].data.root_ang_vel_b[:, :2]), dim=1)


def flat_orientation_l2(env, asset_cfg=_ROBOT):
    return torch.sum(torch.square(env.scene[asset_cfg.name].data.projected_gravity_b[:, :2]), dim=1)


def joint_torques_l2(env, asset_cfg=_ROBOT):
    return torch.sum(torch.square(env.scene[asset_cfg.name].data.applied_torque[:, asset_cfg.joint_ids]), dim=1)


def joint_vel_l2(env, asset_cfg=_ROBOT):
    return torch.sum(torch.square(env.scene[asset_cfg.name].data.joint_vel[:, asset_cfg.joint_ids]), dim=1)


def joint_acc_l2(env, asset_cfg=_ROBOT):
    return torch.sum(torch.square(env.scene[asset_cfg.name].data.joint_acc[:, asset_cfg.joint_ids]), dim=1)


def joint_deviation_l1(env, asset_cfg=_ROBOT):
    a = env.scene[asset_cfg.name]
    return torch.sum(torch.abs(a.data.joint_pos[:, asset_cfg.joint_ids] - a.data.default_joint_pos[:, asset_cfg.joint_ids]), dim=1)


def joint_pos_limits(env, asset_cfg=_ROBOT):
    a = env.scene[asset_cfg.name]
    q = a.data.joint_pos[:, asset_cfg.joint_ids]
    lim = a.data.soft_joint_pos_limits[:, asset_cfg.joint_ids]
    out = -(q - lim[..., 0]).clip(max=0.0) + (q - lim[..., 1]).clip(min=0.0)
    return torch.sum(out, dim=1)


def action_rate_l2(env):
    return torch.sum(torch.square(env.action_manager.action - env.action_manager.prev_action), dim=1)


def contact_forces(env, threshold, sensor_cfg):
    hist = env.scene.sensors[sensor_cfg.name].data.net_forces_w_history
    violation = torch.max(torch.norm(hist[:, :, sensor_cfg.body_ids], dim=-1), dim=1)[0] - threshold
    return torch.sum(violation.clip(min=0.0), dim=1)


def undesired_contacts(env, threshold, sensor_cfg):
    hist = env.scene.sensors[sensor_cfg.name].data.net_forces_w_history
    is_contact = torch.max(torch.norm(hist[:, :, sensor_cfg.body_ids], dim=-1), dim=1)[0] > threshold
    return torch.sum(is_contact, dim=1)


# ---------------------------------------------------------------- terminations
def time_out(env):
    return env.episode_length_buf >= env.max_episode_length


def illegal_contact(env, threshold, sensor_cfg):
    hist = env.scene.sensors[sensor_cfg.name].data.net_forces_w_history
    return torch.any(torch.max(torch.norm(hist[:, :, sensor_cfg.body_ids], dim=-1), dim=1)[0] > threshold, dim=1)


# ---------------------------------------------------------------- events (evaluated in-kernel; names only)
for _n in (
    "randomize_rigid_body_material", "randomize_rigid_body_mass", "randomize_rigid_body_com",
    "apply_external_force_torque", "reset_joints_by_scale", "reset_joints_by_offset", "randomize_actuator_gains",
    "push_by_setting_velocity", "reset_scene_to_default", "joint_vel_limits", "applied_torque_limits",
    "body_lin_acc_l2", "joint_effort", "root_height_below_minimum", "bad_orientation",
):
    globals()[_n] = named_stub(_n, __name__)


# ---------------------------------------------------------------- actions
@configclass
class JointActionCfg(ActionTermCfg):
    joint_names: list = MISSING
    scale = 1.0
    offset = 0.0
    preserve_order: bool = False


@configclass
class JointPositionActionCfg(JointActionCfg):
    use_default_offset: bool = True


@configclass
class JointVelocityActionCfg(JointActionCfg):
    use_default_offset: bool = True


@configclass
class JointEffortActionCfg(JointActionCfg):
    pass


# ---------------------------------------------------------------- commands [UPSTREAM B7]
class UniformVelocityCommand(CommandTerm):
    def __init__(self, cfg, env):
        super().__init__(cfg, env)
        self.robot = env.scene[cfg.asset_name]
        self.vel_command_b = torch.zeros(self.num_envs, 3, device=self.device)
        self.heading_target = torch.zeros(self.num_envs, device=self.device)
        self.is_heading_env = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        self.is_standing_env = torch.zeros_like(self.is_heading_env)
        self.metrics["error_vel_xy"] = torch.zeros(self.num_envs, device=self.device)
        self.metrics["error_vel_yaw"] = torch.zeros(self.num_envs, device=self.device)

    @property
    def command(self):
        return self.vel_command_b

    def _update_metrics(self):
        max_command_step = self.cfg.resampling_time_range[1] / self._env.step_dt
        self.metrics["error_vel_xy"] += (
            torch.norm(self.vel_command_b[:, :2] - self.robot.data.root_lin_vel_b[:, :2], dim=-1) / max_command_step)
        self.metrics["error_vel_yaw"] += (
            torch.abs(self.vel_command_b[:, 2] - self.robot.data.root_ang_vel_b[:, 2]) / max_command_step)

    def _resample_command(self, env_ids):
        r = torch.empty(len(env_ids), device=self.device)
        self.vel_command_b[env_ids, 0] = r.uniform_(*self.cfg.ranges.lin_vel_x)
        self.vel_command_b[env_ids, 1] = r.uniform_(*self.cfg.ranges.lin_vel_y)
        self.vel_command_b[env_ids, 2] = r.uniform_(*self.cfg.ranges.ang_vel_z)
        if self.cfg.heading_command:
            self.heading_target[env_ids] = r.uniform_(*self.cfg.ranges.heading)
            self.is_heading_env[env_ids] = r.uniform_(0.0, 1.0) <= self.cfg.rel_heading_envs
        self.is_standing_env[env_ids] = r.uniform_(0.0, 1.0) <= self.cfg.rel_standing_envs

    def _update_command(self):
        if self.cfg.heading_command:
            env_ids = self.is_heading_env.nonzero(as_tuple=False).flatten()
            heading_error = wrap_to_pi(self.heading_target[env_ids] - self.robot.data.heading_w[env_ids])
            self.vel_command_b[env_ids, 2] = torch.clip(
                self.cfg.heading_control_stiffness * heading_error,
                min=self.cfg.ranges.ang_vel_z[0], max=self.cfg.ranges.ang_vel_z[1])
        standing_env_ids = self.is_standing_env.nonzero(as_tuple=False).flatten()
        self.vel_command_b[standing_env_ids, :] = 0.0


@configclass
class UniformVelocityCommandCfg(CommandTermCfg):
    class_type: type = UniformVelocityCommand
    asset_name: str = MISSING
    heading_command: bool = False
    heading_control_stiffness: float = 1.0
    rel_standing_envs: float = 0.0
    rel_heading_envs: float = 1.0

    @configclass
    class Ranges:
        lin_vel_x = MISSING
        lin_vel_y = MISSING
        ang_vel_z = MISSING
        heading = None

    ranges = MISSING


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    from isaaclab.utils.configclass import GenericCfg

    val = type(name, (GenericCfg,), {"__module__": __name__}) if name[:1].isupper() else named_stub(name, __name__)
    globals()[name] = val
    return val
