"""[UPSTREAM isaaclab.utils.math] the handful of torch helpers the reference's term functions call
(`VEL/mdp/rewards.py:10,16`, `VEL/mdp/events.py:10`).  Quaternions are (w, x, y, z)."""
import math

import torch


def quat_conjugate(q):
    return torch.cat((q[..., 0:1], -q[..., 1:]), dim=-1)


def quat_mul(q1, q2):
    w1, x1, y1, z1 = q1[..., 0], q1[..., 1], q1[..., 2], q1[..., 3]
    w2, x2, y2, z2 = q2[..., 0], q2[..., 1], q2[..., 2], q2[..., 3]
    return torch.stack(
        [
            w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
            w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
            w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
        ],
        dim=-1,
    )


def quat_apply(quat, vec):
    xyz = quat[..., 1:]
    t = torch.linalg.cross(xyz, vec, dim=-1) * 2
    return vec + quat[..., 0:1] * t + torch.linalg.cross(xyz, t, dim=-1)


def quat_apply_inverse(quat, vec):
    xyz = quat[..., 1:]
    t = torch.linalg.cross(xyz, vec, dim=-1) * 2
    return vec - quat[..., 0:1] * t + torch.linalg.cross(xyz, t, dim=-1)


quat_rotate = quat_apply
quat_rotate_inverse = quat_apply_inverse


def yaw_quat(quat):
    qw, qx, qy, qz = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    yaw = torch.atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz))
    out = torch.zeros_like(quat)
    out[..., 0] = torch.cos(yaw / 2)
    out[..., 3] = torch.sin(yaw / 2)
    return out


def quat_apply_yaw(quat, vec):
    return quat_apply(yaw_quat(quat), vec)


def quat_from_euler_xyz(roll, pitch, yaw):
    cy, sy = torch.cos(yaw * 0.5), torch.sin(yaw * 0.5)
    cr, sr = torch.cos(roll * 0.5), torch.sin(roll * 0.5)
    cp, sp = torch.cos(pitch * 0.5), torch.sin(pitch * 0.5)
    return torch.stack(
        [cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp],
        dim=-1,
    )


def wrap_to_pi(angles):
    wrapped = (angles + math.pi) % (2 * math.pi)
    return torch.where((wrapped == 0) & (angles > 0), torch.full_like(wrapped, math.pi), wrapped - math.pi)


def sample_uniform(lower, upper, size, device):
    if isinstance(size, int):
        size = (size,)
    return torch.rand(*size, device=device) * (upper - lower) + lower


def normalize(x, eps=1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps).unsqueeze(-1)
