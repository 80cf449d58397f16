"""TEST INFRASTRUCTURE ONLY (oracle).  Batched fp64 spatial-algebra helpers (numpy).

Conventions: spatial vectors are [angular(3); linear(3)]; quaternions are (w, x, y, z) as in the
reference (`SURVEY.md` B3; call sites `VEL/mdp/rewards.py:30,34`).  Everything is batched over a
leading env axis N.
"""
import numpy as np


def skew(v):
    z = np.zeros(v.shape[:-1])
    return np.stack(
        [
            np.stack([z, -v[..., 2], v[..., 1]], -1),
            np.stack([v[..., 2], z, -v[..., 0]], -1),
            np.stack([-v[..., 1], v[..., 0], z], -1),
        ],
        -2,
    )


def axis_angle_mat(axis, ang):
    """Rotation matrix R(axis, ang) (maps child coords -> parent coords); axis [3], ang [N]."""
    a = np.asarray(axis, dtype=np.float64)
    K = skew(a)
    s = np.sin(ang)[:, None, None]
    c = np.cos(ang)[:, None, None]
    return np.eye(3) + s * K + (1.0 - c) * (K @ K)


def quat_to_mat(q):
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return np.stack(
        [
            np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
            np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
            np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1),
        ],
        -2,
    )


def quat_mul(a, b):
    aw, ax, ay, az = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bw, bx, by, bz = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack(
        [
            aw * bw - ax * bx - ay * by - az * bz,
            aw * bx + ax * bw + ay * bz - az * by,
            aw * by - ax * bz + ay * bw + az * bx,
            aw * bz + ax * by - ay * bx + az * bw,
        ],
        -1,
    )


def quat_from_euler_xyz(roll, pitch, yaw):
    cy, sy = np.cos(yaw * 0.5), np.sin(yaw * 0.5)
    cr, sr = np.cos(roll * 0.5), np.sin(roll * 0.5)
    cp, sp = np.cos(pitch * 0.5), np.sin(pitch * 0.5)
    return np.stack(
        [
            cy * cr * cp + sy * sr * sp,
            cy * sr * cp - sy * cr * sp,
            cy * cr * sp + sy * sr * cp,
            sy * cr * cp - cy * sr * sp,
        ],
        -1,
    )


def xform_motion(E, r):
    """6x6 motion transform parent->child coords for child frame at r (parent coords), E = R^T."""
    N = E.shape[0]
    X = np.zeros((N, 6, 6))
    X[:, :3, :3] = E
    X[:, 3:, 3:] = E
    X[:, 3:, :3] = -E @ skew(r)
    return X


def crm(v):
    """Spatial motion cross-product matrix."""
    N = v.shape[0]
    out = np.zeros((N, 6, 6))
    w, l = skew(v[:, :3]), skew(v[:, 3:])
    out[:, :3, :3] = w
    out[:, 3:, :3] = l
    out[:, 3:, 3:] = w
    return out


def crf(v):
    return -np.swapaxes(crm(v), 1, 2)


def spatial_inertia(mass, h, Io):
    """mass [N], h = m*c [N,3], Io = inertia about the frame origin [N,3,3] -> [N,6,6]."""
    N = mass.shape[0]
    I = np.zeros((N, 6, 6))
    hx = skew(h)
    I[:, :3, :3] = Io
    I[:, :3, 3:] = hx
    I[:, 3:, :3] = -hx
    I[:, 3:, 3:] = mass[:, None, None] * np.eye(3)
    return I
