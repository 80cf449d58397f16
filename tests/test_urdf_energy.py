"""CPU tier: the compiled robot models against the reference's URDF files, by an evaluator that shares NO code with the product
or the oracle (the parity suites compare kernel and oracle on the SAME descriptor, so a wrong URDF -> table compile - an inertia
expressed in the wrong frame, a joint axis taken in the parent frame, a fixed-joint merge that forgets the parallel-axis
term, a joint listed under the wrong name - is invisible to them; VERDICT r1 "common-mode risk").

For random joint positions, joint velocities and base twists the kinetic energy of the robot is computed twice:
  * here, straight from the URDF XML: every <link> with an <inertial> (fixed joints included - nothing is merged), its world
    pose by chaining the <joint> origins (xyz, rpy) and axis rotations, its COM velocity and angular velocity, 1/2 m v^2 + 1/2 w.I w;
  * by the oracle's joint-space inertia matrix H(q) of the committed descriptor bundle (oracle/physics.py: sum K^T I K over
    the merged links), 1/2 nu^T H nu.
Equal energies for arbitrary (q, nu) mean equal mass matrices: masses, COMs, inertia tensors and their frames, joint origins,
orientations, axes, the tree and the joint naming all agree with the file the reference loads (`assets/unitree.py:24,76,126,470`).
Total mass, COM and the whole-robot inertia tensor about the base origin follow as special cases and are checked by name.
Needs /root/reference (the URDFs are not copied into this repository): skipped where it is absent."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from oracle.physics import Physics
from oracle import spatial as sp
from robot_lab_amd.desc import arr
from robot_lab_amd.scene import load_bundle

ROBOTS = "/root/reference/source/robot_lab/data/Robots/unitree"
CASES = [  # task id, URDF the reference's ArticulationCfg names
    ("RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0", "a1_description/urdf/a1.urdf"),
    ("RobotLab-Isaac-Velocity-Flat-Unitree-Go2-v0", "go2_description/urdf/go2_description.urdf"),
    ("RobotLab-Isaac-Velocity-Flat-Unitree-Go2W-v0", "go2w_description/urdf/go2w_description.urdf"),
    ("RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0", "g1_description/urdf/g1_29dof_rev_1_0.urdf"),
]
pytestmark = pytest.mark.skipif(not os.path.isdir(ROBOTS), reason="needs the reference's URDF files")


def _vec(s, default):
    return np.array([float(x) for x in s.split()]) if s else np.array(default, dtype=float)


def _rpy(r, p, y):  # URDF convention: fixed-axis roll, pitch, yaw = Rz(y) Ry(p) Rx(r)
    cr, sr, cp, s_p, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, s_p], [0, 1, 0], [-s_p, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _rodrigues(axis, ang):
    a = axis / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)


class Urdf:
    def __init__(self, path):
        root = ET.parse(path).getroot()
        self.inertial = {}
        for ln in root.findall("link"):
            ine = ln.find("inertial")
            if ine is None:
                continue
            o = ine.find("origin")
            i = ine.find("inertia").attrib
            I = np.array([[float(i["ixx"]), float(i["ixy"]), float(i["ixz"])], [float(i["ixy"]), float(i["iyy"]), float(i["iyz"])],
                          [float(i["ixz"]), float(i["iyz"]), float(i["izz"])]])
            self.inertial[ln.attrib["name"]] = (float(ine.find("mass").attrib["value"]), _vec(o.attrib.get("xyz") if o is not None else None, [0, 0, 0]),
                                                _rpy(*_vec(o.attrib.get("rpy") if o is not None else None, [0, 0, 0])), I)
        self.children = {}
        childs = set()
        for j in root.findall("joint"):
            o = j.find("origin")
            ax = j.find("axis")
            rec = dict(name=j.attrib["name"], type=j.attrib["type"], child=j.find("child").attrib["link"],
                       xyz=_vec(o.attrib.get("xyz") if o is not None else None, [0, 0, 0]), R0=_rpy(*_vec(o.attrib.get("rpy") if o is not None else None, [0, 0, 0])),
                       axis=_vec(ax.attrib.get("xyz") if ax is not None else None, [1, 0, 0]))
            self.children.setdefault(j.find("parent").attrib["link"], []).append(rec)
            childs.add(rec["child"])
        self.root = [ln.attrib["name"] for ln in root.findall("link") if ln.attrib["name"] not in childs][0]

    def walk(self, q, qd, w0, v0):
        """Yields (mass, world COM, world inertia about the COM, angular velocity, COM velocity) per link with an inertial; base
        frame = world frame, base twist (w0, v0 of the base origin)."""
        stack = [(self.root, np.eye(3), np.zeros(3), np.asarray(w0, float), np.asarray(v0, float))]
        while stack:
            link, R, p, w, v = stack.pop()
            if link in self.inertial:
                m, c, Ri, I = self.inertial[link]
                cw = p + R @ c
                Rw = R @ Ri
                yield m, cw, Rw @ I @ Rw.T, w, v + np.cross(w, cw - p)
            for j in self.children.get(link, []):
                moving = j["type"] in ("revolute", "continuous")
                assert moving or j["type"] == "fixed", j["type"]
                Rc = R @ j["R0"] @ (_rodrigues(j["axis"], q.get(j["name"], 0.0)) if moving else np.eye(3))
                pc = p + R @ j["xyz"]
                wc = w + (Rc @ (j["axis"] / np.linalg.norm(j["axis"])) * qd.get(j["name"], 0.0) if moving else 0.0)
                stack.append((j["child"], Rc, pc, wc, v + np.cross(w, pc - p)))


@pytest.mark.parametrize("task,urdf", CASES)
def test_kinetic_energy_matches_the_urdf(task, urdf):
    _check(task, os.path.join(ROBOTS, urdf))


def test_every_committed_robot_matches_its_urdf():
    """The same for every robot with a committed bundle; the URDF path comes from the reference's own ArticulationCfg
    (`cfg.scene.robot.spawn.asset_path`, parsed through the shims)."""
    import glob

    from robot_lab_amd import shims

    shims.install(shims.REFERENCE_SOURCE)
    import robot_lab.tasks  # noqa: F401
    from isaaclab_tasks.utils import parse_env_cfg

    from robot_lab_amd.scene import DATA_DIR

    seen = 0
    for path in sorted(glob.glob(os.path.join(DATA_DIR, "RobotLab-Isaac-Velocity-Flat-*.json"))):
        task = os.path.basename(path)[:-5]
        urdf = parse_env_cfg(task, device="cpu", num_envs=4).scene.robot.spawn.asset_path
        assert urdf.endswith(".urdf") and os.path.isfile(urdf), (task, urdf)
        _check(task, urdf)
        seen += 1
    assert seen >= 18


def _check(task, urdf_path):
    desc, _ = load_bundle(task)
    m = desc.model
    D = m.num_dof
    U = Urdf(urdf_path)
    n = 12
    t = desc.terrain  # (Flat ids: mostly a plane; the model is the Rough id's)
    ph = Physics(desc, None if t.is_plane else np.zeros(t.nx * t.ny), n)
    mass, h, Io = ph.link_inertias(np.broadcast_to(arr(m.body_mass, m.num_bodies).astype(np.float64), (n, m.num_bodies)).copy(), np.zeros((n, 3)))
    rng = np.random.default_rng(7)
    lo, hi = arr(m.joint_lower, D).astype(np.float64), arr(m.joint_upper, D).astype(np.float64)
    lo, hi = np.maximum(lo, -3.0), np.minimum(hi, 3.0)  # continuous joints (wheels) carry +-inf-like limits
    q = rng.uniform(lo, hi, (n, D))
    q[0] = 0.0
    nu = rng.uniform(-2, 2, (n, 6 + D))
    nu[1, 6:] = 0.0      # rigid-body motion only: whole-robot mass / COM / inertia
    nu[2, :6] = 0.0      # fixed base: the joint block of H
    quat = np.tile([1.0, 0, 0, 0], (n, 1))
    Rw, ow, X = ph.kinematics(np.zeros((n, 3)), quat, q)
    K = ph.link_jacobians(X)
    I = np.stack([sp.spatial_inertia(mass[:, i], h[:, i], Io[:, i]) for i in range(ph.L)], 1)
    H = np.einsum("nlij,nljk,nlkm->nim", np.swapaxes(K, 2, 3), I, K)
    T_model = 0.5 * np.einsum("ni,nij,nj->n", nu, H, nu)
    T_urdf = np.zeros(n)
    for e in range(n):
        qe = {name: q[e, j] for j, name in enumerate(desc.joint_names)}
        qde = {name: nu[e, 6 + j] for j, name in enumerate(desc.joint_names)}
        for mk, cw, Iw, w, vc in U.walk(qe, qde, nu[e, :3], nu[e, 3:6]):
            T_urdf[e] += 0.5 * mk * vc @ vc + 0.5 * w @ Iw @ w
    moving = {j["name"] for js in U.children.values() for j in js if j["type"] != "fixed"}
    assert moving == set(desc.joint_names), (moving ^ set(desc.joint_names))  # every moving joint of the file is a task joint, by name
    np.testing.assert_allclose(T_model, T_urdf, rtol=2e-6)  # the bundle stores fp32 numbers
    # by name: total mass, COM, whole-robot inertia about the base origin at the zero pose
    links = list(U.walk({}, {}, np.zeros(3), np.zeros(3)))
    M = sum(x[0] for x in links)
    com = sum(x[0] * x[1] for x in links) / M
    Io_u = sum(x[2] + x[0] * (x[1] @ x[1] * np.eye(3) - np.outer(x[1], x[1])) for x in links)
    H0 = H[0]
    np.testing.assert_allclose(H0[3, 3], M, rtol=1e-6)
    np.testing.assert_allclose(np.array([H0[2, 4], H0[0, 5], H0[1, 3]]) / M, com, atol=2e-7)  # H[w, v] block = [m c]x
    np.testing.assert_allclose(H0[:3, :3], Io_u, rtol=2e-6, atol=3e-7 * np.abs(Io_u).max())  # fp32 storage of the summands
