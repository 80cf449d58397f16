"""CPU tier: the oracle's BIAS FORCES (Coriolis, centrifugal, gravity) pinned to mechanics, not to the builder's own equations.

`tests/test_urdf_energy.py` pins the joint-space inertia `H(q)` of every compiled model to the reference's URDF files with an
evaluator that shares no code with `model/` or `oracle/`.  What the forward dynamics adds on top of `H` - the bias `b(q, nu)` of
`H nu_dot + b = tau` - was checked only through free fall and a first-order momentum-conservation convergence, and kernel
(articulated-body recursion) and oracle (dense RNEA) are both builder-authored from the same textbook equations.
This test closes that hole with Lagrange's equations:

    d/dt dL/dy' - dL/dy = Q,      L = T - V

* `T(y, y')` and `V(y)` come from the INDEPENDENT evaluator of `tests/test_urdf_energy.py` (raw URDF XML: every link with an
  <inertial>, fixed joints unmerged): kinetic energy of the tree for a base twist and joint rates, potential energy g sum m z.
* Generalised coordinates of the floating base: y = (phi, p, q) - a rotation vector phi about the current orientation
  (R = R0 Exp(phi), body angular velocity omega_b = J_r(phi) phi', J_r the right Jacobian of SO(3)), the WORLD position p of the base
  origin (v_b = R^T p'), the joint angles.  Holonomic coordinates: Lagrange's equations hold as written, no quasi-velocity terms.
* All derivatives of the scalars T, V are finite differences (T is quadratic in y': those are exact; central differences in y).
* `y''` is the ORACLE's: one `Physics.substep` from a contact-free state with limits, velocity limits and armature off is
  exactly nu+ = nu + dt H^-1 (tau - b), so nu_dot = (nu+ - nu) / dt, converted to (phi'', p'', q'') - at phi = 0: phi'' = omega_b_dot,
  p'' = R (a_lin + omega_b x v_b) (the spatial acceleration's linear part is that of the body point passing the frame origin).

The residual of Lagrange's equations with the oracle's accelerations must vanish to 1e-6 of the largest term, for random
configurations, base twists, joint rates and joint torques of A1 and G1.  Needs the reference's URDF files: skipped where absent."""
import os

import numpy as np
import pytest

from oracle import spatial as sp
from oracle.physics import Physics
from robot_lab_amd.desc import arr
from robot_lab_amd.scene import load_bundle
from test_urdf_energy import ROBOTS, Urdf

pytestmark = pytest.mark.skipif(not os.path.isdir(ROBOTS), reason="needs the reference's URDF files")
CASES = [("RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0", "a1_description/urdf/a1.urdf"),
         ("RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0", "g1_description/urdf/g1_29dof_rev_1_0.urdf")]


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _exp(phi):
    t = np.linalg.norm(phi)
    K = _skew(phi)
    if t < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(t) / t * K + (1 - np.cos(t)) / t**2 * (K @ K)


def _right_jacobian(phi):
    t = np.linalg.norm(phi)
    K = _skew(phi)
    if t < 1e-8:
        return np.eye(3) - 0.5 * K
    return np.eye(3) - (1 - np.cos(t)) / t**2 * K + (t - np.sin(t)) / t**3 * (K @ K)


class Lagrangian:
    """T(y, y') and V(y) of the URDF tree in the holonomic coordinates y = (phi, p, q) around the base orientation R0."""

    def __init__(self, urdf, names, R0, g):
        self.U, self.names, self.R0, self.g = urdf, names, R0, g

    def split(self, y):
        return y[:3], y[3:6], {n: y[6 + j] for j, n in enumerate(self.names)}

    def T(self, y, yd):
        phi, _, q = self.split(y)
        phid, pd, qd = self.split(yd)
        R = self.R0 @ _exp(phi)
        wb, vb = _right_jacobian(phi) @ phid, R.T @ pd
        return sum(0.5 * m * vc @ vc + 0.5 * w @ Iw @ w for m, _, Iw, w, vc in self.U.walk(q, qd, wb, vb))

    def V(self, y):
        phi, p, q = self.split(y)
        R = self.R0 @ _exp(phi)
        zero = {n: 0.0 for n in self.names}
        return self.g * sum(m * (p + R @ cw)[2] for m, cw, _, _, _ in self.U.walk(q, zero, np.zeros(3), np.zeros(3)))

    def residual(self, y, yd, ydd, Q, h=1e-4):
        """d/dt dT/dy' - dT/dy + dV/dy - Q, every derivative by finite differences of the scalars; also returns the size of its largest term"""
        n = len(y)
        E = np.eye(n)
        # d/dt dT/dy' = (d2T/dy'dy') y'' + (d2T/dy'dy) y' ;  T quadratic in y': a^T M b = (T(a + b) - T(a - b)) / 2, exact
        Mydd = np.array([(self.T(y, E[i] + ydd) - self.T(y, E[i] - ydd)) / 2 for i in range(n)])
        mom = lambda yy: np.array([(self.T(yy, yd + E[i]) - self.T(yy, yd - E[i])) / 2 for i in range(n)])  # noqa: E731  dT/dy' at (yy, yd)
        dmom = (mom(y + h * yd) - mom(y - h * yd)) / (2 * h)
        dTdy = np.array([(self.T(y + h * E[i], yd) - self.T(y - h * E[i], yd)) / (2 * h) for i in range(n)])
        dVdy = np.array([(self.V(y + h * E[i]) - self.V(y - h * E[i])) / (2 * h) for i in range(n)])
        terms = (Mydd, dmom, dTdy, dVdy, Q)
        return Mydd + dmom - dTdy + dVdy - Q, max(np.abs(t).max() for t in terms)


@pytest.mark.parametrize("task,urdf", CASES)
def test_oracle_accelerations_satisfy_lagranges_equations(task, urdf):
    desc, _ = load_bundle(task)
    m = desc.model
    D, B = m.num_dof, m.num_bodies
    nmax = len(m.joint_lower)
    m.joint_lower[:] = [-1e9] * nmax  # contact-free, limit-free, no reflected rotor inertia (it is not in the URDF's kinetic energy:
    m.joint_upper[:] = [1e9] * nmax   # a joint-local 1/2 a q'^2 the descriptor adds by itself): what is left is H nu_dot + b = tau
    m.joint_vel_limit[:] = [1e9] * nmax
    m.joint_armature[:] = [0.0] * nmax
    m.num_self_pairs = 0              # (G1: random postures put capsule proxies into each other - the penalty force is not a bias force)
    N = 3
    ph = Physics(desc, None, N)  # a plane; the robots start 5 m above it
    rng = np.random.default_rng(11)
    bm = np.tile(arr(m.body_mass, B).astype(float), (N, 1))
    mass, hh, Io = ph.link_inertias(bm, np.zeros((N, 3)))
    lo, hi = np.maximum(arr(desc.model.soft_lower, D).astype(float), -2.5), np.minimum(arr(desc.model.soft_upper, D).astype(float), 2.5)
    lo, hi = np.where(np.isfinite(lo) & (lo > -1e8), lo, -2.5), np.where(np.isfinite(hi) & (hi < 1e8), hi, 2.5)
    st = dict(root_pos=np.tile([0.3, -0.2, 5.0], (N, 1)), root_quat=sp.quat_from_euler_xyz(*rng.uniform(-3, 3, (N, 3)).T), root_lin_vel=rng.uniform(-1, 1, (N, 3)),
              root_ang_vel=rng.uniform(-2, 2, (N, 3)), q=rng.uniform(lo, hi, (N, D)), qd=rng.uniform(-3, 3, (N, D)), link_mass=mass, link_h=hh, link_Io=Io,
              body_mu_s=np.ones((N, B)), body_mu_d=np.ones((N, B)), body_rest=np.zeros((N, B)), ext_force=np.zeros((N, 3)), ext_torque=np.zeros((N, 3)),
              base_com=np.tile(arr(m.body_com, B)[0].astype(float), (N, 1)))
    st["qd"][1] = 0.0             # one env with the joints at rest: the base's own gyroscopic terms
    st["root_ang_vel"][2] = 0.0   # one without base rotation: the joints' Coriolis terms alone
    tau = rng.uniform(-5, 5, (N, D))
    before = {k: v.copy() for k, v in st.items()}
    nu0 = ph.gen_vel(st)
    dt = float(desc.sim.dt)
    ph.substep(st, tau)
    R0 = sp.quat_to_mat(before["root_quat"])
    wb1 = np.einsum("nji,nj->ni", R0, st["root_ang_vel"])                                   # nu+ in the frame of time t (oracle/physics.py: rotated with
    vb1 = np.einsum("nji,nj->ni", R0, st["root_lin_vel"]) - dt * np.cross(nu0[:, :3], nu0[:, 3:6])  # the OLD orientation, reference point shifted by dt w x v)
    nud = (np.concatenate([wb1, vb1, st["qd"]], -1) - nu0) / dt
    U = Urdf(os.path.join(ROBOTS, urdf))
    names = list(desc.joint_names)
    worst = 0.0
    for e in range(N):
        L = Lagrangian(U, names, R0[e], float(desc.sim.gravity))
        wb, vb = nu0[e, :3], nu0[e, 3:6]
        y = np.concatenate([np.zeros(3), before["root_pos"][e], before["q"][e]])
        yd = np.concatenate([wb, R0[e] @ vb, before["qd"][e]])
        ydd = np.concatenate([nud[e, :3], R0[e] @ (nud[e, 3:6] + np.cross(wb, vb)), nud[e, 6:]])
        Q = np.concatenate([np.zeros(6), tau[e]])
        res, scale = L.residual(y, yd, ydd, Q)
        worst = max(worst, np.abs(res).max() / scale)
        assert np.abs(res).max() <= 1e-6 * scale, (task, e, np.abs(res).max(), scale, int(np.abs(res).argmax()))
        # the check has teeth: the same accelerations WITHOUT the velocity-product terms (nu_dot = H^-1 (tau - gravity), i.e. the state at rest)
        # leave a residual of the order of the terms themselves
        if e == 0:
            rest = {k: v.copy() for k, v in before.items()}
            rest["qd"][:] = 0.0
            rest["root_ang_vel"][:] = 0.0
            rest["root_lin_vel"][:] = 0.0
            ph.substep(rest, tau)
            nud_rest = np.concatenate([np.einsum("nji,nj->ni", R0, rest["root_ang_vel"]), np.einsum("nji,nj->ni", R0, rest["root_lin_vel"]), rest["qd"]], -1) / dt
            ydd_w = np.concatenate([nud_rest[e, :3], R0[e] @ (nud_rest[e, 3:6] + np.cross(wb, vb)), nud_rest[e, 6:]])
            res_w, _ = L.residual(y, yd, ydd_w, Q)
            assert np.abs(res_w).max() > 1e-2 * scale
    print(f"\n[lagrange] {task}: worst residual / largest term = {worst:.2e}")
