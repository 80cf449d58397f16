"""TEST INFRASTRUCTURE ONLY (oracle).  fp64 numpy restatement of the articulated-body simulator.

PARITY UNPINNED for the physics: the reference delegates rigid-body dynamics and contact to
PhysX 5 (closed source, not in /root/reference), so nothing in the reference
pins these numbers.  This file *defines* the simulator the north-star asks for and is the checker
the HIP kernels are compared against.  It is deliberately a different formulation from the kernel:

    kernel : star-topology CRBA in base coordinates, block-arrow Schur complement per leg lane, fp32
    oracle : generic tree, per-link 6x(6+D) Jacobians K_i, H = sum K_i^T I_i K_i, dense solve, fp64

Model (one physics substep of length dt; generalized velocity nu = [omega_b, v_b, qd] with the base
twist in base coordinates at the base-link origin):

    (H + A) nu+ = H nu + dt (tau - b) + r                      semi-implicit (symplectic) Euler
    q+ = q + dt qd+ ;  quat+ = normalize(quat * [1, dt/2 omega_b+]) ;  v_w+ = R (v_b+ + dt omega_b x v_b) ;  p+ = p + dt v_w+

    b   : Coriolis/centrifugal/gravity bias  (RNEA with qdd = 0, base acceleration = -g)
    A,r : linearly-implicit contact / joint-limit / implicit-PD terms (all PSD, so the step is
          stable for any stiffness - needed because robots spawn upside-down and in penetration,
          .../unitree_a1/rough_env_cfg.py:56-73)

Contact: collision spheres against a bilinear heightfield.  For sphere g (radius r) on link i with
centre c_w: phi = r - (c_w.z - h(x,y)) n_z, n = normalised heightfield normal, contact point
x_c = c_w - r n, point velocity u = J nu.  With d_n = k dt + c min(1, phi/phi_ref)(1 - e),
bias = min(k phi, v_dep d_n) (depenetration-velocity cap, unitree.py:33), lagged normal force
fn0 = bias - d_n u_n, the contact is active iff phi > 0 and fn0 > 0, and applies

    F = n bias - D (J nu+),   D = d_t I + (d_n - d_t) n n^T,   d_t = min(c_t, mu fn0 / |u_t|)

(lagged regularised Coulomb friction; mu = mu_s below v_stick else mu_d).
"""
import numpy as np

from robot_lab_amd.desc import arr

from . import spatial as sp


class TerrainSampler:
    def __init__(self, tdesc, heights):
        self.t = tdesc
        self.h = None if tdesc.is_plane else np.asarray(heights, dtype=np.float64).reshape(tdesc.nx, tdesc.ny)

    def sample(self, x, y):
        """height and unit normal of the bilinear heightfield at world (x, y); arrays of any shape."""
        if self.t.is_plane:
            z = np.zeros_like(x)
            n = np.zeros(x.shape + (3,))
            n[..., 2] = 1.0
            return z, n
        t = self.t
        gx = (x - t.x0) / t.hscale
        gy = (y - t.y0) / t.hscale
        ix = np.clip(np.floor(gx), 0, t.nx - 2).astype(np.int64)
        iy = np.clip(np.floor(gy), 0, t.ny - 2).astype(np.int64)
        fx = np.clip(gx - ix, 0.0, 1.0)
        fy = np.clip(gy - iy, 0.0, 1.0)
        h00, h10 = self.h[ix, iy], self.h[ix + 1, iy]
        h01, h11 = self.h[ix, iy + 1], self.h[ix + 1, iy + 1]
        hx0 = h00 + fx * (h10 - h00)
        hx1 = h01 + fx * (h11 - h01)
        z = hx0 + fy * (hx1 - hx0)
        dzdx = ((1 - fy) * (h10 - h00) + fy * (h11 - h01)) / t.hscale
        dzdy = ((1 - fx) * (h01 - h00) + fx * (h11 - h10)) / t.hscale
        inv = 1.0 / np.sqrt(dzdx * dzdx + dzdy * dzdy + 1.0)
        n = np.stack([-dzdx * inv, -dzdy * inv, inv], -1)
        return z, n


def segment_closest(a0, a1, b0, b1):
    """Closest points of the segments a0-a1 and b0-b1, batched over the leading axis (Ericson, Real-Time Collision Detection 5.1.9;
    the same branch order as csrc/env_step.h `segment_closest`, so that both sides pick the same points in the degenerate cases)."""
    d1, d2, r = a1 - a0, b1 - b0, a0 - b0
    a = np.einsum("ni,ni->n", d1, d1)
    e = np.einsum("ni,ni->n", d2, d2)
    f = np.einsum("ni,ni->n", d2, r)
    c = np.einsum("ni,ni->n", d1, r)
    b = np.einsum("ni,ni->n", d1, d2)
    eps = 1e-12
    den = a * e - b * b
    s = np.where(den > eps, np.clip((b * f - c * e) / np.maximum(den, eps), 0.0, 1.0), 0.0)
    s = np.where(a > eps, s, 0.0)
    t = np.where(e > eps, (b * s + f) / np.maximum(e, eps), 0.0)
    # t outside [0, 1]: clamp it and recompute s for the clamped t
    s_lo = np.where(a > eps, np.clip(-c / np.maximum(a, eps), 0.0, 1.0), 0.0)
    s_hi = np.where(a > eps, np.clip((b - c) / np.maximum(a, eps), 0.0, 1.0), 0.0)
    s = np.where(e > eps, np.where(t < 0.0, s_lo, np.where(t > 1.0, s_hi, s)), s_lo)  # (b degenerate: the point of a nearest to b0)
    t = np.clip(t, 0.0, 1.0)
    return a0 + s[:, None] * d1, b0 + t[:, None] * d2


class Physics:
    def __init__(self, desc, terrain_heights, num_envs):
        self.desc = desc
        m = desc.model
        self.N = num_envs
        self.L, self.D, self.B, self.G = m.num_links, m.num_dof, m.num_bodies, m.num_spheres
        L, D, B, G = self.L, self.D, self.B, self.G
        self.parent = arr(m.link_parent, L).astype(int)
        # link index = 1 + task joint index, which need not be topological (MagicLab Z1 lists hip roll before the hip
        # pitch it hangs from): walk the tree parents first
        self.order, done = [], {0}
        while len(self.order) < L - 1:
            for i in range(1, L):
                if i not in done and self.parent[i] in done:
                    self.order.append(i)
                    done.add(i)
        self.origin = arr(m.link_origin, L).astype(np.float64)
        self.axis = arr(m.link_axis, L).astype(np.float64)
        # unit axes (fp32 components of a tilted axis are off by ~2e-9: FFTAI GR1), as csrc/rl_env_host.h does; link 0 and fixed links: zero
        an = np.linalg.norm(self.axis, axis=1, keepdims=True)
        self.axis = np.where(an > 0, self.axis / np.where(an > 0, an, 1.0), 0.0)
        lq = arr(m.link_quat, L).astype(np.float64)
        lq[np.abs(lq).sum(1) == 0] = [1.0, 0.0, 0.0, 0.0]  # descriptors written before link_quat existed
        # unit quaternions: the descriptor stores fp32 components (norm off by ~3e-8), and quat_to_mat is a rotation only for a unit
        # quaternion.  csrc/rl_env_host.h quat_to_rows normalises too; found by the fp64 lane program (tests/test_fp64_lane_program.py), which
        # differed from this oracle by 1e-8 on the robots with rotated joint frames (G1, GR1) and by 1e-12 on the others
        lq /= np.linalg.norm(lq, axis=1, keepdims=True)
        self.rot0 = sp.quat_to_mat(lq)  # joint frame axes in the parent link frame (URDF joint rpy)
        self.wrench_link = int(m.body_link[desc.task.base_body])  # link carrying the body the wrench event addresses
        self.body_link = arr(m.body_link, B).astype(int)
        self.sphere_body = arr(m.sphere_body, G).astype(int)
        self.sphere_link = self.body_link[self.sphere_body]
        self.sphere_center = arr(m.sphere_center, G).astype(np.float64)
        self.sphere_radius = arr(m.sphere_radius, G).astype(np.float64)
        self.terrain = TerrainSampler(desc.terrain, terrain_heights)
        self.S = np.zeros((L, 6))
        self.S[:, :3] = self.axis
        self.S[0] = 0.0
        # Optional record of how close each env came to one of the model's DISCONTINUITIES during the substeps run since
        # it was set (tests/helpers.py switch_mask): dict of per-env minima, filled by substep() when not None.
        #   phi    [m]   |penetration| of the sphere nearest to touching / leaving (contact on-off: the damping force jumps)
        #   fn0    [N]   |lagged normal force| of a penetrating sphere nearest to its activation threshold 0
        #   stick  [m/s] ||u_t| - v_stick| of an active contact (mu_s <-> mu_d)
        #   limit  [rad] distance of a joint to a limit it is about to cross (the limit damper switches on)
        #   cell   0 / inf: a touching sphere sits within 2e-5 m (fp32 resolution of world coordinates at +-60 m) of a heightfield
        #          cell edge across which the bilinear patch's NORMAL jumps (the height is continuous there, its gradient is not)
        self.margins = None
        # np.float32: the linear solve of every substep is done in single precision (everything else stays fp64).  Used by the
        # "twins" of tests/helpers.py teacher_forced_check to measure how much an fp32 solve of THIS env's system moves the
        # result (conditioning of H + A: light distal links next to a heavy trunk, stiff contacts) - never by the checker itself.
        self.solve_dtype = None

    cell_probe = 2e-6  # [m] a touching sphere this close to a heightfield cell edge counts as sitting on it (margins["cell"])

    def _margin(self, key, val):
        m = self.margins
        m[key] = np.minimum(m[key], val) if key in m else np.asarray(val, dtype=np.float64).copy()

    # ------------------------------------------------------------------ per-env inertial tables
    def link_inertias(self, body_mass, base_com_shift):
        """body_mass [N,B] (randomised), base_com_shift [N,3] -> mass [N,L], h [N,L,3], Io [N,L,3,3].

        Inertia scales with mass (`recompute_inertia=True`, velocity_env_cfg.py:281,292)."""
        m = self.desc.model
        N, L, B = self.N, self.L, self.B
        m0 = arr(m.body_mass, B).astype(np.float64)
        com = np.broadcast_to(arr(m.body_com, B).astype(np.float64), (N, B, 3)).copy()
        com[:, self.desc.task.base_body] += base_com_shift
        I6 = arr(m.body_inertia, B).astype(np.float64)
        I0 = np.zeros((B, 3, 3))
        I0[:, 0, 0], I0[:, 1, 1], I0[:, 2, 2] = I6[:, 0], I6[:, 1], I6[:, 2]
        I0[:, 0, 1] = I0[:, 1, 0] = I6[:, 3]
        I0[:, 0, 2] = I0[:, 2, 0] = I6[:, 4]
        I0[:, 1, 2] = I0[:, 2, 1] = I6[:, 5]
        scale = np.where(m0 > 0, body_mass / np.where(m0 > 0, m0, 1.0), 0.0)
        Icom = scale[:, :, None, None] * I0[None]
        cx = sp.skew(com)
        Io_b = Icom - body_mass[:, :, None, None] * (cx @ cx)  # parallel axis to link origin
        mass = np.zeros((N, L))
        h = np.zeros((N, L, 3))
        Io = np.zeros((N, L, 3, 3))
        for b in range(B):
            l = self.body_link[b]
            mass[:, l] += body_mass[:, b]
            h[:, l] += body_mass[:, b, None] * com[:, b]
            Io[:, l] += Io_b[:, b]
        return mass, h, Io

    # ------------------------------------------------------------------ kinematics
    def kinematics(self, root_pos, root_quat, q):
        """World pose of every link and the parent->child motion transforms."""
        N, L = self.N, self.L
        Rw = np.zeros((N, L, 3, 3))
        ow = np.zeros((N, L, 3))
        X = np.zeros((N, L, 6, 6))
        Rw[:, 0] = sp.quat_to_mat(root_quat)
        ow[:, 0] = root_pos
        for i in self.order:
            p = self.parent[i]
            Rj = self.rot0[i][None] @ sp.axis_angle_mat(self.axis[i], q[:, i - 1])  # child -> parent
            E = np.swapaxes(Rj, 1, 2)
            X[:, i] = sp.xform_motion(E, np.broadcast_to(self.origin[i], (N, 3)))
            Rw[:, i] = Rw[:, p] @ Rj
            ow[:, i] = ow[:, p] + np.einsum("nij,j->ni", Rw[:, p], self.origin[i])
        return Rw, ow, X

    def link_jacobians(self, X):
        N, L, D = self.N, self.L, self.D
        K = np.zeros((N, L, 6, 6 + D))
        K[:, 0, :, :6] = np.eye(6)
        for i in self.order:
            K[:, i] = X[:, i] @ K[:, self.parent[i]]
            K[:, i, :, 6 + i - 1] += self.S[i]
        return K

    def body_kinematics(self, st):
        """World position / linear velocity of every body frame origin (for foot terms, rewards.py:536-540)."""
        m = self.desc.model
        Rw, ow, X = self.kinematics(st["root_pos"], st["root_quat"], st["q"])
        K = self.link_jacobians(X)
        nu = self.gen_vel(st)
        bp = arr(m.body_pos, self.B).astype(np.float64)
        pos = np.zeros((self.N, self.B, 3))
        vel = np.zeros((self.N, self.B, 3))
        for b in range(self.B):
            l = self.body_link[b]
            pos[:, b] = ow[:, l] + np.einsum("nij,j->ni", Rw[:, l], bp[b])
            v = np.einsum("nij,nj->ni", K[:, l], nu)
            vel[:, b] = np.einsum("nij,nj->ni", Rw[:, l], v[:, 3:] + np.cross(v[:, :3], bp[b]))
        return pos, vel

    @staticmethod
    def gen_vel(st):
        R = sp.quat_to_mat(st["root_quat"])
        wb = np.einsum("nji,nj->ni", R, st["root_ang_vel"])
        vb = np.einsum("nji,nj->ni", R, st["root_lin_vel"])
        return np.concatenate([wb, vb, st["qd"]], -1)

    # ------------------------------------------------------------------ one substep
    def substep(self, st, tau_explicit, pd=None):
        """Advance ``st`` (dict of arrays, modified in place) by sim.dt.

        st keys: root_pos[N,3] root_quat[N,4] root_lin_vel[N,3] (world, link origin) root_ang_vel[N,3]
                 q[N,D] qd[N,D] link_mass/link_h/link_Io, body_mu_s/body_mu_d/body_rest [N,B],
                 ext_force[N,3] ext_torque[N,3] (base-body frame, applied at the base-body COM), base_com[N,3]
        tau_explicit [N,D]: explicit joint torques (DC motors).
        pd: optional dict(kp, kd, q_tgt, qd_tgt, mask[N,D] bool) for implicitly integrated PD joints.
        Returns dict(contact_force[N,B,3], joint_acc[N,D]).
        """
        sim = self.desc.sim
        dt, g = float(sim.dt), float(sim.gravity)
        N, L, D, G = self.N, self.L, self.D, self.G
        ND = 6 + D
        Rw, ow, X = self.kinematics(st["root_pos"], st["root_quat"], st["q"])
        K = self.link_jacobians(X)
        nu = self.gen_vel(st)
        I = np.stack([sp.spatial_inertia(st["link_mass"][:, i], st["link_h"][:, i], st["link_Io"][:, i]) for i in range(L)], 1)
        # joint-space inertia and bias via the link Jacobians
        H = np.einsum("nlij,nljk,nlkm->nim", np.swapaxes(K, 2, 3), I, K)
        v = np.einsum("nlij,nj->nli", K, nu)
        a = np.zeros((N, L, 6))
        a[:, 0, 3:] = np.einsum("nji,j->ni", Rw[:, 0], np.array([0.0, 0.0, g]))
        for i in self.order:
            vj = self.S[i][None] * st["qd"][:, i - 1, None]
            a[:, i] = np.einsum("nij,nj->ni", X[:, i], a[:, self.parent[i]]) + np.einsum("nij,nj->ni", sp.crm(v[:, i]), vj)
        f = np.einsum("nlij,nlj->nli", I, a)
        for i in range(L):
            f[:, i] += np.einsum("nij,nj->ni", sp.crf(v[:, i]), np.einsum("nij,nj->ni", I[:, i], v[:, i]))
        # persistent external wrench on the base body (G1: the torso), body frame, at its COM [UPSTREAM B8]
        wl = self.wrench_link
        f[:, wl, :3] -= st["ext_torque"] + np.cross(st["base_com"], st["ext_force"])
        f[:, wl, 3:] -= st["ext_force"]
        b = np.einsum("nlji,nlj->ni", K, f)

        A = np.zeros((N, ND, ND))
        r = np.zeros((N, ND))
        idx = np.arange(D)
        arm = arr(self.desc.model.joint_armature, D).astype(np.float64)
        H[:, 6 + idx, 6 + idx] += arm
        # joint limits (hard limits in the reference: a1.urdf:369,411,439) as implicit spring-dampers
        lo = arr(self.desc.model.joint_lower, D).astype(np.float64)
        hi = arr(self.desc.model.joint_upper, D).astype(np.float64)
        kl, cl = float(sim.limit_k), float(sim.limit_c)
        below, above = st["q"] < lo, st["q"] > hi
        if self.margins is not None:
            self._margin("limit", np.minimum(np.abs(st["q"] - lo), np.abs(st["q"] - hi)).min(axis=1))
        viol = np.where(below, lo - st["q"], 0.0) - np.where(above, st["q"] - hi, 0.0)
        act = below | above
        A[:, 6 + idx, 6 + idx] += np.where(act, dt * (kl * dt + cl), 0.0)
        r[:, 6:] += dt * kl * viol
        tau = tau_explicit.copy()
        if pd is not None:
            e = pd["q_tgt"] - st["q"]
            ed = pd["qd_tgt"] - st["qd"]
            A[:, 6 + idx, 6 + idx] += np.where(pd["mask"], dt * (pd["kd"] + pd["kp"] * dt), 0.0)
            r[:, 6:] += np.where(pd["mask"], dt * (pd["kp"] * e + pd["kd"] * pd["qd_tgt"]), 0.0)
        # contacts
        k, c, phi_ref = float(sim.contact_k), float(sim.contact_c), float(sim.contact_phi_ref)
        ct, vdep, vstick = float(sim.contact_ct), float(sim.contact_vdep), float(sim.contact_vstick)
        cw = np.zeros((N, G, 3))
        for gi in range(G):
            l = self.sphere_link[gi]
            cw[:, gi] = ow[:, l] + np.einsum("nij,j->ni", Rw[:, l], self.sphere_center[gi])
        hz, nrm = self.terrain.sample(cw[..., 0], cw[..., 1])
        phi = self.sphere_radius[None] - (cw[..., 2] - hz) * nrm[..., 2]
        if self.margins is not None and G > 0:
            self._margin("phi", np.abs(phi).min(axis=1))
            if not self.desc.terrain.is_plane:
                near = phi > -1e-4  # touching, or about to
                jump = np.zeros_like(phi)
                pr = self.cell_probe
                for dx, dy in ((pr, 0.0), (-pr, 0.0), (0.0, pr), (0.0, -pr)):
                    _, n2 = self.terrain.sample(cw[..., 0] + dx, cw[..., 1] + dy)
                    jump = np.maximum(jump, np.abs(n2 - nrm).max(axis=-1))
                self._margin("cell", np.where(near & (jump > 1e-3), 0.0, np.inf).min(axis=1))
        contacts = []
        for gi in range(G):
            if not np.any(phi[:, gi] > 0):
                continue
            l, bdy = self.sphere_link[gi], self.sphere_body[gi]
            n = nrm[:, gi]
            ploc = self.sphere_center[gi][None] - self.sphere_radius[gi] * np.einsum("nji,nj->ni", Rw[:, l], n)
            P = np.zeros((N, 3, 6))
            P[:, :, :3] = -sp.skew(ploc)
            P[:, :, 3:] = np.eye(3)
            J = Rw[:, l] @ P @ K[:, l]
            u = np.einsum("nij,nj->ni", J, nu)
            un = np.einsum("ni,ni->n", n, u)
            ut = u - un[:, None] * n
            utn = np.linalg.norm(ut, axis=-1)
            ph = phi[:, gi]
            cn = c * np.minimum(1.0, np.maximum(ph, 0.0) / phi_ref) * (1.0 - st["body_rest"][:, bdy])
            dn = cn + k * dt
            bias = np.minimum(k * ph, vdep * dn)
            fn0 = bias - dn * un
            active = (ph > 0) & (fn0 > 0)
            mu = np.where(utn < vstick, st["body_mu_s"][:, bdy], st["body_mu_d"][:, bdy])
            dtan = np.minimum(ct, mu * np.maximum(fn0, 0.0) / np.maximum(utn, 1e-6))
            if self.margins is not None:
                self._margin("fn0", np.where(ph > 0, np.abs(fn0), np.inf))
                # the friction coefficient only matters where the Coulomb bound (not the stick damping c_t) is the smaller one
                coulomb = active & (np.maximum(st["body_mu_s"][:, bdy], st["body_mu_d"][:, bdy]) * np.maximum(fn0, 0.0) / np.maximum(utn, 1e-6) < ct)
                self._margin("stick", np.where(coulomb, np.abs(utn - vstick), np.inf))
            Dm = dtan[:, None, None] * np.eye(3) + (dn - dtan)[:, None, None] * (n[:, :, None] * n[:, None, :])
            w = active.astype(np.float64)
            A += dt * w[:, None, None] * (np.swapaxes(J, 1, 2) @ Dm @ J)
            r += dt * (w * bias)[:, None] * np.einsum("nij,ni->nj", J, n)
            contacts.append((bdy, J, n, bias, Dm, w))
        # self-collision (include/rl_env.h rl_model_desc.self_pair; the reference: enabled_self_collisions, assets/unitree.py:482):
        # capsule pairs repel with an EXPLICIT penalty force k * penetration along the line between the segments' closest points,
        # evaluated at the positions of the start of the substep (no damping, no friction, not part of the contact sensor)
        m = self.desc.model
        if int(m.num_self_pairs) > 0:
            ks = float(sim.self_k)
            cw0, cw1, crad, clink = [], [], [], []
            for c in range(int(m.num_capsules)):
                l = int(m.capsule_link[c])
                cw0.append(ow[:, l] + np.einsum("nij,j->ni", Rw[:, l], np.array(m.capsule_p0[c][:], dtype=np.float64)))
                cw1.append(ow[:, l] + np.einsum("nij,j->ni", Rw[:, l], np.array(m.capsule_p1[c][:], dtype=np.float64)))
                crad.append(float(m.capsule_radius[c]))
                clink.append(l)
            for pi in range(int(m.num_self_pairs)):
                ca, cb = int(m.self_pair[pi][0]), int(m.self_pair[pi][1])
                xa, xb = segment_closest(cw0[ca], cw1[ca], cw0[cb], cw1[cb])
                dvec = xa - xb
                dist = np.linalg.norm(dvec, axis=-1)
                pen = crad[ca] + crad[cb] - dist
                hit = pen > 0.0  # (no switch margin: the force is continuous at the onset of the contact)
                if not np.any(hit):
                    continue
                nrm_ab = dvec / np.maximum(dist, 1e-9)[:, None]
                F = (ks * np.where(hit, pen, 0.0))[:, None] * nrm_ab  # on a; -F on b
                for l, x, sgn in ((clink[ca], xa, 1.0), (clink[cb], xb, -1.0)):
                    ploc = np.einsum("nji,nj->ni", Rw[:, l], x - ow[:, l])
                    P = np.zeros((N, 3, 6))
                    P[:, :, :3] = -sp.skew(ploc)
                    P[:, :, 3:] = np.eye(3)
                    J = Rw[:, l] @ P @ K[:, l]
                    r += dt * sgn * np.einsum("nij,ni->nj", J, F)
        rhs = np.einsum("nij,nj->ni", H, nu) + dt * (np.concatenate([np.zeros((N, 6)), tau], -1) - b) + r
        if self.solve_dtype is None:
            nu_new = np.linalg.solve(H + A, rhs[..., None])[..., 0]
        else:
            nu_new = np.linalg.solve((H + A).astype(self.solve_dtype), rhs[..., None].astype(self.solve_dtype))[..., 0].astype(np.float64)
        vlim = arr(self.desc.model.joint_vel_limit, D).astype(np.float64)
        nu_new[:, 6:] = np.clip(nu_new[:, 6:], -vlim, vlim)
        # contact sensor: net contact force per body, world frame
        cf = np.zeros((N, self.B, 3))
        for bdy, J, n, bias, Dm, w in contacts:
            up = np.einsum("nij,nj->ni", J, nu_new)
            F = n * bias[:, None] - np.einsum("nij,nj->ni", Dm, up)
            cf[:, bdy] += w[:, None] * F
        qdd = (nu_new[:, 6:] - st["qd"]) / dt
        # integrate positions with the new velocities
        st["q"] = st["q"] + dt * nu_new[:, 6:]
        st["qd"] = nu_new[:, 6:].copy()
        dq = np.concatenate([np.ones((N, 1)), 0.5 * dt * nu_new[:, :3]], -1)
        dq /= np.linalg.norm(dq, axis=-1, keepdims=True)
        quat = sp.quat_mul(st["root_quat"], dq)
        quat /= np.linalg.norm(quat, axis=-1, keepdims=True)
        st["root_quat"] = quat
        # nu+ is the spatial velocity at t+dt in the (fixed) frame that coincides with the body frame at t,
        # referred to the OLD origin location: rotate with the OLD orientation, and move the reference
        # point to the new origin (+ dt omega x v, the classical-vs-spatial acceleration term).  Using the
        # new orientation here instead integrates v_b' = -omega x v_b explicitly and makes |v| of a fast
        # spinning robot grow by sqrt(1 + (omega dt)^2) every substep.
        Ro = Rw[:, 0]
        st["root_ang_vel"] = np.einsum("nij,nj->ni", Ro, nu_new[:, :3])
        st["root_lin_vel"] = np.einsum("nij,nj->ni", Ro, nu_new[:, 3:6] + dt * np.cross(nu[:, :3], nu[:, 3:6]))
        st["root_pos"] = st["root_pos"] + dt * st["root_lin_vel"]
        return dict(contact_force=cf, joint_acc=qdd)
