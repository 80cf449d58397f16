"""RobotModel + neutral spec dict -> EnvDesc (host logic).

The *spec* is a plain dict (see ``cfg_compile.compile_cfg`` which produces it from the reference's
cfg objects, and ``tests`` which write it by hand).  Name patterns follow the upstream
``resolve_matching_names`` semantics the reference relies on (``re.fullmatch`` of every pattern
against every name; e.g. ``"^(?!.*_foot).*"`` in ``.../unitree_a1/rough_env_cfg.py:116``).
"""
from __future__ import annotations

import math
import re

import numpy as np

from ..desc import EnvDesc, OBS, REW, RL_MAX_BODIES, RL_MAX_CAPSULES, RL_MAX_DOF, RL_MAX_LINKS, RL_MAX_SELF_PAIRS, RL_MAX_SPHERES, mask_of, set_arr
from .urdf import RobotModel, cap_spheres


def find_names(patterns, names, preserve_order=False):
    """Indices of ``names`` matching any regex in ``patterns`` (fullmatch)."""
    if isinstance(patterns, str):
        patterns = [patterns]
    if preserve_order:
        out = []
        for p in patterns:
            for i, n in enumerate(names):
                if re.fullmatch(p, n) and i not in out:
                    out.append(i)
        return out
    return [i for i, n in enumerate(names) if any(re.fullmatch(p, n) for p in patterns)]


def resolve_dict(val, names, default=None):
    """float | {regex: float} -> per-name list."""
    if isinstance(val, dict):
        out = [default] * len(names)
        for k, v in val.items():
            for i in find_names(k, names):
                out[i] = v
        if any(o is None for o in out):
            missing = [n for n, o in zip(names, out) if o is None]
            raise ValueError(f"no value for {missing}")
        return out
    return [val] * len(names)


def _mat_to_quat(R):
    """rotation matrix -> (w, x, y, z)"""
    R = np.asarray(R, dtype=np.float64)
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0] * 4
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q)


DEFAULT_SIM = dict(
    dt=0.005, decimation=4, gravity=9.81,
    contact_k=2.0e4, contact_c=400.0, contact_phi_ref=0.005, contact_ct=4000.0, contact_vdep=1.0,
    contact_vstick=0.01, limit_k=2000.0, limit_c=20.0, force_threshold=1.0, self_k=5.0e3,
)


def build_desc(model: RobotModel, spec: dict) -> EnvDesc:
    d = EnvDesc()
    m = d.model
    L, D, B = len(model.links), len(model.links) - 1, len(model.bodies)
    if L > RL_MAX_LINKS or D > RL_MAX_DOF or B > RL_MAX_BODIES:
        raise ValueError("model exceeds descriptor capacity")
    jn, bn = model.joint_names, model.body_names
    d.joint_names, d.body_names = list(jn), list(bn)
    m.num_links, m.num_dof, m.num_bodies = L, D, B
    # topology: serial limb chains hanging off the base or off the end of a serial trunk chain (the lane
    # program simulates one limb per lane group).  Chains are discovered by following the links, so the
    # task's joint order need not be chain-major (Go2W lists the 12 leg joints first and the 4 wheel
    # joints last, unitree_go2w/rough_env_cfg.py:25-31; G1 uses the importer's breadth-first order).
    children = {i: [c for c in range(1, L) if model.links[c].parent == i] for i in range(L)}

    def serial(r):
        ch, cur = [r], r
        while len(children[cur]) == 1:
            cur = children[cur][0]
            ch.append(cur)
        return ch, children[cur]

    def discover(continue_spine: bool):
        limbs, trunk = [], []  # limbs: (attach depth, [links])
        for r in children[0]:
            ch, tail = serial(r)
            if not tail:
                limbs.append((0, ch))
            elif not trunk:  # one branching serial chain = the trunk (G1 waist -> torso -> arms)
                trunk = ch
                subs = [serial(rr) for rr in tail]
                attach = len(trunk)
                # a spine that goes on past the branching link (FFTAI GR1: waist -> torso -> head, `GR1T1.urdf`; the arms leave it at the
                # torso): with more than four limb candidates the shortest terminal child chain continues the trunk
                others = len(children[0]) - 1
                if continue_spine and others + len(subs) > 4 and all(not t2 for _, t2 in subs):
                    k = min(range(len(subs)), key=lambda i: len(subs[i][0]))
                    trunk = trunk + subs[k][0]
                    subs = subs[:k] + subs[k + 1:]
                for ch2, tail2 in subs:
                    limbs.append((attach, ch2) if not tail2 else (None, ch2))
            else:
                limbs.append((None, ch))
        # more than four limb candidates with some of them on the base itself (Booster T1, `t1_description/urdf/robot.urdf`: a two-joint
        # neck and the arms on the trunk body, the legs behind a one-joint waist): the shortest chain on the base becomes a second PIECE
        # of the trunk - trunk joints that start again at the base (rl_model_desc.trunk_parent = -1) -, simulated redundantly by all lanes
        piece_starts = []
        while len(limbs) > 4 and any(a == 0 for a, _ in limbs):
            k = min((i for i, (a, _) in enumerate(limbs) if a == 0), key=lambda i: len(limbs[i][1]))
            if len(trunk) + len(limbs[k][1]) > 6:
                break
            piece_starts.append(len(trunk))
            trunk = trunk + limbs[k][1]
            limbs = limbs[:k] + limbs[k + 1:]
        ok = (1 <= len(limbs) <= 4 and all(a is not None for a, _ in limbs) and max(len(c) for _, c in limbs) <= 7 and len(trunk) <= 6
              and sum(len(c) for _, c in limbs) + len(trunk) == D)
        return ok, limbs, trunk, piece_starts

    ok, limbs, trunk, piece_starts = discover(True)
    rule = "continue_spine"
    if not ok:  # (GR1's rule first: it is what the committed bundles were compiled with)
        ok, limbs, trunk, piece_starts = discover(False)
        rule = "branching_trunk"
    # which rule decomposed the tree, and into what - recorded with the compiled descriptor (the choice used to be silent: ADVICE r4)
    spec["topology"] = dict(rule=rule if ok else None, trunk_links=[model.links[l].name for l in trunk],
                            piece_starts=list(piece_starts), limb_lengths=[len(c) for _, c in limbs], limb_attach=[a for a, _ in limbs])
    # fewer than 4 limbs (bipeds without arms): the spare lane groups simulate empty chains
    m.num_chains, m.chain_len, m.num_trunk = (4, max(len(c) for _, c in limbs), len(trunk)) if ok else (0, 0, 0)
    for k in range(4):
        for j in range(8):
            m.chain_link[k][j] = -1
    if ok:
        for k, (a, ch) in enumerate(limbs):
            m.chain_nj[k], m.chain_attach[k] = len(ch), a
            for j, l in enumerate(ch):
                m.chain_link[k][j] = l
        for i, l in enumerate(trunk):
            m.trunk_link[i] = l
            m.trunk_parent[i] = -1 if (i in piece_starts and i > 0) else 0
    # quadruped instances (Topo<3|4,0,3,6>) need 4 equal chains of <= 4 joints and no trunk; everything else that fits
    # runs on a trunk + limbs instance (Topo<7,3,4,9>; a trunk of 4 - 6 joints: Topo<7,6,4,9>) with inert padding joints (rl_env_host.h: topo_shape)
    quad = ok and not trunk and len(limbs) == 4 and len({len(c) for _, c in limbs}) == 1 and len(limbs[0][1]) <= 4
    # collision-sphere budget of the lane-program instance that will simulate this topology
    if ok:
        # a trunk link's spheres are hosted by the lanes whose limb hangs off it (their link group 0): a trunk link nothing hangs off
        # (G1's two inner waist links have no geometry anyway; GR1's inner waist links and its head do) cannot touch the ground here
        cap_spheres(model, [0] + list(trunk), per_link=3 if quad else 4)
        # Which lane's group 0 rides on which trunk link (depth 0 = the base).  Default: the link the lane's limb hangs off (spare lanes
        # of a biped: the base).  A spine link with collision geometry that nothing hangs off - FFTAI GR1's head, whose contact the cfg
        # terminates on (fftai_gr1t1/rough_env_cfg.py:133-139) - takes the group 0 of a lane that shares its own link with another lane,
        # deepest link first, as long as the donor link's spheres still fit the lanes it keeps (SPL = 4 sphere slots per group).
        lane_depth = [a for a, _ in limbs] + [0] * (4 - len(limbs))
        nsph = lambda link: sum(1 for sp_ in model.spheres if model.bodies[sp_.body].link == link)  # noqa: E731
        link_at = lambda d: 0 if d == 0 else trunk[d - 1]  # noqa: E731
        for dep in sorted((x for x in range(1, len(trunk) + 1) if x not in lane_depth and nsph(trunk[x - 1]) > 0), reverse=True):
            donors = [dd for dd in set(lane_depth) if lane_depth.count(dd) >= 2 and nsph(link_at(dd)) <= 4 * (lane_depth.count(dd) - 1)]
            if not donors or quad:
                break
            dd = max(donors, key=lambda x: (lane_depth.count(x), x))
            k = max(i for i in range(4) if lane_depth[i] == dd)
            lane_depth[k] = dep
        for k in range(4):
            attach = limbs[k][0] if k < len(limbs) else 0
            m.chain_grp0[k] = 0 if lane_depth[k] == attach else lane_depth[k] + 1
        # what is still unhosted cannot touch the ground here (G1's two inner waist links have no geometry anyway; GR1's inner waist
        # links do): their spheres are dropped, LOUDLY - the bundle records the links, and a termination / reward term that names a
        # body of such a link is inert
        unhosted = {l for i, l in enumerate(trunk) if (i + 1) not in lane_depth}
        dropped = sorted({model.bodies[s.body].link for s in model.spheres if model.bodies[s.body].link in unhosted})
        if dropped:
            model.spheres = [s for s in model.spheres if model.bodies[s.body].link not in unhosted]
            names = [bn[b] for b, body in enumerate(model.bodies) if body.link in dropped]
            spec["dropped_contact_bodies"] = names
            import warnings

            warnings.warn(f"collision spheres of trunk links nothing rides on were dropped: bodies {names} cannot touch the ground in this "
                          f"simulator (illegal_contact / undesired_contacts on them are inert)")
    G = len(model.spheres)
    if G > RL_MAX_SPHERES:
        raise ValueError("model exceeds descriptor capacity (collision spheres)")
    m.num_spheres = G
    for i, l in enumerate(model.links):
        m.link_parent[i] = l.parent
        set_arr(m.link_origin[i], l.origin)
        set_arr(m.link_quat[i], _mat_to_quat(l.rot))
        set_arr(m.link_axis[i], l.axis)
    rob = spec["robot"]
    set_arr(m.joint_lower, [l.lower for l in model.links[1:]])
    set_arr(m.joint_upper, [l.upper for l in model.links[1:]])
    set_arr(m.joint_vel_limit, [l.vel_limit for l in model.links[1:]])
    dq = resolve_dict(rob["init_joint_pos"], jn, 0.0)
    dqd = resolve_dict(rob.get("init_joint_vel", 0.0), jn, 0.0)
    set_arr(m.default_joint_pos, dq)
    set_arr(m.default_joint_vel, dqd)
    f = rob.get("soft_joint_pos_limit_factor", 1.0)
    lo = np.array([l.lower for l in model.links[1:]])
    hi = np.array([l.upper for l in model.links[1:]])
    mid, rng = 0.5 * (lo + hi), hi - lo
    set_arr(m.soft_lower, mid - 0.5 * rng * f)
    set_arr(m.soft_upper, mid + 0.5 * rng * f)
    for b, body in enumerate(model.bodies):
        m.body_link[b] = body.link
        set_arr(m.body_pos[b], body.pos)
        m.body_mass[b] = body.mass
        set_arr(m.body_com[b], body.com)
        I = body.inertia
        set_arr(m.body_inertia[b], [I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]])
    for g, s in enumerate(model.spheres):
        m.sphere_body[g] = s.body
        set_arr(m.sphere_center[g], s.center)
        m.sphere_radius[g] = s.radius
    # self-collision proxies (model/selfcol.py): only where the cfg enables it, and only on the trunk + limbs instance
    m.self_collision = 1 if rob.get("self_collisions") else 0
    m.num_capsules = m.num_self_pairs = 0
    if m.self_collision and ok and not quad:
        from .selfcol import fit

        caps, pairs = fit(model, dq, list(trunk), [ch for _, ch in limbs], RL_MAX_CAPSULES, RL_MAX_SELF_PAIRS)
        m.num_capsules, m.num_self_pairs = len(caps), len(pairs)
        # the pair force is an EXPLICIT spring (csrc/env_step.h self_apply): stable only while self_k dt^2 stays well below the inertia it
        # acts on - checked against the lightest link that carries a capsule of a listed pair
        sim_ = dict(DEFAULT_SIM)
        sim_.update(spec.get("sim", {}))
        k_dt2 = float(sim_["self_k"]) * float(sim_["dt"]) ** 2
        in_pairs = {caps[i][0] for ab in pairs for i in ab}
        link_mass = {l: sum(b.mass for b in model.bodies if b.link == l) for l in in_pairs}
        if link_mass and k_dt2 > 0.5 * min(link_mass.values()):
            import warnings

            light = min(link_mass, key=link_mass.get)
            warnings.warn(f"self-collision: self_k dt^2 = {k_dt2:.3f} kg against {link_mass[light]:.3f} kg of link '{model.links[light].name}': the explicit penalty spring may ring")
        for c, (link, p0, p1, r) in enumerate(caps):
            m.capsule_link[c], m.capsule_radius[c] = link, r
            set_arr(m.capsule_p0[c], p0)
            set_arr(m.capsule_p1[c], p1)
        for i, (a, b) in enumerate(pairs):
            m.self_pair[i][0], m.self_pair[i][1] = a, b
    set_arr(m.default_root_pos, rob["init_pos"])
    set_arr(m.default_root_quat, rob.get("init_rot", (1.0, 0.0, 0.0, 0.0)))
    # actuators
    assigned = [False] * D
    arm = [0.0] * D
    for act in rob["actuators"]:
        ids = find_names(act["joint_names_expr"], jn)
        for key, dst in (("stiffness", m.act_kp), ("damping", m.act_kd), ("effort_limit", m.act_effort_limit),
                         ("saturation_effort", m.act_saturation), ("velocity_limit", m.act_vel_limit)):
            vals = resolve_dict(act[key], [jn[i] for i in ids]) if act.get(key) is not None else [1e9] * len(ids)
            for i, v in zip(ids, vals):
                dst[i] = v
        a = resolve_dict(act.get("armature") or 0.0, [jn[i] for i in ids], 0.0)
        for i, v in zip(ids, a):
            m.act_implicit[i] = 1 if act["type"] == "implicit" else 0
            arm[i] = v
            assigned[i] = True
    if not all(assigned) and rob.get("unactuated_passive"):
        for i in range(D):
            if not assigned[i]:  # passive hinge: no drive, no torque
                m.act_kp[i] = m.act_kd[i] = m.act_effort_limit[i] = m.act_saturation[i] = 0.0
                m.act_vel_limit[i] = 1e9
                m.act_implicit[i] = 1
                assigned[i] = True
    if not all(assigned):
        raise ValueError(f"joints without actuator: {[n for n, a in zip(jn, assigned) if not a]}")
    set_arr(m.joint_armature, arm)
    # actions: terms consume consecutive slices of the action vector in declaration order [UPSTREAM B2]
    a_idx = 0
    seen = [False] * D
    order = []
    for at in spec["actions"]:
        ids = find_names(at["joint_names"], jn, preserve_order=at.get("preserve_order", False))
        names = [jn[i] for i in ids]
        scale = resolve_dict(at["scale"], names)
        clip = at.get("clip")
        for k, i in enumerate(ids):
            m.action_is_vel[i] = 1 if at["type"] == "vel" else 0
            m.action_scale[i] = scale[k]
            off = (dqd[i] if at["type"] == "vel" else dq[i]) if at.get("use_default_offset", True) else 0.0
            m.action_offset[i] = off
            lohi = (-1e30, 1e30)
            if clip:
                for pat, v in clip.items():
                    if re.fullmatch(pat, jn[i]):
                        lohi = v
            m.action_clip_lo[i], m.action_clip_hi[i] = lohi
            seen[i] = True
            order.append(i)
        a_idx += len(ids)
    if order != list(range(D)):
        raise ValueError("action terms must cover the joints in task joint order")
    # sim
    sim = dict(DEFAULT_SIM)
    sim.update(spec.get("sim", {}))
    for k, v in sim.items():
        setattr(d.sim, k, v)
    # terrain
    ter = spec["terrain"]
    for k, v in ter.items():
        if k != "heights":
            setattr(d.terrain, k, v)
    # task
    t = d.task
    ts = spec["task"]
    t.episode_length_s = ts["episode_length_s"]
    c = ts["command"]
    set_arr(t.cmd_range, [c["lin_vel_x"], c["lin_vel_y"], c["ang_vel_z"], c["heading"]])
    set_arr(t.cmd_resample, c["resampling_time_range"])
    t.cmd_rel_standing, t.cmd_rel_heading = c["rel_standing_envs"], c["rel_heading_envs"]
    t.cmd_heading_stiffness, t.cmd_heading = c["heading_control_stiffness"], int(c["heading_command"])
    t.cmd_small_threshold = c.get("small_threshold", 0.2)
    for gname, dst, cnt_attr in (("policy", t.policy, "n_policy"), ("critic", t.critic, "n_critic")):
        grp = ts["observations"][gname]
        for i, ot in enumerate(grp["terms"]):
            o = dst[i]
            o.kind = OBS[ot["func"]]
            o.scale = 1.0 if ot.get("scale") is None else ot["scale"]
            clip = ot.get("clip") or (-1e30, 1e30)
            o.clip_lo, o.clip_hi = clip
            if ot.get("noise"):
                o.has_noise, (o.noise_lo, o.noise_hi) = 1, ot["noise"]
            if ot["func"] == "joint_pos_rel_without_wheel":
                t.wheel_joint_mask = mask_of(find_names(ot["wheel_joint_names"], jn))
        setattr(t, cnt_attr, len(grp["terms"]))
        setattr(t, f"{gname}_corrupt", int(grp.get("enable_corruption", False)))
    sc = ts.get("height_scan") or dict(size=(1.6, 1.0), resolution=0.1, offset=0.5)
    t.scan_res = sc["resolution"]
    t.scan_nx = int(round(sc["size"][0] / sc["resolution"])) + 1
    t.scan_ny = int(round(sc["size"][1] / sc["resolution"])) + 1
    t.scan_offset = sc.get("offset", 0.5)
    t.scan_body = find_names(sc["body"], bn)[0] if sc.get("body") else 0
    d.reward_names = []
    for i, rt in enumerate(ts["rewards"]):
        r = t.rewards[i]
        r.kind = REW[rt["func"]]
        r.weight = rt["weight"]
        set_arr(r.p, rt.get("p", []))
        if rt.get("joint_names") is not None:
            r.joint_mask = mask_of(find_names(rt["joint_names"], jn))
        else:
            r.joint_mask = (1 << D) - 1
        if rt.get("body_names") is not None:
            r.body_mask = mask_of(find_names(rt["body_names"], bn))
        if rt["func"] == "action_sync":  # every group's action columns one after the other, the group of each beside it (rewards.py:305-337)
            ia, ib = [], []
            for gi, grp in enumerate(rt["joint_groups"]):
                per_name = [find_names(name, jn) for name in grp]  # (the reference resolves every name on its own: asset.find_joints(joint_name))
                if any(len(c) != 1 for c in per_name):
                    raise ValueError("action_sync: a joint name of a group must match exactly one joint (the reference stacks one action column per name)")
                cols = [c[0] for c in per_name]
                if len(cols) >= 2:
                    ia += cols
                    ib += [gi] * len(cols)
            if len(ia) > 16 or len(rt["joint_groups"]) > 8:
                raise ValueError("action_sync: more than 16 joints / 8 groups")
            set_arr(r.idx_a, ia)
            set_arr(r.idx_b, ib)
            r.n_idx = len(ia)
            r.p[0] = 1.0 / len(rt["joint_groups"]) if rt["joint_groups"] else 0.0
        if rt["func"] in ("joint_mirror", "action_mirror"):
            ia, ib = [], []
            for pa, pb in rt["mirror_joints"]:
                a, b = find_names(pa, jn), find_names(pb, jn)
                if len(a) != len(b):
                    raise ValueError("mirror joint groups differ in size")
                ia += a
                ib += b
            set_arr(r.idx_a, ia)
            set_arr(r.idx_b, ib)
            r.n_idx = len(ia)
            r.p[0] = 1.0 / len(rt["mirror_joints"]) if rt["mirror_joints"] else 0.0
        if rt["func"] == "wheel_vel_penalty":  # in_air[:, body_ids] * joint_vel[:, joint_ids], elementwise (rewards.py:146)
            ia, ib = find_names(rt["body_names"], bn), find_names(rt["joint_names"], jn)
            if len(ia) != len(ib):
                raise ValueError("wheel_vel_penalty pairs one wheel body with one wheel joint")
            set_arr(r.idx_a, ia)
            set_arr(r.idx_b, ib)
            r.n_idx = len(ia)
        if rt["func"] in ("feet_distance_y_exp", "feet_distance_xy_exp"):  # the foot's place in body_ids picks its side
            ia = find_names(rt["body_names"], bn)
            if rt["func"] == "feet_distance_xy_exp" and len(ia) != 4:
                raise ValueError("feet_distance_xy_exp is written for 4 feet")
            set_arr(r.idx_a, ia)
            r.n_idx = len(ia)
        if rt["func"] == "GaitReward":
            pairs = rt["synced_feet_pair_names"]
            feet = [find_names(list(pairs[0]), bn, True), find_names(list(pairs[1]), bn, True)]
            set_arr(r.idx_a, [feet[0][0], feet[0][1], feet[1][0], feet[1][1]])
            r.n_idx = 4
        d.reward_names.append(rt["name"])
    t.n_rewards = len(ts["rewards"])
    tm = ts["terminations"]
    t.term_time_out = int(tm.get("time_out", True))
    t.term_out_of_bounds = int(tm.get("terrain_out_of_bounds") is not None)
    t.oob_buffer = (tm.get("terrain_out_of_bounds") or {}).get("distance_buffer", 3.0)
    if tm.get("illegal_contact") is not None:
        t.term_illegal_contact = 1
        t.illegal_body_mask = mask_of(find_names(tm["illegal_contact"]["body_names"], bn))
        t.illegal_threshold = tm["illegal_contact"]["threshold"]
    ev = ts["events"]
    t.base_body = find_names(ts["base_body_name"], bn)[0]
    for key, pre in (("lin_vel", "cur_cmd_lin"), ("ang_vel", "cur_cmd_ang")):  # command_levels_* curricula (curriculums.py:21-94)
        c = (ts.get("command_levels") or {}).get(key)
        if c:
            setattr(t, pre, 1)
            setattr(t, pre + "_term", d.reward_names.index(c["reward_term_name"]))
            set_arr(getattr(t, pre + "_mult"), c["range_multiplier"])
    if ev.get("material"):
        e = ev["material"]
        t.ev_material = 1
        set_arr(t.friction_static, e["static_friction_range"])
        set_arr(t.friction_dynamic, e["dynamic_friction_range"])
        set_arr(t.restitution, e["restitution_range"])
        t.friction_buckets = e["num_buckets"]
    if ev.get("mass_base"):
        t.ev_mass_base = 1
        set_arr(t.mass_base_add, ev["mass_base"]["range"])
        t.mass_base_mask = mask_of(find_names(ev["mass_base"]["body_names"], bn))
    if ev.get("mass_others"):
        t.ev_mass_others = 1
        set_arr(t.mass_scale, ev["mass_others"]["range"])
        t.mass_scale_mask = mask_of(find_names(ev["mass_others"]["body_names"], bn))
    if ev.get("com"):
        t.ev_com = 1
        set_arr(t.com_range, [ev["com"]["range"].get(k, (0.0, 0.0)) for k in "xyz"])
        t.com_mask = mask_of(find_names(ev["com"]["body_names"], bn))
    if ev.get("wrench"):
        t.ev_wrench = 1
        set_arr(t.wrench_force, ev["wrench"]["force_range"])
        set_arr(t.wrench_torque, ev["wrench"]["torque_range"])
    if ev.get("reset_joints"):
        t.ev_reset_joints = 1
        set_arr(t.reset_joint_pos_scale, ev["reset_joints"]["position_range"])
        set_arr(t.reset_joint_vel_scale, ev["reset_joints"]["velocity_range"])
    if ev.get("gains"):
        t.ev_gains = 1
        set_arr(t.gain_kp_scale, ev["gains"]["stiffness"])
        set_arr(t.gain_kd_scale, ev["gains"]["damping"])
    if ev.get("reset_base"):
        t.ev_reset_base = 1
        keys = ["x", "y", "z", "roll", "pitch", "yaw"]
        set_arr(t.reset_pose, [ev["reset_base"]["pose_range"].get(k, (0.0, 0.0)) for k in keys])
        set_arr(t.reset_vel, [ev["reset_base"]["velocity_range"].get(k, (0.0, 0.0)) for k in keys])
    if ev.get("push"):
        t.ev_push = 1
        keys = ["x", "y", "z", "roll", "pitch", "yaw"]
        set_arr(t.push_interval, ev["push"]["interval_range_s"])
        set_arr(t.push_vel, [ev["push"]["velocity_range"].get(k, (0.0, 0.0)) for k in keys])
    return d


def max_episode_length(desc: EnvDesc) -> int:
    return int(math.ceil(desc.task.episode_length_s / (desc.sim.decimation * desc.sim.dt) - 1e-4))
