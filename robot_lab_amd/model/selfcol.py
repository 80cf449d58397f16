"""Self-collision proxies for the articulations whose cfg asks for them (`enabled_self_collisions=True`: assets/unitree.py:482
G1, assets/roboparty.py:33 ATOM01 - every other asset of the reference has it off).

The reference hands the links' collision meshes to PhysX, which collides every pair of links of the articulation except
parent / child.  The lane program has no mesh-mesh test and had 2.4 KB of LDS to spare per workgroup on the trunk + limbs
instance (a second workgroup must still fit the CU), so the model here is deliberately small:

* a link is ONE capsule (segment p0 - p1 in the link frame, radius r) fitted to the spheres its collision geometry was turned into
  (model/urdf.py `_geom_to_spheres`, before the thinning the ground contact's budget asks for) - axis = principal direction of the sphere centres,
  r = the largest (distance of a centre from the axis + that sphere's radius);
* at most RL_MAX_CAPSULES links get one: the base, every trunk link that has geometry, per limb its outermost link and its two
  largest others;
* the pairs that are tested are all pairs of capsules on different links that are not parent / child, MINUS the pairs that
  already overlap (or come within `margin`) in the default joint pose - fat proxies of neighbouring links (pelvis / thigh,
  torso / shoulder) intersect where the meshes do not, and a pair that starts in contact would push forever - and minus the pairs
  that are more than `max_gap` (0.35 m) apart in that pose (a foot and the opposite shoulder, a knee and an elbow);
* a tested pair repels with an explicit penalty force `self_k * penetration` along the line between the closest points of the
  two segments, once per substep (csrc/env_step.h `self_collision_pass`, oracle/physics.py).

`fit(model, default_q, ...)` -> (capsules, pairs) for `build_desc`."""
from __future__ import annotations

import numpy as np

from .urdf import RobotModel


def link_capsule(centers: np.ndarray, radii: np.ndarray):
    """Capsule around a link's spheres: (p0, p1, r)."""
    c = centers.mean(0)
    if len(centers) == 1:
        return centers[0].copy(), centers[0].copy(), float(radii[0])
    d = centers - c
    w, vec = np.linalg.eigh(d.T @ d)
    a = vec[:, int(np.argmax(w))]
    t = d @ a
    perp = np.linalg.norm(d - np.outer(t, a), axis=1)
    r = float(np.max(perp + radii))
    # the segment ends where the outermost spheres end, pulled in by the capsule's own cap
    lo, hi = float(np.min(t - radii)) + r, float(np.max(t + radii)) - r
    if hi < lo:
        lo = hi = 0.5 * (lo + hi)
    return c + lo * a, c + hi * a, r


def _axis_angle(axis, ang):
    a = np.asarray(axis, dtype=np.float64)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1.0 - np.cos(ang)) * (K @ K)


def forward_kinematics(model: RobotModel, q: np.ndarray):
    """Link frames in the base frame for joint positions q (task joint order: link i + 1 <-> q[i])."""
    L = len(model.links)
    R, p = [np.eye(3)] * L, [np.zeros(3)] * L
    done = {0}
    while len(done) < L:
        for i in range(1, L):
            l = model.links[i]
            if i in done or l.parent not in done:
                continue
            R0 = R[l.parent] @ l.rot
            R[i] = R0 @ _axis_angle(l.axis, q[i - 1])
            p[i] = p[l.parent] + R[l.parent] @ l.origin
            done.add(i)
    return R, p


def segment_distance(a0, a1, b0, b1):
    """Distance between two segments and the closest points (Ericson, Real-Time Collision Detection 5.1.9; clamped)."""
    d1, d2, r = a1 - a0, b1 - b0, a0 - b0
    a, e, f = d1 @ d1, d2 @ d2, d2 @ r
    eps = 1e-12
    if a <= eps and e <= eps:
        s = t = 0.0
    elif a <= eps:
        s, t = 0.0, np.clip(f / e, 0.0, 1.0)
    else:
        c = d1 @ r
        if e <= eps:
            t, s = 0.0, np.clip(-c / a, 0.0, 1.0)
        else:
            b = d1 @ d2
            den = a * e - b * b
            s = np.clip((b * f - c * e) / den, 0.0, 1.0) if den > eps else 0.0
            t = (b * s + f) / e
            if t < 0.0:
                t, s = 0.0, np.clip(-c / a, 0.0, 1.0)
            elif t > 1.0:
                t, s = 1.0, np.clip((b - c) / a, 0.0, 1.0)
    pa, pb = a0 + s * d1, b0 + t * d2
    return float(np.linalg.norm(pa - pb)), pa, pb


def fit(model: RobotModel, default_q, trunk_links, limbs, max_capsules: int, max_pairs: int, margin: float = 0.01, min_radius: float = 0.015,
        max_gap: float = 0.35):
    """-> (capsules [(link, p0, p1, r)], pairs [(a, b)] with a < b indexing capsules).  `limbs`: the limb chains' link lists, root first."""
    by_link: dict[int, list] = {}
    for s in (model.geom_spheres or model.spheres):  # the geometry as read, not the thinned set the ground contact budgets
        by_link.setdefault(model.bodies[s.body].link, []).append(s)
    caps = {}
    for link, sph in by_link.items():
        p0, p1, r = link_capsule(np.array([s.center for s in sph], dtype=np.float64), np.array([s.radius for s in sph], dtype=np.float64))
        caps[link] = (p0, p1, max(r, min_radius))
    size = lambda l: np.linalg.norm(caps[l][1] - caps[l][0]) + 2.0 * caps[l][2]  # noqa: E731
    # which links: the base and the trunk links that have geometry; per limb its outermost link with geometry (hand, foot) and its
    # two largest others (G1: thigh + shin, upper arm + forearm)
    chosen = [l for l in [0] + list(trunk_links) if l in caps]
    per_limb = []
    for chain in limbs:
        have = [l for l in chain if l in caps]
        per_limb.append(([have[-1]] + sorted(have[:-1], key=size, reverse=True)[:2]) if have else [])
    # Over budget: drop the SMALLEST limb link of the limb that has the most left, round robin - never "whatever comes last", which took
    # the capsules of the last limb(s) only and left self-collision acting on one side of the body.  Loudly.
    chosen = list(dict.fromkeys(chosen))  # (a link listed twice would have taken two capsule slots)
    total = len(chosen) + sum(len(x) for x in per_limb)
    if total > max_capsules:
        import warnings

        warnings.warn(f"self-collision: {total} candidate links for {max_capsules} capsule slots - the smallest limb links are dropped, evenly over the limbs")
        while len(chosen) + sum(len(x) for x in per_limb) > max_capsules and any(len(x) > 1 for x in per_limb):
            k = max(range(len(per_limb)), key=lambda i: len(per_limb[i]))
            per_limb[k].remove(min(per_limb[k][1:], key=size))  # ([0] is the outermost link - hand, foot: kept)
    for x in per_limb:
        chosen += [l for l in x if l not in chosen]
    chosen = sorted(chosen[:max_capsules])
    capsules = [(l, *caps[l]) for l in chosen]
    R, p = forward_kinematics(model, np.asarray(default_q, dtype=np.float64))
    world = [(R[l] @ p0 + p[l], R[l] @ p1 + p[l], r) for l, p0, p1, r in capsules]
    pairs = []
    for i in range(len(capsules)):
        for j in range(i + 1, len(capsules)):
            li, lj = capsules[i][0], capsules[j][0]
            if model.links[li].parent == lj or model.links[lj].parent == li:
                continue
            d, _, _ = segment_distance(world[i][0], world[i][1], world[j][0], world[j][1])
            gap = d - world[i][2] - world[j][2]
            if gap < margin:  # in contact in the default pose: proxies of neighbouring links, not a collision to resolve
                continue
            if gap > max_gap:  # far apart when the robot stands: a foot and the opposite shoulder do not meet
                continue
            pairs.append((gap, i, j))
    # closest in the default pose first: those are the pairs that can meet - they survive a truncation, and the lane program, which
    # deals pair p to lane p % 16 in trip p / 16, finds the far pairs together in its last trips and leaves those after a bounding test
    pairs.sort()
    return capsules, [(i, j) for _, i, j in pairs[:max_pairs]]
