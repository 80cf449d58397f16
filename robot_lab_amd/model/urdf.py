"""URDF -> articulated-model tables (host logic, numpy only).

Consumes the robot description data the reference points its ``ArticulationCfg`` at
(``/root/reference/source/robot_lab/robot_lab/assets/unitree.py:19-65`` for A1,
``:71-117`` Go2, ``:121-175`` Go2W) and reproduces what the upstream URDF importer does with
``merge_fixed_joints=True`` (``unitree.py:22``): links connected by ``fixed`` joints are collapsed
into their parent *body* unless the joint carries ``dont_collapse="true"``
(``a1.urdf:461``), in which case the child stays a separate (sensor) body.

Two index spaces come out of this:

* **links** - the rigid bodies that move relative to each other (base + one per actuated joint);
  this is what the dynamics see.  A ``dont_collapse`` foot is rigidly attached to its calf, so it
  is *not* a link.
* **bodies** - the body list the contact sensor / randomisation events / ``body_names`` regexes
  of the reference address (A1: 17 = base + 4x{hip, thigh, calf, foot}).  Each body belongs to one
  link.

Collision primitives (``a1.urdf:326-331,379-384,397-404,421-426,449-454``) are converted to a small
set of collision *spheres* per body (see :func:`_geom_to_spheres`); the simulator's contact model
is sphere-vs-heightfield.  Mesh collision geometry (G1: ``g1_29dof_rev_1_0.urdf`` STL files under
``g1_description/meshes``) is replaced by a capsule-like row of <= 3 spheres fitted to the vertex
cloud (:func:`_mesh_to_spheres`).

Joint frames may be rotated with respect to the parent link (``rpy`` of the joint origin, G1
``g1_29dof_rev_1_0.urdf:124,182,648,677``): ``Link.rot`` is that fixed rotation.
"""
from __future__ import annotations

import math
import os
import struct
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np


def rpy_to_mat(rpy) -> np.ndarray:
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ]
    )


def _vec(s, n=3, default=0.0):
    if s is None:
        return np.full(n, default, dtype=np.float64)
    return np.array([float(v) for v in s.split()], dtype=np.float64)


@dataclass
class _Inertial:
    mass: float
    com: np.ndarray  # in owner frame
    inertia: np.ndarray  # 3x3 about com, owner-frame axes


@dataclass
class Sphere:
    body: int
    center: np.ndarray  # in LINK frame
    radius: float
    explicit: bool = False  # a <sphere> primitive of the URDF (kept as is), not a fit of another shape


@dataclass
class Body:
    name: str
    link: int
    pos: np.ndarray  # body frame origin in link frame
    rot: np.ndarray  # body frame axes in link frame
    mass: float
    com: np.ndarray  # in LINK frame
    inertia: np.ndarray  # 3x3 about com, LINK-frame axes


@dataclass
class Link:
    name: str
    parent: int
    joint_name: str
    joint_type: str  # "floating" | "revolute" | "continuous"
    origin: np.ndarray  # joint origin in parent link frame
    rot: np.ndarray  # joint frame axes in the parent link frame (fixed rotation, URDF joint rpy)
    axis: np.ndarray
    lower: float
    upper: float
    vel_limit: float
    effort_limit: float


@dataclass
class RobotModel:
    name: str
    links: list[Link] = field(default_factory=list)
    bodies: list[Body] = field(default_factory=list)
    spheres: list[Sphere] = field(default_factory=list)
    geom_spheres: list[Sphere] = field(default_factory=list)  # the spheres of the collision geometry before pruning / thinning / capping (capsule fits: model/selfcol.py)

    @property
    def joint_names(self):
        return [l.joint_name for l in self.links[1:]]

    @property
    def body_names(self):
        return [b.name for b in self.bodies]

    def total_mass(self):
        return float(sum(b.mass for b in self.bodies))


def _read_stl_vertices(path: str) -> np.ndarray:
    """Vertices [n, 3] of a binary (or ASCII) STL file."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) >= 84:
        ntri = struct.unpack_from("<I", data, 80)[0]
        if 84 + 50 * ntri == len(data):
            rec = np.frombuffer(data, dtype=np.uint8, count=50 * ntri, offset=84).reshape(ntri, 50)
            return rec[:, 12:48].copy().view("<f4").reshape(-1, 3).astype(np.float64)
    verts = [[float(v) for v in line.split()[1:4]] for line in data.decode("ascii", "ignore").splitlines()
             if line.strip().startswith("vertex")]
    return np.asarray(verts, dtype=np.float64)


def _read_dae_vertices(path: str) -> np.ndarray:
    """Vertices [n, 3] of a COLLADA file (Unitree B2W: calves and wheels, `b2w_description.urdf` FL_calf / FL_foot collision):
    the position arrays of every geometry, carried through the transforms of the scene nodes that instance them - <matrix>
    (Blender exports millimetres with a 0.001 scale matrix), <translate>, <rotate> (axis + degrees) and <scale>, composed in
    document order as the COLLADA 1.4 specification prescribes -, the asset's unit, and its up axis (Y_UP / X_UP files are
    turned into the Z_UP convention URDF meshes are interpreted in).  Anything else that is malformed raises ValueError: the
    caller treats that as "unreadable file: no spheres" (_read_mesh_vertices)."""
    ns = {"c": "http://www.collada.org/2005/11/COLLADASchema"}
    root = ET.parse(path).getroot()
    geoms = {}
    for g in root.iterfind(".//c:library_geometries/c:geometry", ns):
        mesh = g.find("c:mesh", ns)
        vin = mesh.find("c:vertices/c:input[@semantic='POSITION']", ns) if mesh is not None else None
        if vin is None:
            continue
        src = mesh.find(f"c:source[@id='{vin.get('source')[1:]}']/c:float_array", ns)
        if src is not None and src.text:
            geoms[g.get("id")] = np.array(src.text.split(), dtype=np.float64).reshape(-1, 3)
    unit = root.find("c:asset/c:unit", ns)
    meter = float(unit.get("meter", "1")) if unit is not None else 1.0
    up = root.find("c:asset/c:up_axis", ns)
    up_axis = (up.text or "Y_UP").strip().upper() if up is not None else "Y_UP"  # the specification's default
    # right-handed change of basis into Z_UP (COLLADA 1.4.1 "up_axis": Y_UP = x right, y up, z in;  X_UP = x up, y left (-), z in)
    UP = {"Z_UP": np.eye(3), "Y_UP": np.array([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]]), "X_UP": np.array([[0, -1.0, 0], [0, 0, -1.0], [1.0, 0, 0]])}
    if up_axis not in UP:
        raise ValueError(f"{path}: unknown up_axis {up_axis!r}")
    out = []
    tag = lambda el: el.tag.rsplit("}", 1)[-1]  # noqa: E731

    def local(el):
        v = np.array((el.text or "").split(), dtype=np.float64)
        M = np.eye(4)
        kind = tag(el)
        if kind == "matrix" and v.size == 16:
            M = v.reshape(4, 4)
        elif kind == "translate" and v.size == 3:
            M[:3, 3] = v
        elif kind == "scale" and v.size == 3:
            M[:3, :3] = np.diag(v)
        elif kind == "rotate" and v.size == 4:
            a, th = v[:3] / max(np.linalg.norm(v[:3]), 1e-30), np.deg2rad(v[3])
            K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            M[:3, :3] = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
        elif kind in ("lookat", "skew"):
            raise ValueError(f"{path}: <{kind}> node transforms are not supported")
        elif kind in ("matrix", "translate", "scale", "rotate"):
            raise ValueError(f"{path}: malformed <{kind}>")
        else:
            return None
        return M

    def visit(node, M):
        for el in node:  # transforms apply in document order, ahead of the geometry and the child nodes they precede or follow alike
            L = local(el)
            if L is not None:
                M = M @ L
        for ig in node.findall("c:instance_geometry", ns):
            v = geoms.get(ig.get("url")[1:])
            if v is not None:
                out.append((v @ M[:3, :3].T + M[:3, 3]) * meter)
        for ch in node.findall("c:node", ns):
            visit(ch, M)

    for scene in root.iterfind(".//c:library_visual_scenes/c:visual_scene", ns):
        for node in scene.findall("c:node", ns):
            visit(node, np.eye(4))
    if not out:  # no scene graph: the bare arrays
        out = [v * meter for v in geoms.values()]
    verts = np.concatenate(out) if out else np.zeros((0, 3))
    return verts @ UP[up_axis].T


def _read_obj_vertices(path: str) -> np.ndarray:
    """Vertices [n, 3] of a Wavefront OBJ file (RobotEra Xbot: four links)."""
    with open(path, "r", errors="ignore") as f:
        return np.asarray([[float(x) for x in line.split()[1:4]] for line in f if line.startswith("v ")], dtype=np.float64).reshape(-1, 3)


def _read_mesh_vertices(path: str) -> np.ndarray:
    ext = os.path.splitext(path)[1].lower()
    try:
        return _read_dae_vertices(path) if ext == ".dae" else _read_obj_vertices(path) if ext == ".obj" else _read_stl_vertices(path)
    except (ET.ParseError, ValueError, IndexError, KeyError, AttributeError, OSError) as exc:  # -> "unreadable file: no spheres" (_mesh_to_spheres)
        import warnings

        warnings.warn(f"collision mesh {path} could not be read ({type(exc).__name__}: {exc}); the link gets no spheres from it")
        return np.zeros((0, 3))


def _mesh_to_spheres(verts: np.ndarray, scale: np.ndarray):
    """Capsule-like sphere row fitted to a vertex cloud: principal axis a (largest variance), half
    length h, radius r = 90th percentile of the distance from the axis (clamped to [1 cm, h]).
    h <= 1.5 r -> one sphere of radius max(r, h) at the centre; else 3 spheres at 0, +-(h - r) a."""
    if verts.ndim != 2 or len(verts) < 4:  # unreadable file: no spheres
        return []
    v = np.unique(np.round(verts * scale[None], 5), axis=0)
    if len(v) < 4:
        return []
    lo, hi = v.min(0), v.max(0)
    c = 0.5 * (lo + hi)
    w, vec = np.linalg.eigh(np.cov((v - v.mean(0)).T))
    a = vec[:, int(np.argmax(w))]
    t = (v - c) @ a
    c = c + 0.5 * (t.max() + t.min()) * a - ((c - c) @ a) * a
    t = (v - c) @ a
    h = 0.5 * (t.max() - t.min())
    perp = np.linalg.norm((v - c) - np.outer(t, a), axis=1)
    r = float(np.clip(np.percentile(perp, 90), 0.01, max(h, 0.01)))
    if h <= 1.5 * r:
        return [(c, float(max(r, h)))]
    return [(c + s * (h - r) * a, r) for s in (-1.0, 0.0, 1.0)]


def _geom_to_spheres(geom: ET.Element, T_pos: np.ndarray, T_rot: np.ndarray, mesh_dir: str | None = None):
    """Collision primitive -> list of (center, radius) in the frame T maps into.

    sphere   -> itself.
    box      -> slender (longest side >= 3x the second longest): 3 spheres along the long axis with
                radius = half the smaller cross-section side; otherwise 8 corner spheres inset by
                r = min(0.02, min_half_extent) (a rounded box).
    cylinder -> length <= 2.5 r: one sphere of the cylinder radius at the centre; else 3 along the axis.
    mesh     -> :func:`_mesh_to_spheres` of the vertex cloud of the STL / COLLADA / OBJ file (looked up in ``mesh_dir``).
    """
    out = []
    sph = geom.find("sphere")
    box = geom.find("box")
    cyl = geom.find("cylinder")
    mesh = geom.find("mesh")
    if mesh is not None and mesh_dir is not None:
        fn = os.path.join(mesh_dir, os.path.basename(mesh.get("filename")))
        if os.path.isfile(fn):
            out = _mesh_to_spheres(_read_mesh_vertices(fn), _vec(mesh.get("scale"), 3, 1.0))
    elif sph is not None:
        out.append((np.zeros(3), float(sph.get("radius"))))
    elif box is not None:
        size = _vec(box.get("size"))
        if np.max(size) < 0.01:
            return []  # marker boxes (imu etc., a1.urdf:353-358)
        order = np.argsort(size)[::-1]
        if size[order[0]] >= 3.0 * size[order[1]]:
            r = 0.5 * size[order[2]]
            half = 0.5 * size[order[0]]
            for t in (-half, 0.0, half):
                c = np.zeros(3)
                c[order[0]] = t
                out.append((c, r))
        else:
            r = min(0.02, 0.5 * float(np.min(size)))
            for sx in (-1, 1):
                for sy in (-1, 1):
                    for sz in (-1, 1):
                        c = 0.5 * size * np.array([sx, sy, sz]) - r * np.array([sx, sy, sz])
                        out.append((c, r))
    elif cyl is not None:
        r = float(cyl.get("radius"))
        length = float(cyl.get("length"))
        if length <= 2.5 * r:
            out.append((np.zeros(3), r))
        else:
            for t in (-0.5 * length, 0.0, 0.5 * length):
                out.append((np.array([0.0, 0.0, t]), r))
    return [(T_pos + T_rot @ c, r, sph is not None) for c, r in out]


def load_urdf(path: str, name: str | None = None, joint_order: list[str] | None = None) -> RobotModel:
    """Parse a URDF and collapse fixed joints.

    ``joint_order``: the order the task config addresses the joints in
    (``.../unitree_a1/rough_env_cfg.py:22-27``, ``preserve_order=True``); links are emitted in that
    order (link i+1 <-> joint_order[i]).  Default: breadth-first URDF order.
    """
    root = ET.parse(path).getroot()
    mesh_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(path))), "meshes")
    ulinks = {l.get("name"): l for l in root.findall("link")}
    children: dict[str, list[ET.Element]] = {}
    child_names = set()
    for j in root.findall("joint"):
        if j.find("parent") is None or j.find("child") is None:
            continue  # <transmission><joint .../> entries have no parent/child
        p = j.find("parent").get("link")
        if p not in ulinks:
            continue
        children.setdefault(p, []).append(j)
        child_names.add(j.find("child").get("link"))
    roots = [n for n in ulinks if n not in child_names]
    if len(roots) != 1:
        raise ValueError(f"URDF {path}: expected one root link, found {roots}")

    model = RobotModel(name=name or root.get("name", "robot"))
    # BFS over moving joints; fixed children are folded into the current link.
    model.links.append(Link(roots[0], -1, "floating_base", "floating", np.zeros(3), np.eye(3), np.array([0, 0, 1.0]), 0, 0, 0, 0))
    pending = [(roots[0], 0, np.zeros(3), np.eye(3), None)]  # (urdf link, link idx, pos, rot in link frame, body idx)
    raw_bodies: list[dict] = []

    def new_body(bname, link_idx, pos, rot):
        raw_bodies.append(dict(name=bname, link=link_idx, pos=pos.copy(), rot=rot.copy(), inertials=[], spheres=[]))
        return len(raw_bodies) - 1

    queue = list(pending)
    while queue:
        uname, lidx, pos, rot, bidx = queue.pop(0)
        if bidx is None:
            bidx = new_body(uname, lidx, pos, rot)
        ul = ulinks[uname]
        ine = ul.find("inertial")
        if ine is not None:
            o = ine.find("origin")
            ipos = _vec(o.get("xyz")) if o is not None else np.zeros(3)
            irot = rpy_to_mat(_vec(o.get("rpy"))) if o is not None and o.get("rpy") else np.eye(3)
            m = float(ine.find("mass").get("value"))
            it = ine.find("inertia")
            I = np.array(
                [
                    [float(it.get("ixx")), float(it.get("ixy")), float(it.get("ixz"))],
                    [float(it.get("ixy")), float(it.get("iyy")), float(it.get("iyz"))],
                    [float(it.get("ixz")), float(it.get("iyz")), float(it.get("izz"))],
                ]
            )
            R = rot @ irot
            raw_bodies[bidx]["inertials"].append(_Inertial(m, pos + rot @ ipos, R @ I @ R.T))
        for col in ul.findall("collision"):
            o = col.find("origin")
            cpos = _vec(o.get("xyz")) if o is not None else np.zeros(3)
            crot = rpy_to_mat(_vec(o.get("rpy"))) if o is not None and o.get("rpy") else np.eye(3)
            g = col.find("geometry")
            raw_bodies[bidx]["spheres"] += _geom_to_spheres(g, pos + rot @ cpos, rot @ crot, mesh_dir)
        for j in children.get(uname, []):
            cname = j.find("child").get("link")
            o = j.find("origin")
            jpos = _vec(o.get("xyz")) if o is not None else np.zeros(3)
            jrot = rpy_to_mat(_vec(o.get("rpy"))) if o is not None and o.get("rpy") else np.eye(3)
            jtype = j.get("type")
            if jtype == "fixed":
                keep = j.get("dont_collapse", "false").lower() == "true"
                queue.append((cname, lidx, pos + rot @ jpos, rot @ jrot, None if keep else bidx))
            elif jtype in ("revolute", "continuous"):
                lim = j.find("limit")
                lo = float(lim.get("lower", "-1e9")) if (lim is not None and jtype == "revolute") else -1e9
                hi = float(lim.get("upper", "1e9")) if (lim is not None and jtype == "revolute") else 1e9
                vl = float(lim.get("velocity", "1e9")) if lim is not None else 1e9
                ef = float(lim.get("effort", "1e9")) if lim is not None else 1e9
                # CAD exporters write velocity="0" effort="0" for "not specified" (agibot/d1/urdf/edu.urdf:99-103): a literal zero would
                # freeze the joint - which no importer does; like the missing attribute it means no limit from the URDF (the actuator
                # cfg sets effort / velocity limits of its own, assets/agibot.py)
                if vl <= 0.0:
                    vl = 1e9
                if ef <= 0.0:
                    ef = 1e9
                ax = _vec(j.find("axis").get("xyz")) if j.find("axis") is not None else np.array([1.0, 0, 0])
                model.links.append(Link(cname, lidx, j.get("name"), jtype, pos + rot @ jpos, rot @ jrot, ax / np.linalg.norm(ax), lo, hi, vl, ef))
                queue.append((cname, len(model.links) - 1, np.zeros(3), np.eye(3), None))
            else:
                raise NotImplementedError(f"joint type {jtype}")

    # reorder links to the task's joint order
    if joint_order is not None:
        names = [l.joint_name for l in model.links]
        perm = [0] + [names.index(n) for n in joint_order]
        if sorted(perm) != list(range(len(model.links))):
            raise ValueError("joint_order must list every moving joint exactly once")
        inv = {old: new for new, old in enumerate(perm)}
        model.links = [model.links[i] for i in perm]
        for l in model.links[1:]:
            l.parent = inv[l.parent]
        for b in raw_bodies:
            b["link"] = inv[b["link"]]

    # finalise bodies: composite inertial per body, expressed in the link frame
    for bi, rb in enumerate(raw_bodies):
        m = sum(i.mass for i in rb["inertials"])
        if m > 0:
            com = sum(i.mass * i.com for i in rb["inertials"]) / m
            I = np.zeros((3, 3))
            for i in rb["inertials"]:
                d = i.com - com
                I += i.inertia + i.mass * (d @ d * np.eye(3) - np.outer(d, d))
        else:
            com, I = rb["pos"].copy(), np.zeros((3, 3))
        model.bodies.append(Body(rb["name"], rb["link"], rb["pos"], rb["rot"], m, com, I))
        for c, r, ex in rb["spheres"]:
            model.spheres.append(Sphere(bi, c, r, ex))
    model.geom_spheres = list(model.spheres)
    _prune_contained_spheres(model)
    _thin_spheres(model)
    return model


def cap_spheres(model: RobotModel, trunk_links, per_link: int, per_trunk_body: int = 8):
    """The lane program budgets `per_link` collision spheres per limb link (3 on the quadruped instances, 4 on G1)
    and up to `per_trunk_body` per body of a trunk link (spread over the lanes).  Links that still exceed it after
    thinning (Unitree B2 trunk, ZSL1 calves) keep a farthest-point subset: the largest sphere first, then
    repeatedly the one farthest from those kept.  Called by `build_desc` once the topology is known."""
    trunk_links = set(trunk_links)
    groups: dict[tuple[int, int], list[int]] = {}
    for i, sph in enumerate(model.spheres):
        link = model.bodies[sph.body].link
        groups.setdefault((link, sph.body if link in trunk_links else -1), []).append(i)
    budgets = {key: (per_trunk_body if key[0] in trunk_links else per_link) for key in groups}
    if len(trunk_links) == 1:
        # no trunk joints: the bodies of the base link share the 4 lanes' base groups, one body per lane and
        # `per_link` spheres per lane (MagicLab Dog-W: an 8-sphere base and a 4-sphere head are 5 lanes' worth)
        base = [key for key in groups if key[0] in trunk_links]
        lanes = lambda: sum(-(-min(len(groups[k]), budgets[k]) // per_link) for k in base)
        while base and lanes() > 4:
            big = max(base, key=lambda k: min(len(groups[k]), budgets[k]))
            budgets[big] = min(len(groups[big]), budgets[big]) - 1
    keep = set()
    for key, ids in groups.items():
        link, budget = key[0], budgets[key]
        if len(ids) <= budget:
            keep.update(ids)
            continue
        kept = [max(ids, key=lambda i: model.spheres[i].radius)]
        while len(kept) < budget:
            rest = [i for i in ids if i not in kept]
            kept.append(max(rest, key=lambda i: min(np.linalg.norm(model.spheres[i].center - model.spheres[k].center) for k in kept)))
        keep.update(kept)
    model.spheres = [sph for i, sph in enumerate(model.spheres) if i in keep]


def _thin_spheres(model: RobotModel, min_sep: float = 0.07):
    """Several URDFs tile a slender link with many overlapping primitives (Go2 calf: three cylinders,
    `go2_description.urdf:149-185`).  Per link, keep spheres greedily by descending radius and drop any
    whose centre is closer than ``min_sep`` to an already kept one (explicit ``<sphere>`` primitives
    such as the four r = 5 mm contact points of a G1 foot, `g1_29dof_rev_1_0.urdf:262-283`, are
    never dropped against each other) - the lane program budgets a few collision spheres per link."""
    by_link: dict[int, list[int]] = {}
    for i, sph in enumerate(model.spheres):
        by_link.setdefault(model.bodies[sph.body].link, []).append(i)
    keep = set()
    for link, ids in by_link.items():
        kept: list[int] = []
        for i in sorted(ids, key=lambda i: -model.spheres[i].radius):
            c = model.spheres[i].center
            ex = model.spheres[i].explicit
            if all((ex and model.spheres[k].explicit) or np.linalg.norm(c - model.spheres[k].center) >= min_sep for k in kept):
                kept.append(i)
        keep.update(kept)
    model.spheres = [sph for i, sph in enumerate(model.spheres) if i in keep]


def _prune_contained_spheres(model: RobotModel):
    """Drop spheres that can never touch anything first:
    (a) fully inside another sphere rigidly attached to the same link;
    (b) centred on the link's own joint origin and fully inside a sphere of the parent link
        (pose independent, e.g. the top of the A1 thigh box inside the thigh_shoulder cylinder)."""
    keep = []
    for i, s in enumerate(model.spheres):
        li = model.bodies[s.body].link
        contained = False
        for k, t in enumerate(model.spheres):
            if k == i:
                continue
            lk = model.bodies[t.body].link
            if lk == li:
                c_t = t.center
                c_s = s.center
            elif lk == model.links[li].parent and np.linalg.norm(s.center) < 1e-9:
                c_t = t.center
                c_s = model.links[li].origin
            else:
                continue
            d = np.linalg.norm(c_s - c_t)
            if d + s.radius <= t.radius + 1e-12 and (s.radius < t.radius or i > k):
                contained = True
                break
        if not contained:
            keep.append(s)
    model.spheres = keep
