"""Self-collision of the articulations whose cfg enables it (`enabled_self_collisions=True`: assets/unitree.py:482 G1,
assets/roboparty.py:33 ATOM01): capsule proxies, explicit penalty forces between the listed capsule pairs
(robot_lab_amd/model/selfcol.py, oracle/physics.py, csrc/env_step.h self_place / self_apply).  CPU tier: the data, the
plausibility KAT the review asked for (an arm cannot pass through the torso), and the lane program against the oracle IN contact."""
import glob
import os

import numpy as np
import pytest

from helpers import assert_close, emu_load_state, host_view, make_pair, oracle_root_state
from oracle.env import OracleEnv
from oracle.physics import segment_closest
from robot_lab_amd.model.selfcol import segment_distance
from robot_lab_amd.scene import DATA_DIR, build_world, load_bundle

G1 = "RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0"


def list_bundles():
    return sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(DATA_DIR, "*.json")))


def capsule_gaps(ora, q=None):
    """[(gap, a, b)] of the model's pairs for the oracle env's current state (env 0)."""
    m = ora.desc.model
    Rw, ow, _ = ora.phys.kinematics(ora.st["root_pos"], ora.st["root_quat"], ora.st["q"] if q is None else q)
    seg = []
    for c in range(m.num_capsules):
        l = m.capsule_link[c]
        seg.append((ow[:, l] + Rw[:, l] @ np.array(m.capsule_p0[c][:]), ow[:, l] + Rw[:, l] @ np.array(m.capsule_p1[c][:]), m.capsule_radius[c]))
    out = []
    for p in range(m.num_self_pairs):
        a, b = m.self_pair[p][0], m.self_pair[p][1]
        xa, xb = segment_closest(seg[a][0], seg[a][1], seg[b][0], seg[b][1])
        out.append((float(np.linalg.norm(xa - xb, axis=-1)[0]) - seg[a][2] - seg[b][2], a, b))
    return out


def test_only_the_cfgs_that_enable_it_carry_pairs():
    on = {t for t in list_bundles() if load_bundle(t)[0].model.num_self_pairs > 0}
    assert on and all("Unitree-G1" in t or "ATOM01" in t for t in on), on
    assert {t for t in list_bundles() if "Unitree-G1" in t or "ATOM01" in t} == on
    for t in list_bundles():
        m = load_bundle(t)[0].model
        assert (m.self_collision == 1) == (t in on)


def test_capsules_and_pairs_of_g1():
    desc, extra = load_bundle(G1)
    m = desc.model
    assert 12 <= m.num_capsules <= 16 and 40 <= m.num_self_pairs <= 80 and desc.sim.self_k > 0
    links = [m.capsule_link[c] for c in range(m.num_capsules)]
    assert len(set(links)) == len(links) and 0 in links  # one capsule per link, the base among them
    names = {int(m.body_link[b]): desc.body_names[b] for b in reversed(range(m.num_bodies))}
    have = {names[l] for l in links}
    for must in ("torso_link", "left_knee_link", "right_knee_link", "left_elbow_link", "right_wrist_yaw_link", "left_ankle_roll_link"):
        assert must in have, (must, have)
    for p in range(m.num_self_pairs):
        a, b = m.self_pair[p][0], m.self_pair[p][1]
        la, lb = links[a], links[b]
        assert a < b and m.link_parent[la] != lb and m.link_parent[lb] != la  # never parent / child
    # nothing pushes in the default pose: every listed pair starts at least a centimetre apart
    h, to, eo = build_world(desc, extra, 1, 0)
    ora = OracleEnv(desc, h, to, 1, 1, eo)
    ora.reset()
    q0 = np.array(m.default_joint_pos[: m.num_dof], dtype=np.float64)[None]
    assert min(g for g, _, _ in capsule_gaps(ora, q0)) >= 0.01 - 1e-6


def test_atom01_pairs_start_apart_and_its_lane_program_matches_the_oracle(emu_lib):
    """The other cfg with self-collisions on (assets/roboparty.py:33): its pair list obeys the same rules, and with the left arm rolled
    1.2 rad inwards - the arm pressed against the base capsule - the lane program and the oracle agree with the pass running."""
    task = "RobotLab-Isaac-Velocity-Flat-RoboParty-ATOM01-v0"
    desc, ora, nat = make_pair(task, 4, 3, emu_lib)
    m = desc.model
    assert m.self_collision == 1 and 10 <= m.num_capsules <= 16 and 20 <= m.num_self_pairs <= 80
    ora.reset()
    nat.reset()
    q0 = np.array(m.default_joint_pos[: m.num_dof], dtype=np.float64)[None].repeat(4, 0)
    assert min(g for g, _, _ in capsule_gaps(ora, q0)) >= 0.01 - 1e-6
    j = list(desc.joint_names).index("left_arm_roll_joint")
    a = np.zeros((4, m.num_dof), dtype=np.float32)
    a[:, j] = -1.2 / m.action_scale[j]
    closest = np.inf
    for _ in range(12):
        ora.step(a)
        nat.step(a.ctypes.data)
        closest = min(closest, min(g for g, _, _ in capsule_gaps(ora)))
    nat.export_state()
    assert -0.04 < closest < -0.002, closest  # touched, and stopped
    assert_close("root", host_view(nat, "ROOT_STATE"), oracle_root_state(ora), 2e-3, 2e-4)
    assert_close("qd", host_view(nat, "JOINT_VEL"), ora.st["qd"], 5e-3, 5e-3)
    nat.close()


def test_segment_closest_points():
    rng = np.random.default_rng(0)
    for i in range(400):
        pts = rng.normal(size=(4, 3))
        if i % 5 == 0:
            pts[1] = pts[0]  # a is a point
        if i % 7 == 0:
            pts[3] = pts[2]  # b is a point
        if i % 11 == 0:
            pts[3] = pts[2] + (pts[1] - pts[0]) * rng.random()  # parallel
        d, _, _ = segment_distance(*pts)
        xa, xb = segment_closest(*[p[None] for p in pts])
        assert abs(np.linalg.norm(xa - xb) - d) < 1e-12
        if i % 20 == 0:  # against a brute-force scan of both parameters
            S, T = np.meshgrid(np.linspace(0, 1, 201), np.linspace(0, 1, 201))
            A = pts[0] + S[..., None] * (pts[1] - pts[0])
            B = pts[2] + T[..., None] * (pts[3] - pts[2])
            bf = np.linalg.norm(A - B, axis=-1).min()
            assert bf - 0.02 <= d <= bf + 1e-9


def _drive_arm_into_torso(with_pairs, steps=40):
    """Left shoulder roll commanded 1.2 rad inwards of its default: the upper arm is pushed against the torso.  Returns the
    smallest torso / upper-arm gap seen and the oracle env (in contact at the end when the pairs are on)."""
    desc, extra = load_bundle(G1)
    if not with_pairs:
        desc.model.num_self_pairs = 0
    m = desc.model
    h, to, eo = build_world(desc, extra, 1, 0)
    ora = OracleEnv(desc, h, to, 1, 7, eo)
    ora.reset()
    jn = list(desc.joint_names)
    j = jn.index("left_shoulder_roll_joint")
    a = np.zeros((1, m.num_dof), dtype=np.float32)
    a[0, j] = -1.2 / m.action_scale[j]
    names = {int(m.body_link[b]): desc.body_names[b] for b in reversed(range(m.num_bodies))}
    cap_of = {names[m.capsule_link[c]]: c for c in range(m.num_capsules)}
    pair = (cap_of["torso_link"], cap_of["left_shoulder_yaw_link"])
    desc2, _ = load_bundle(G1)  # the gap is always measured with the full pair list
    probe = OracleEnv(desc2, h, to, 1, 7, eo)
    probe.reset()
    worst = np.inf
    for _ in range(steps):
        ora.step(a)
        probe.st = ora.st
        worst = min(worst, min(g for g, x, y in capsule_gaps(probe) if (x, y) == pair))
    return worst, ora, a


def test_an_arm_does_not_pass_through_the_torso():
    """Plausibility KAT: with the pairs the upper arm stops at the torso (a few centimetres of the soft penalty contact),
    without them the same command drives it more than 10 cm into the torso capsule."""
    with_pairs, ora, _ = _drive_arm_into_torso(True)
    without, _, _ = _drive_arm_into_torso(False)
    assert without < -0.10, without
    assert with_pairs > -0.04, with_pairs
    assert np.isfinite(ora.st["qd"]).all() and np.abs(ora.st["qd"]).max() < 30.0  # an explicit spring: it must not ring up


@pytest.mark.parametrize("sub", ["1", "4", "8"])
def test_lane_program_matches_oracle_in_self_contact(sub, emu_lib, monkeypatch):
    """Both sides step from a state in which the upper arm presses on the torso (and the push goes on): same end state.
    (sub 8: the 32-lane mapping - the first four sub-lanes of a limb play the 16 virtual lanes the pairs are dealt to.)"""
    monkeypatch.setenv("RL_EMU_SUB", sub)
    if sub == "8":
        monkeypatch.setenv("RL_EMU_FIBERS", "1")
    _, src, a = _drive_arm_into_torso(True, steps=25)
    state = src.read_state()
    gaps0 = capsule_gaps(src)
    assert min(g for g, _, _ in gaps0) < -0.005  # in contact
    desc, ora, nat = make_pair(G1, 1, 7, emu_lib)
    ora.reset()
    nat.reset()
    ora.load_state(state)
    emu_load_state(nat, state)
    for _ in range(2):
        ora.step(a)
        nat.step(a.ctypes.data)
    nat.export_state()
    assert_close("root", host_view(nat, "ROOT_STATE"), oracle_root_state(ora), 1e-3, 1e-4)
    assert_close("q", host_view(nat, "JOINT_POS"), ora.st["q"], 1e-3, 1e-4)
    assert_close("qd", host_view(nat, "JOINT_VEL"), ora.st["qd"], 3e-3, 3e-3)
    # and the pass really acted: without the pairs the same two steps end elsewhere
    desc0, extra0 = load_bundle(G1)
    desc0.model.num_self_pairs = 0
    h, to, eo = build_world(desc0, extra0, 1, 0)
    off = OracleEnv(desc0, h, to, 1, 7, eo)
    off.reset()
    off.load_state(state)
    for _ in range(2):
        off.step(a)
    assert np.abs(off.st["qd"] - ora.st["qd"]).max() > 0.05
    nat.close()
