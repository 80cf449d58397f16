"""Known-answer facts derivable from the reference (SURVEY.md section 8(c) "KATs") checked on the committed
descriptor bundles, and - where /root/reference exists - that recompiling the reference's cfg classes
through the shims reproduces the committed bundles."""
import json
import os

import numpy as np
import pytest

from robot_lab_amd.desc import arr, desc_to_json
from robot_lab_amd.model.build import max_episode_length
from robot_lab_amd.scene import load_bundle

REF = "/root/reference/source/robot_lab"
A1R = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"


def test_a1_census():
    d, _ = load_bundle(A1R)
    m = d.model
    assert (m.num_links, m.num_dof, m.num_bodies, m.num_chains, m.chain_len) == (13, 12, 17, 4, 3)
    assert abs(float(arr(m.body_mass, 17).sum()) - 13.741) < 1e-4                     # a1.urdf masses
    assert d.joint_names[:3] == ["FR_hip_joint", "FR_thigh_joint", "FR_calf_joint"]    # rough_env_cfg.py:22-27
    q0 = arr(m.default_joint_pos, 12)
    np.testing.assert_allclose(q0.reshape(4, 3), np.tile([0.0, 0.8, -1.5], (4, 1)), atol=1e-6)  # unitree.py:43-50
    np.testing.assert_allclose(arr(m.default_root_pos), [0, 0, 0.38], atol=1e-6)
    lo, hi = arr(m.joint_lower, 12).reshape(4, 3), arr(m.joint_upper, 12).reshape(4, 3)
    np.testing.assert_allclose(lo[0], [-0.802851, -1.047198, -2.696534], atol=1e-5)   # a1.urdf:369,411,439
    np.testing.assert_allclose(hi[0], [0.802851, 4.188790, -0.916298], atol=1e-5)
    mid, rng = 0.5 * (lo + hi), hi - lo
    np.testing.assert_allclose(arr(m.soft_lower, 12).reshape(4, 3), mid - 0.45 * rng, atol=1e-5)  # soft factor 0.9, unitree.py:53
    np.testing.assert_allclose(arr(m.action_scale, 12).reshape(4, 3), np.tile([0.125, 0.25, 0.25], (4, 1)))  # rough_env_cfg.py:51
    assert max_episode_length(d) == 1000                                               # ceil(20 / 0.02)
    assert (d.obs_dim(0), d.obs_dim(1)) == (45, 235) and d.task.scan_nx * d.task.scan_ny == 187
    assert d.task.n_rewards == 17 and not d.task.term_illegal_contact                  # rough_env_cfg.py:83-153
    w = {n: d.task.rewards[i].weight for i, n in enumerate(d.reward_names)}
    assert w["track_lin_vel_xy_exp"] == 3.0 and w["feet_height_body"] == -5.0 and abs(w["joint_torques_l2"] + 2.5e-5) < 1e-12
    feet = [i for i, n in enumerate(d.body_names) if n.endswith("_foot")]
    r = d.task.rewards[d.reward_names.index("undesired_contacts")]
    assert all(not (r.body_mask >> f) & 1 for f in feet) and bin(r.body_mask).count("1") == 13  # "^(?!.*_foot).*"


@pytest.mark.parametrize("task,dims,bodies,mass", [
    ("RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0", (45, 48), 17, 13.741),
    ("RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0", (45, 235), 19, 16.087),
    ("RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0", (57, 247), 19, 19.523),
    ("RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0", (96, 286), 33, 33.341),
    ("RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0", (96, 99), 33, 33.341),
    ("RobotLab-Isaac-Velocity-Rough-Unitree-B2-v0", (45, 235), 17, None),
    ("RobotLab-Isaac-Velocity-Rough-Deeprobotics-Lite3-v0", (45, 235), 17, None),
    ("RobotLab-Isaac-Velocity-Rough-Deeprobotics-M20-v0", (57, 247), 17, None),
    ("RobotLab-Isaac-Velocity-Rough-Zsibot-ZSL1-v0", (45, 235), 17, None),
    ("RobotLab-Isaac-Velocity-Flat-Zsibot-ZSL1W-v0", (57, 60), 17, None),
    ("RobotLab-Isaac-Velocity-Rough-RoboParty-ATOM01-v0", (78, 268), 24, None),
    ("RobotLab-Isaac-Velocity-Rough-RobotEra-Xbot-v0", (93, 283), 29, None),
    ("RobotLab-Isaac-Velocity-Flat-MagicLab-Bot-Gen1-v0", (51, 54), 15, None),
    ("RobotLab-Isaac-Velocity-Rough-Openloong-Loong-v0", (45, 235), 13, None),
])
def test_other_bundles(task, dims, bodies, mass):
    d, _ = load_bundle(task)
    assert (d.obs_dim(0), d.obs_dim(1)) == dims and d.model.num_bodies == bodies
    if mass is not None:
        assert abs(float(arr(d.model.body_mass, bodies).sum()) - mass) < 2e-3
    assert d.model.num_chains == 4 and 3 <= d.model.chain_len <= 7


def test_gr1_is_a_six_joint_spine_with_the_arms_leaving_it_at_the_torso():
    """FFTAI GR1T1 (`GR1T1.urdf`): legs on the base; waist yaw / pitch / roll; on the waist's last link the head's three joints AND the
    two 7-joint arms - five candidate limbs.  The builder lets the trunk continue into the shortest terminal child chain (the head):
    trunk = waist + head, the arms attach at depth 3 (model/build.py), which the trunk + limbs program already supports."""
    d, _ = load_bundle("RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0")
    m = d.model
    assert (m.num_dof, m.num_links, m.num_chains, m.chain_len, m.num_trunk) == (32, 33, 4, 7, 6)
    assert [d.joint_names[m.trunk_link[i] - 1] for i in range(6)] == ["waist_yaw", "waist_pitch", "waist_roll", "head_yaw", "head_roll", "head_pitch"]
    assert sorted(m.chain_nj) == [6, 6, 7, 7]
    arms = [k for k in range(4) if m.chain_nj[k] == 7]
    legs = [k for k in range(4) if m.chain_nj[k] == 6]
    assert all(m.chain_attach[k] == 3 for k in arms) and all(m.chain_attach[k] == 0 for k in legs)
    assert all("shoulder_pitch" in d.joint_names[m.chain_link[k][0] - 1] for k in arms)
    # the spine links nothing hangs off (inner waist links, head) carry no collision spheres: no lane could host them
    hosted = {0, m.trunk_link[2]}
    spine = {m.trunk_link[i] for i in range(6)}
    for g in range(m.num_spheres):
        link = m.body_link[m.sphere_body[g]]
        assert link not in spine or link in hosted
    assert m.self_collision == 0 and m.num_self_pairs == 0  # assets/fftai.py:41: enabled_self_collisions=False


def test_g1_census():
    """Known-answer facts of SURVEY.md 8(c): 29 joints, trunk = 3 waist joints carrying the arms, per-joint
    action scale 0.25 * effort / stiffness (unitree.py:625-636), torso-mounted events and scanner."""
    d, _ = load_bundle("RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0")
    m, t = d.model, d.task
    assert (m.num_dof, m.num_links, m.num_chains, m.chain_len, m.num_trunk) == (29, 30, 4, 7, 3)
    assert list(m.chain_nj) == [6, 6, 7, 7] and list(m.chain_attach) == [0, 0, 3, 3]
    assert [d.joint_names[m.trunk_link[i] - 1] for i in range(3)] == ["waist_yaw_joint", "waist_roll_joint", "waist_pitch_joint"]
    j = d.joint_names.index("left_hip_pitch_joint")
    assert abs(m.action_scale[j] - 0.25 * 88.0 / (0.010177520 * (20 * 3.1415926535) ** 2)) < 1e-6  # = 0.5475
    assert abs(m.default_root_pos[2] - 0.76) < 1e-6 and abs(m.default_joint_pos[d.joint_names.index("left_knee_joint")] - 0.669) < 1e-6
    assert d.body_names[t.base_body] == "torso_link" and d.body_names[t.scan_body] == "torso_link"
    assert t.term_illegal_contact == 1 and t.illegal_body_mask == 1 << d.body_names.index("torso_link")
    feet = [d.body_names.index(n) for n in ("left_ankle_roll_link", "right_ankle_roll_link")]
    n_foot_spheres = sum(1 for g in range(m.num_spheres) if m.sphere_body[g] in feet)
    assert n_foot_spheres == 8  # 4 contact points per foot (g1_29dof_rev_1_0.urdf:262-283)
    kinds = {d.reward_names[i]: t.rewards[i].kind for i in range(t.n_rewards)}
    from robot_lab_amd.desc import REW
    assert kinds["track_lin_vel_xy_exp"] == REW["track_lin_vel_xy_yaw_frame_exp"] and kinds["feet_air_time"] == REW["feet_air_time_positive_biped"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")
def test_bundles_recompile_from_reference_cfg():
    from robot_lab_amd import shims

    shims.install(shims.REFERENCE_SOURCE)
    import gymnasium as gym
    import robot_lab.tasks  # noqa: F401
    from isaaclab_tasks.utils import parse_env_cfg

    from robot_lab_amd.model.cfg_compile import compile_cfg

    assert gym.spec(A1R).entry_point == "isaaclab.envs:ManagerBasedRLEnv"
    for task in (A1R, "RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0", "RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0",
                 "RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0"):
        desc, spec = compile_cfg(parse_env_cfg(task, device="cpu", num_envs=8))
        committed, _ = load_bundle(task)
        assert json.loads(desc_to_json(desc)) == json.loads(desc_to_json(committed)), task


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")
def test_command_level_curricula_compile_into_the_descriptor():
    """The base cfg's command_levels_* terms (velocity_env_cfg.py:673-690), which every shipped robot cfg deletes, put back the way
    the commented-out lines of the robot cfgs would (unitree_g1/rough_env_cfg.py:159-160)."""
    from robot_lab_amd import shims

    shims.install(shims.REFERENCE_SOURCE)
    import robot_lab.tasks  # noqa: F401
    from isaaclab_tasks.utils import parse_env_cfg
    from robot_lab.tasks.manager_based.locomotion.velocity.velocity_env_cfg import CurriculumCfg

    from robot_lab_amd.model.cfg_compile import UnsupportedTerm, compile_cfg

    cfg = parse_env_cfg(A1R, device="cpu", num_envs=8)
    assert cfg.curriculum.command_levels_lin_vel is None
    base = CurriculumCfg()
    cfg.curriculum.command_levels_lin_vel = base.command_levels_lin_vel
    cfg.curriculum.command_levels_ang_vel = base.command_levels_ang_vel
    cfg.curriculum.command_levels_ang_vel.params["range_multiplier"] = (0.2, 1.0)
    desc, _ = compile_cfg(cfg)
    t, names = desc.task, list(desc.reward_names)
    assert t.cur_cmd_lin == 1 and t.cur_cmd_ang == 1
    assert names[t.cur_cmd_lin_term] == "track_lin_vel_xy_exp" and names[t.cur_cmd_ang_term] == "track_ang_vel_z_exp"
    assert tuple(t.cur_cmd_lin_mult) == pytest.approx((0.1, 1.0)) and tuple(t.cur_cmd_ang_mult) == pytest.approx((0.2, 1.0))
    cfg.curriculum.command_levels_lin_vel.params["reward_term_name"] = "feet_gait"  # weight 0 in the A1 cfg: not an active term
    with pytest.raises(UnsupportedTerm):
        compile_cfg(cfg)


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs /root/reference")
def test_entry_point_resolves_to_the_hip_env():
    from robot_lab_amd import shims

    shims.install(shims.REFERENCE_SOURCE)
    import isaaclab.envs

    from robot_lab_amd.env import ManagerBasedRLEnv

    assert isaaclab.envs.ManagerBasedRLEnv is ManagerBasedRLEnv


def test_mesh_vertex_readers(tmp_path):
    """COLLADA / OBJ collision meshes (B2W calves and wheels, four Xbot links): vertex clouds as the sphere fit consumes them.  A
    COLLADA geometry is carried through the matrices of the scene nodes that instance it (Blender: millimetres + a 0.001 scale
    and a rotation in the node) and the asset unit; the wheel-like cloud below becomes ONE sphere of the wheel radius."""
    from robot_lab_amd.model.urdf import _mesh_to_spheres, _read_mesh_vertices

    ang = np.linspace(0, 2 * np.pi, 48, endpoint=False)
    rim = np.concatenate([np.stack([100 * np.cos(ang), 100 * np.sin(ang), np.full_like(ang, z)], 1) for z in (-20.0, 20.0)])  # mm, axis z
    dae = tmp_path / "wheel.dae"
    dae.write_text(f"""<?xml version="1.0"?>
<COLLADA xmlns="http://www.collada.org/2005/11/COLLADASchema" version="1.4.1">
 <asset><unit name="meter" meter="1"/><up_axis>Z_UP</up_axis></asset>
 <library_geometries><geometry id="g"><mesh>
  <source id="g-pos"><float_array id="g-pos-array" count="{rim.size}">{' '.join(f'{v:.6f}' for v in rim.ravel())}</float_array></source>
  <source id="g-nrm"><float_array id="g-nrm-array" count="3">0 0 1</float_array></source>
  <vertices id="g-vtx"><input semantic="POSITION" source="#g-pos"/></vertices>
 </mesh></geometry></library_geometries>
 <library_visual_scenes><visual_scene id="s"><node id="n"><matrix sid="transform">0.001 0 0 0  0 0 -0.001 0.05  0 0.001 0 0  0 0 0 1</matrix>
  <instance_geometry url="#g"/></node></visual_scene></library_visual_scenes>
</COLLADA>""")
    v = _read_mesh_vertices(str(dae))
    assert v.shape == (96, 3)
    np.testing.assert_allclose(v.min(0), [-0.1, 0.03, -0.1], atol=2e-3)   # rotated about x (wheel axis now y), shifted 0.05 in y, metres
    np.testing.assert_allclose(v.max(0), [0.1, 0.07, 0.1], atol=2e-3)
    sph = _mesh_to_spheres(v, np.ones(3))
    assert len(sph) == 1 and abs(sph[0][1] - 0.1) < 5e-3 and np.allclose(sph[0][0], [0, 0.05, 0], atol=2e-3)
    # the same wheel through <scale> / <rotate> / <translate> node elements (document order) in a Y_UP file: the reader composes them
    # and turns the result into the Z_UP convention (other exporters than Blender's matrix form)
    dae2 = tmp_path / "wheel_yup.dae"
    dae2.write_text(dae.read_text().replace("<up_axis>Z_UP</up_axis>", "<up_axis>Y_UP</up_axis>").replace(
        '<matrix sid="transform">0.001 0 0 0  0 0 -0.001 0.05  0 0.001 0 0  0 0 0 1</matrix>',
        "<translate>0 0.05 0</translate><rotate>1 0 0 90</rotate><scale>0.001 0.001 0.001</scale>"))
    v2 = _read_mesh_vertices(str(dae2))
    # node transform = T R S = the matrix above (wheel axis y, centre y = 0.05 in the file's frame); Y_UP -> Z_UP maps (x, y, z) to (x, -z, y)
    np.testing.assert_allclose(v2, np.stack([v[:, 0], -v[:, 2], v[:, 1]], 1), atol=1e-12)
    bad = tmp_path / "bad.dae"
    bad.write_text(dae.read_text().replace('<matrix sid="transform">', '<lookat>0 0 0 1 1 1 0 0 1</lookat><matrix sid="transform">'))
    with pytest.warns(UserWarning, match="could not be read"):
        assert _read_mesh_vertices(str(bad)).shape == (0, 3) and _mesh_to_spheres(_read_mesh_vertices(str(bad)), np.ones(3)) == []
    trunc = tmp_path / "trunc.dae"
    trunc.write_text(dae.read_text()[:400])
    with pytest.warns(UserWarning, match="could not be read"):
        assert _read_mesh_vertices(str(trunc)).shape == (0, 3)
    obj = tmp_path / "box.obj"
    obj.write_text("# box\n" + "".join(f"v {x} {y} {z}\n" for x in (0, 1) for y in (0, 2) for z in (0, 3)) + "vn 0 0 1\nf 1 2 3\n")
    assert _read_mesh_vertices(str(obj)).shape == (8, 3) and _read_mesh_vertices(str(obj)).max() == 3.0


def test_no_bundle_freezes_a_joint():
    """A URDF velocity limit of 0 (CAD exporters write it for "not specified": agibot/d1/urdf/edu.urdf:99-103) must not reach the
    descriptor: the solver would compute joint velocities and the integrator would clamp them to zero - energy from nowhere (Agibot D1
    diverged within ~100 steps until round 3).  Likewise no massless robot and no empty limit interval."""
    import glob

    from robot_lab_amd.scene import DATA_DIR, load_bundle

    for path in sorted(glob.glob(os.path.join(DATA_DIR, "*.json"))):
        m = load_bundle(os.path.basename(path)[:-5])[0].model
        D = m.num_dof
        vl = np.array([m.joint_vel_limit[i] for i in range(D)])
        assert (vl > 0.1).all(), (path, vl)
        assert all(m.joint_upper[i] > m.joint_lower[i] for i in range(D)), path
        assert sum(m.body_mass[i] for i in range(m.num_bodies)) > 1.0, path


# ------------------------------------------------------------------------------------------------------------------------------------
# spine links nothing hangs off: hosted where a lane can be spared, dropped LOUDLY otherwise
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference's cfg classes and URDFs")
def test_a_spine_link_without_a_limb_is_hosted_or_dropped_loudly(monkeypatch):
    """FFTAI GR1: waist(3) + head(3) spine, legs on the base, arms on the torso (depth 3).  The shipped URDF puts collision geometry on
    the torso only.  Give the HEAD and an INNER WAIST link a sphere: the head (deepest) takes the group 0 of the second arm lane (the
    torso's one sphere fits the first), the waist link the second leg lane's (the base has no spheres to host) - nothing is dropped.
    With a third orphan (waist_pitch) there is no lane left: its body is dropped with a warning and listed in the bundle."""
    import warnings

    import numpy as np

    from robot_lab_amd import shims

    shims.install(shims.REFERENCE_SOURCE)
    import robot_lab.tasks  # noqa: F401
    from isaaclab_tasks.utils import parse_env_cfg

    import robot_lab_amd.model.cfg_compile as cc
    from robot_lab_amd.model.urdf import Sphere

    task = "RobotLab-Isaac-Velocity-Flat-FFTAI-GR1T1-v0"
    real_load = cc.load_urdf

    def with_orphans(names):
        def load(path, **kw):
            model = real_load(path, **kw)
            for n in names:
                b = model.body_names.index(n)
                model.spheres.append(Sphere(body=b, center=np.zeros(3), radius=0.08))
            return model
        return load

    monkeypatch.setattr(cc, "load_urdf", with_orphans(["head_pitch", "waist_yaw"]))
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # nothing may be dropped
        desc, spec = cc.compile_cfg(parse_env_cfg(task, device="cpu", num_envs=4))
    m = desc.model
    depth = {int(m.trunk_link[i]): i + 1 for i in range(m.num_trunk)}
    head_link, waist_link = int(m.body_link[desc.body_names.index("head_pitch")]), int(m.body_link[desc.body_names.index("waist_yaw")])
    assert list(m.chain_attach) == [0, 0, 3, 3]
    assert list(m.chain_grp0) == [0, depth[waist_link] + 1, 0, depth[head_link] + 1] and "dropped_contact_bodies" not in spec
    hosted = {desc.body_names[m.sphere_body[g]] for g in range(m.num_spheres)}
    assert {"head_pitch", "waist_yaw", "waist_roll"} <= hosted

    monkeypatch.setattr(cc, "load_urdf", with_orphans(["head_pitch", "waist_yaw", "waist_pitch"]))
    with pytest.warns(UserWarning, match="cannot touch the ground"):
        desc, spec = cc.compile_cfg(parse_env_cfg(task, device="cpu", num_envs=4))
    assert spec["dropped_contact_bodies"] == ["waist_yaw"]  # deepest first: head, then waist_pitch get the two spare lanes
    assert "waist_yaw" not in {desc.body_names[desc.model.sphere_body[g]] for g in range(desc.model.num_spheres)}
    # the shipped robots: nothing to host, nothing dropped
    monkeypatch.setattr(cc, "load_urdf", real_load)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        desc, spec = cc.compile_cfg(parse_env_cfg(task, device="cpu", num_envs=4))
    assert list(desc.model.chain_grp0) == [0, 0, 0, 0] and "dropped_contact_bodies" not in spec


def test_booster_t1_is_a_trunk_of_two_pieces():
    """`assets/booster.py:10-109`, `t1_description/urdf/robot.urdf`: a two-joint neck, two four-joint arms and a one-joint waist on the
    trunk body, the six-joint legs behind the waist - five chains on the base.  The neck becomes a second piece of the trunk (joints all
    lanes simulate redundantly, restarting at the base), the arms hang off the base, the legs off the waist; the head's collision sphere
    rides in the group 0 of a leg lane (the waist link, which both leg lanes would otherwise share, has no geometry)."""
    d, _ = load_bundle("RobotLab-Isaac-Velocity-Rough-Booster-T1-v0")
    m = d.model
    assert (m.num_links, m.num_dof, m.num_chains, m.chain_len, m.num_trunk) == (24, 23, 4, 6, 3)
    trunk_joints = [d.joint_names[m.trunk_link[i] - 1] for i in range(m.num_trunk)]
    assert trunk_joints == ["Waist", "AAHead_yaw", "Head_pitch"] and list(m.trunk_parent)[:3] == [0, -1, 0]
    assert sorted(zip(m.chain_nj, m.chain_attach)) == [(4, 0), (4, 0), (6, 1), (6, 1)]
    legs = [k for k in range(4) if m.chain_nj[k] == 6]
    assert sorted(m.chain_grp0[k] for k in legs) == [0, 4] and all(m.chain_grp0[k] == 0 for k in range(4) if k not in legs)  # depth 3 = the head link
    hosted = {d.body_names[m.sphere_body[g]] for g in range(m.num_spheres)}
    assert {"Trunk", "H2", "left_foot_link", "right_foot_link"} <= hosted
    # the head joints have no actuator group (assets/booster.py: legs / feet / arms): passive hinges
    for n in ("AAHead_yaw", "Head_pitch"):
        j = d.joint_names.index(n)
        assert m.act_kp[j] == 0.0 and m.act_kd[j] == 0.0


def test_feet_sit_on_limbs_and_the_topology_rule_is_recorded():
    """topology discovery tries two rules and takes the first decomposition that fits; the choice is recorded with the bundle
    (`topology`: rule, trunk links, pieces, limbs), and whatever rule won, a body a FEET term names - the contact-sensor feet of the
    task's reward terms - rides on a limb chain, never on a trunk link (a leg absorbed into the spine would be simulated redundantly by
    every lane and its foot contacts hosted by a borrowed lane)."""
    import glob
    import json

    from robot_lab_amd.desc import REW
    from robot_lab_amd.scene import DATA_DIR, load_bundle

    feet_kinds = {REW[k] for k in ("feet_air_time", "feet_air_time_variance_penalty", "feet_slide", "feet_contact_without_cmd", "feet_height_body",
                                   "feet_air_time_positive_biped", "feet_stumble", "feet_contact", "feet_height", "contact_forces")}
    seen_rules = set()
    for path in sorted(glob.glob(os.path.join(DATA_DIR, "*.json"))):
        blob = json.load(open(path))
        topo = blob.get("topology")
        assert topo and topo["rule"] in ("continue_spine", "branching_trunk"), path
        seen_rules.add(topo["rule"])
        desc, _ = load_bundle(path)
        m, t = desc.model, desc.task
        assert len(topo["trunk_links"]) == m.num_trunk and len(topo["limb_lengths"]) <= 4
        limb_links = {m.chain_link[k][j] for k in range(4) for j in range(m.chain_nj[k])}
        for i in range(t.n_rewards):
            r = t.rewards[i]
            if r.kind not in feet_kinds:
                continue
            for b in range(m.num_bodies):
                if (r.body_mask >> b) & 1:
                    assert m.body_link[b] in limb_links, (os.path.basename(path), desc.reward_names[i], desc.body_names[b])
    assert seen_rules == {"continue_spine", "branching_trunk"}
