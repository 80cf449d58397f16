"""Oracle term restatements vs the REFERENCE's own functions.  tests/golden/terms_*.npz hold the outputs
of `VEL/mdp/rewards.py` / `commands.py` / `events.py` (imported unchanged from /root/reference by
tools/gen_golden_terms.py) on a recorded simulator state; here the oracle is put into the same state and
must reproduce them (fp64 arithmetic on both sides; the descriptor stores term parameters as fp32, hence rtol 2e-6)."""
import os

import numpy as np
import pytest

from oracle import spatial as sp
from oracle.env import OracleEnv
from robot_lab_amd.desc import arr
from robot_lab_amd.scene import build_world, load_bundle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(robot):
    g = np.load(os.path.join(GOLD, f"terms_{robot}.npz"))
    task = str(g["task"])
    desc, extra = load_bundle(task)
    N = int(g["N"])
    h, to, eo = build_world(desc, extra, N, 0)
    ora = OracleEnv(desc, h, to, N, int(g["seed"]), eo)
    for k in ("root_pos", "root_quat", "root_lin_vel", "root_ang_vel", "q", "qd", "base_com", "root_com"):
        if "st_" + k in g:
            ora.st[k] = g["st_" + k].copy()
    for k in ("applied_torque", "joint_acc", "force_hist", "contact_force", "timers", "action", "prev_action", "vel_command_b", "terminated"):
        setattr(ora, k, g[k].copy())
    return g, desc, ora


@pytest.mark.parametrize("robot", ["a1", "go2", "go2w", "g1", "a1_handstand", "tita"])
def test_reward_terms_match_reference_functions(robot):
    g, desc, ora = _load(robot)
    hist = np.linalg.norm(ora.force_hist, axis=-1).max(axis=1)
    ora.compute_rewards(ora.derived(), hist)
    names = [str(n) for n in g["term_names"]]
    assert names == list(desc.reward_names)
    for i, name in enumerate(names):
        w = float(desc.task.rewards[i].weight)
        got = ora.reward_terms[i] / (w * ora.step_dt)
        np.testing.assert_allclose(got, g["term_values"][i], rtol=2e-6, atol=1e-7, err_msg=name)  # descriptor parameters are fp32
    # at least the contact / timer driven terms must be exercised by the recorded state
    exercised = {"g1": ("track_lin_vel_xy_exp", "track_ang_vel_z_exp", "feet_air_time", "feet_slide", "joint_deviation_arms_l1", "flat_orientation_l2"),
                 # config/others/unitree_a1_handstand/env/rewards.py:18-59
                 "a1_handstand": ("handstand_feet_height_exp", "handstand_feet_on_air", "handstand_feet_air_time", "handstand_orientation_l2"),
                 # rewards.py:616-644 (ray caster branch), 132-153, 439-461
                 "tita": ("base_height_l2", "wheel_vel_penalty", "feet_distance_y_exp", "feet_slide", "contact_forces"),
                 # wheeled/unitree_go2w/rough_env_cfg.py:149-171: its own term list, incl. the wheel-joint acceleration term
                 "go2w": ("undesired_contacts", "contact_forces", "joint_mirror", "joint_acc_wheel_l2", "joint_pos_penalty")}
    for name in exercised.get(robot, ("undesired_contacts", "contact_forces", "feet_height_body", "joint_mirror")):
        assert np.abs(g["term_values"][names.index(name)]).max() > 0, name


@pytest.mark.parametrize("robot", ["a1", "go2"])
def test_command_threshold_rule(robot):
    """UniformThresholdVelocityCommand._resample_command (VEL/mdp/commands.py:43-47)."""
    g, desc, ora = _load(robot)
    cmd = g["cmd_in"].copy()
    cmd[:, :2] *= (np.linalg.norm(cmd[:, :2], axis=1) > desc.task.cmd_small_threshold)[:, None]
    np.testing.assert_allclose(cmd, g["cmd_out"], rtol=0, atol=0)
    assert (g["cmd_out"][:, :2] == 0).all(axis=1).any()


@pytest.mark.parametrize("robot", ["a1", "go2", "g1"])
def test_reset_root_state_uniform(robot):
    """VEL/mdp/events.py:205-271 with injected uniform samples == the oracle's reset arithmetic."""
    g, desc, ora = _load(robot)
    m = desc.model
    ps, vs = g["reset_pose_samples"], g["reset_vel_samples"]
    pos = arr(m.default_root_pos)[None] + g["env_origins"] + ps[:, :3]
    quat = sp.quat_mul(np.tile(arr(m.default_root_quat).astype(np.float64), (len(ps), 1)), sp.quat_from_euler_xyz(ps[:, 3], ps[:, 4], ps[:, 5]))
    np.testing.assert_allclose(np.concatenate([pos, quat], -1), g["reset_pose"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(vs, g["reset_vel"], rtol=0, atol=0)


@pytest.mark.parametrize("robot", ["a1", "go2", "go2w", "g1", "tita"])
def test_observation_rows_match_reference_cfg(robot):
    """The two observation groups against rows built from the REFERENCE's ObservationsCfg (VEL/velocity_env_cfg.py:134-254 and the
    robot's overrides) term by term in declaration order - its own `joint_pos_rel_without_wheel` (VEL/mdp/observations.py:17-27)
    where the cfg names it (Go2W, Tita), clip, scale, concatenation [UPSTREAM B2].  Noise off (a random draw: the Philox tests)."""
    g, desc, ora = _load(robot)
    desc.task.policy_corrupt = 0
    desc.task.critic_corrupt = 0
    pol, cri = ora.compute_observations()
    np.testing.assert_allclose(pol, g["obs_policy"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(cri, g["obs_critic"], rtol=1e-6, atol=1e-7)
    # the compiled descriptor lists the same terms in the same order with the same widths
    from robot_lab_amd.desc import OBS_KINDS

    for grp, terms, n in (("policy", desc.task.policy, desc.task.n_policy), ("critic", desc.task.critic, desc.task.n_critic)):
        widths = [int(w) for w in g[f"obs_{grp}_widths"]]
        assert n == len(widths)
        D, scan = desc.model.num_dof, desc.task.scan_nx * desc.task.scan_ny
        mine = [scan if OBS_KINDS[terms[i].kind] == "height_scan" else (D if "joint" in OBS_KINDS[terms[i].kind] or OBS_KINDS[terms[i].kind] == "last_action" else 3) for i in range(n)]
        assert mine == widths, (grp, [str(x) for x in g[f"obs_{grp}_terms"]])
    if robot in ("go2w", "tita"):  # the wheel joints' columns of joint_pos are zero, the others are not
        names = [str(x) for x in g["obs_policy_terms"]]
        off = int(np.sum(g["obs_policy_widths"][: names.index("joint_pos")]))
        wheels = [i for i in range(desc.model.num_dof) if (desc.task.wheel_joint_mask >> i) & 1]
        assert wheels and np.all(pol[:, [off + i for i in wheels]] == 0)
        assert np.abs(pol[:, off:off + desc.model.num_dof]).max() > 0


def test_reward_functions_without_a_cfg_match_reference():
    """`feet_contact` (rewards.py:399-413), `feet_height` (rewards.py:507-524), `action_mirror` (rewards.py:281-302) and `action_sync`
    (rewards.py:305-337): no shipped robot cfg gives them a weight, so the
    per-robot fixtures never see them.  tests/golden/terms_extra.npz = the reference's functions on the recorded Go2 state
    (tools/gen_golden_extra_terms.py); here the same two kinds take the place of Go2's `feet_contact_without_cmd` /
    `feet_height_body` terms (same feet) in the descriptor and the oracle must reproduce them."""
    from robot_lab_amd.desc import REW

    x = np.load(os.path.join(GOLD, "terms_extra.npz"))
    g = np.load(os.path.join(GOLD, str(x["source"])))
    desc, extra = load_bundle(str(g["task"]))
    t, names = desc.task, list(desc.reward_names)
    swap = {"feet_contact": ("feet_contact_without_cmd", (float(x["expect_contact_num"]),)),
            "feet_height": ("feet_height_body", (float(x["target_height"]), float(x["tanh_mult"]))),
            # action_mirror (rewards.py:281-302) takes joint_mirror's place: Go2's cfg pairs the same joints (FR <-> RL, FL <-> RR), p0 = 1/2;
            # action_sync (rewards.py:305-337) joint_power's, with the index lists the model compiler writes for the reference's groups
            "action_mirror": ("joint_mirror", ()),
            "action_sync": ("joint_power", ())}
    for kind, (host, params) in swap.items():
        r = t.rewards[names.index(host)]
        r.kind = REW[kind]
        for i, p in enumerate(params):
            r.p[i] = p
    from robot_lab_amd.model.build import find_names

    assert [[str(n) for n in pair] for pair in x["mirror_joints"]] == [["FR.*", "RL.*"], ["FL.*", "RR.*"]]
    r, cols, grp = t.rewards[names.index("joint_power")], [], []
    for gi, group in enumerate(x["joint_groups"]):
        for name in group:
            (c,) = find_names(str(name), list(desc.joint_names))
            cols.append(c)
            grp.append(gi)
    for i, (c, gi) in enumerate(zip(cols, grp)):
        r.idx_a[i], r.idx_b[i] = c, gi
    r.n_idx, r.p[0] = len(cols), 1.0 / len(x["joint_groups"])
    N = int(g["N"])
    h, to, eo = build_world(desc, extra, N, 0)
    ora = OracleEnv(desc, h, to, N, int(g["seed"]), eo)
    for k in ("root_pos", "root_quat", "root_lin_vel", "root_ang_vel", "q", "qd", "base_com", "root_com"):
        ora.st[k] = g["st_" + k].copy()
    for k in ("applied_torque", "joint_acc", "force_hist", "contact_force", "timers", "action", "prev_action", "vel_command_b", "terminated"):
        setattr(ora, k, g[k].copy())
    ora.compute_rewards(ora.derived(), np.linalg.norm(ora.force_hist, axis=-1).max(axis=1))
    for j, kind in enumerate(str(n) for n in x["names"]):
        i = names.index(swap[kind][0])
        got = ora.reward_terms[i] / (float(t.rewards[i].weight) * ora.step_dt)
        np.testing.assert_allclose(got, x["values"][j], rtol=2e-6, atol=1e-7, err_msg=kind)
        assert np.abs(x["values"][j]).max() > 0
