"""CPU tier: the env-step lane program (the exact source hipcc compiles, run by the 4-thread lane
emulator) against the fp64 oracle.  This is the no-GPU stand-in for tests/test_gpu_parity.py."""
import numpy as np
import pytest

from helpers import A1_SYNC_GROUPS, assert_close, host_view, make_pair, oracle_root_state, set_action_sync

TASKS = [
    "RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0",
    "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0",
    "RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0",
    "RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0",
    "RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0",
    "RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0",
    # more quadrupeds of the reference, supported as data
    "RobotLab-Isaac-Velocity-Rough-Unitree-B2-v0",
    "RobotLab-Isaac-Velocity-Rough-Deeprobotics-Lite3-v0",
    "RobotLab-Isaac-Velocity-Rough-Deeprobotics-M20-v0",
    "RobotLab-Isaac-Velocity-Rough-Zsibot-ZSL1-v0",
    "RobotLab-Isaac-Velocity-Flat-Zsibot-ZSL1W-v0",
    # humanoids / bipeds on the G1 instance with padding (shorter trunk, fewer or shorter limbs)
    "RobotLab-Isaac-Velocity-Flat-RoboParty-ATOM01-v0",
    "RobotLab-Isaac-Velocity-Flat-RobotEra-Xbot-v0",
    "RobotLab-Isaac-Velocity-Flat-MagicLab-Bot-Gen1-v0",
    "RobotLab-Isaac-Velocity-Rough-Openloong-Loong-v0",
    # wheeled bipeds / out-of-tree-order joints / the hand-stand task (5 more reward kinds)
    "RobotLab-Isaac-Velocity-Rough-DDTRobot-Tita-v0",
    "RobotLab-Isaac-Velocity-Flat-MagicLab-Bot-Z1-v0",
    "RobotLab-Isaac-Velocity-Rough-HandStand-Unitree-A1-v0",
    "RobotLab-Isaac-Velocity-Flat-Unitree-B2W-v0",
    "RobotLab-Isaac-Velocity-Rough-MagicLab-Dog-W-v0",
    "RobotLab-Isaac-Velocity-Flat-MagicLab-Dog-v0",
    "RobotLab-Isaac-Velocity-Rough-Agibot-D1-v0",  # registers since the shims carry a `cusrl` stand-in (agibot_d1/agents/__init__.py:4)
    # a six-joint spine (waist, then head) with the arms leaving it at depth 3: the Topo<7,6,4,9> instance, 32 DoF
    "RobotLab-Isaac-Velocity-Flat-FFTAI-GR1T1-v0",
    "RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T2-v0",
    # a trunk of two PIECES on the base - the waist (carrying the legs) and a two-joint neck - with the arms on the base as well: Booster T1
    "RobotLab-Isaac-Velocity-Flat-Booster-T1-v0",
    "RobotLab-Isaac-Velocity-Rough-Booster-T1-v0",
]


@pytest.mark.parametrize("task", TASKS)
def test_lane_program_matches_oracle(task, emu_lib, monkeypatch):
    big = any(r in task for r in ("G1", "ATOM01", "Xbot", "Gen1", "Loong", "Tita", "Z1", "GR1", "T1"))
    N = 8 if big else 16  # big models: the fp64 oracle is the slow side
    if big:
        # the trunk + limbs instances add into env-shared words (ds_add_f32; lane THREADS here: the order of the additions is the scheduler's,
        # and over six free-running steps a last-bit difference can flip one contact - seen once in ~30 tier runs: GR1T2, one body's force
        # at step 1).  Lanes as fibers of one thread: one order.  The quadrupeds keep exercising the thread mode.
        monkeypatch.setenv("RL_EMU_FIBERS", "1")
    # GR1 (55 kg on two feet, drive stiffness up to 250 N m / rad): the fp32 program sits 3 - 5 x further from the fp64 oracle than on
    # the 35 kg G1 - root state to 3e-4, joint velocities to 1.4e-2, contact forces to 0.4 N over these six steps, not growing -
    # so its bands are 6 x the others'
    k = 6.0 if "GR1" in task else 1.0
    desc, ora, nat = make_pair(task, N, 42, emu_lib)
    o = ora.reset()
    nat.reset()
    assert_close("policy0", host_view(nat, "OBS_POLICY"), o[0], k * 1e-4, k * 1e-5)
    assert_close("critic0", host_view(nat, "OBS_CRITIC"), o[1], k * 1e-3, k * 1e-4)
    rng = np.random.default_rng(0)
    # hand-stand: joint_acc_l2 carries 10x the usual weight and the joint accelerations of an env whose foot is just
    # touching down differ by ~1 % between fp32 and fp64 (one env of 16 at steps 4 and 5)
    frac = 0.93 if "HandStand" in task else 1.0
    for s in range(6):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        o = ora.step(a)
        nat.step(a.ctypes.data)
        assert_close(f"reward[{s}]", host_view(nat, "REWARD"), ora.reward, k * 1e-3, k * 2e-5, frac)
        assert_close(f"terms[{s}]", host_view(nat, "REWARD_TERMS")[:, :N], ora.reward_terms, k * 2e-3, k * 2e-5, 0.99 if frac < 1 else 1.0)
        assert_close(f"cforce[{s}]", host_view(nat, "CONTACT_FORCE"), ora.contact_force, k * 5e-3, k * 5e-2)
    nat.export_state()
    assert_close("root", host_view(nat, "ROOT_STATE"), oracle_root_state(ora), k * 1e-3, k * 1e-4, 0.99 if frac < 1 else 1.0)
    assert_close("q", host_view(nat, "JOINT_POS"), ora.st["q"], k * 1e-3, k * 1e-4)
    assert_close("qd", host_view(nat, "JOINT_VEL"), ora.st["qd"], k * 2e-3, k * 2e-3, 0.99 if frac < 1 else 1.0)
    assert_close("timers", host_view(nat, "CONTACT_TIMERS"), ora.timers, k * 1e-5, k * 1e-6)
    assert_close("torque", host_view(nat, "JOINT_TORQUE"), ora.applied_torque, k * 2e-3, k * 2e-3)
    assert_close("cmd", host_view(nat, "COMMAND"), ora.vel_command_b, k * 1e-3, k * 1e-4)
    assert_close("policy", host_view(nat, "OBS_POLICY"), o[0], k * 2e-3, k * 2e-3)
    assert_close("critic", host_view(nat, "OBS_CRITIC"), o[1], k * 2e-3, k * 2e-3)
    nat.close()


def test_resets_and_logs(emu_lib):
    task, N = TASKS[1], 16
    desc, ora, nat = make_pair(task, N, 9, emu_lib)
    ora.reset()
    nat.reset()
    ep = np.zeros(N, dtype=np.int64)
    ep[::3] = ora.max_episode_length - 2
    ora.episode_length_buf[:] = ep
    host_view(nat, "EPISODE_LENGTH")[:] = ep
    rng = np.random.default_rng(1)
    seen = 0
    for s in range(3):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        o = ora.step(a)
        nat.step(a.ctypes.data)
        done = host_view(nat, "TERMINATED").astype(bool) | host_view(nat, "TIME_OUT").astype(bool)
        assert np.array_equal(done, ora.terminated | ora.time_outs)
        log = nat.read_log()
        if done.any():
            seen += int(done.sum())
            assert log[0] == done.sum() and log[1] == ora.time_outs_terms[0].sum()
            for i, name in enumerate(desc.reward_names):
                np.testing.assert_allclose(log[8 + i] / log[0] / ora.max_episode_length_s, ora.log["Episode_Reward/" + name], rtol=2e-3, atol=1e-7)
        assert np.array_equal(host_view(nat, "EPISODE_LENGTH"), ora.episode_length_buf)
    assert seen == 6
    nat.export_state()
    assert_close("root", host_view(nat, "ROOT_STATE"), oracle_root_state(ora), 1e-3, 1e-4)
    assert_close("critic", host_view(nat, "OBS_CRITIC"), o[1], 2e-3, 2e-3)
    assert np.array_equal(host_view(nat, "TERRAIN_LEVEL"), ora.terrain_levels)
    nat.close()


# A1, Go2W (4-joint chains: the merged instance - the trunk's spheres ride in free sphere slots of the limb groups), G1 (trunk +
# limbs); then Go2W on the unmerged 4-joint instance (RL_ENV_MERGE=0: what a robot whose trunk spheres do not fit runs on) and M20
# (merged, with the trunk's six spheres in the WHEEL groups: flagged slots on sub-lanes that do not own the trunk body's slot)
@pytest.mark.parametrize("task,N,steps,merge", [(TASKS[1], 16, 2, None), (TASKS[3], 8, 2, None), (TASKS[5], 4, 3, None),
                                                (TASKS[3], 8, 2, "0"), ("RobotLab-Isaac-Velocity-Rough-Deeprobotics-M20-v0", 8, 2, None),
                                                # Go2 (21 terms incl. the gait kind) and the HandStand task's own kinds
                                                (TASKS[2], 8, 2, None), ("RobotLab-Isaac-Velocity-Flat-HandStand-Unitree-A1-v0", 8, 2, None),
                                                # DDT Tita: the rot / pad quadruped instance (rotated joint frames, two empty limbs)
                                                ("RobotLab-Isaac-Velocity-Rough-DDTRobot-Tita-v0", 8, 2, None)])
@pytest.mark.parametrize("sub", ["4", "2"])
def test_sixteen_lane_mapping_matches_oracle(task, N, steps, merge, sub, emu_lib, monkeypatch):
    """The 16-lanes-per-env mapping (a DPP quad per limb; the default on the GPU: link groups dealt to the sub-lanes, contact
    stash, the trunk instance's limb-shared records) run by 16 host threads per env; and the 8-lane mapping (two sub-lanes per limb:
    what 5 - 11 k quadruped envs per GPU launch - each sub-lane takes two link groups)."""
    if sub == "2" and "G1" in task:
        pytest.skip("the trunk + limbs instance has the 16-lane mapping only")
    monkeypatch.setenv("RL_EMU_SUB", sub)
    if merge is not None:
        monkeypatch.setenv("RL_ENV_MERGE", merge)
    desc, ora, nat = make_pair(task, N, 21, emu_lib)
    o = ora.reset()
    nat.reset()
    rng = np.random.default_rng(5)
    for s in range(steps):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        o = ora.step(a)
        nat.step(a.ctypes.data)
    nat.export_state()
    assert_close("root", host_view(nat, "ROOT_STATE"), oracle_root_state(ora), 1e-3, 1e-4)
    assert_close("qd", host_view(nat, "JOINT_VEL"), ora.st["qd"], 2e-3, 2e-3)
    assert_close("terms", host_view(nat, "REWARD_TERMS")[:, :N], ora.reward_terms, 2e-3, 2e-5)
    assert_close("timers", host_view(nat, "CONTACT_TIMERS"), ora.timers, 1e-5, 1e-6)
    assert_close("critic", host_view(nat, "OBS_CRITIC"), o[1], 2e-3, 2e-3)
    nat.close()


# The 32-lanes-per-env mapping of the trunk + limbs instances (eight sub-lanes per limb - sub-lane s evaluates link group s -, two envs
# per wavefront: what a <= 2048-env G1 launch runs on the GPU since round 4): G1 Rough and Flat, a padded humanoid, a biped with two
# empty limbs, the six-joint-spine instance.  32 lanes per env as fibers of one host thread.
@pytest.mark.parametrize("task,k", [(TASKS[5], 1.0), (TASKS[4], 1.0), ("RobotLab-Isaac-Velocity-Flat-RobotEra-Xbot-v0", 1.0),
                                    ("RobotLab-Isaac-Velocity-Rough-Openloong-Loong-v0", 1.0), ("RobotLab-Isaac-Velocity-Flat-FFTAI-GR1T1-v0", 6.0),
                                    ("RobotLab-Isaac-Velocity-Rough-Booster-T1-v0", 1.0)])
def test_thirty_two_lane_mapping_matches_oracle(task, k, emu_lib, monkeypatch):
    monkeypatch.setenv("RL_EMU_SUB", "8")
    monkeypatch.setenv("RL_EMU_FIBERS", "1")
    N, steps = 4, 4
    desc, ora, nat = make_pair(task, N, 21, emu_lib)
    assert nat.envs_per_wavefront() == 2
    o = ora.reset()
    nat.reset()
    assert_close("critic0", host_view(nat, "OBS_CRITIC"), o[1], k * 1e-3, k * 1e-4)
    rng = np.random.default_rng(5)
    for s in range(steps):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        o = ora.step(a)
        nat.step(a.ctypes.data)
        assert_close(f"reward[{s}]", host_view(nat, "REWARD"), ora.reward, k * 1e-3, k * 2e-5)
        assert_close(f"cforce[{s}]", host_view(nat, "CONTACT_FORCE"), ora.contact_force, k * 5e-3, k * 5e-2)
    nat.export_state()
    assert_close("root", host_view(nat, "ROOT_STATE"), oracle_root_state(ora), k * 1e-3, k * 1e-4)
    assert_close("q", host_view(nat, "JOINT_POS"), ora.st["q"], k * 1e-3, k * 1e-4)
    assert_close("qd", host_view(nat, "JOINT_VEL"), ora.st["qd"], k * 2e-3, k * 2e-3)
    assert_close("terms", host_view(nat, "REWARD_TERMS")[:, :N], ora.reward_terms, k * 2e-3, k * 2e-5)
    assert_close("timers", host_view(nat, "CONTACT_TIMERS"), ora.timers, k * 1e-5, k * 1e-6)
    assert_close("torque", host_view(nat, "JOINT_TORQUE"), ora.applied_torque, k * 2e-3, k * 2e-3)
    assert_close("policy", host_view(nat, "OBS_POLICY"), o[0], k * 2e-3, k * 2e-3)
    assert_close("critic", host_view(nat, "OBS_CRITIC"), o[1], k * 2e-3, k * 2e-3)
    nat.close()


def _switch_kinds(desc):
    """The reward kinds no shipped cfg gives a weight: `feet_height` (world frame, rewards.py:507-524) in place of A1's
    `feet_height_body`, `feet_contact` (rewards.py:399-413, expects 2 feet down) in place of `feet_contact_without_cmd`,
    `joint_vel_l2` in place of `joint_acc_l2` - same body / joint masks, the kind and its parameters swapped -, `action_mirror` and
    `action_sync` (rewards.py:281-337) with the parameters velocity_env_cfg.py:478-498 declares."""
    from robot_lab_amd.desc import REW as REW_KINDS

    t, names = desc.task, list(desc.reward_names)
    r = t.rewards[names.index("feet_height_body")]
    r.kind, r.p[0], r.p[1] = REW_KINDS["feet_height"], 0.05, 2.0
    r = t.rewards[names.index("feet_contact_without_cmd")]
    r.kind, r.p[0], r.weight = REW_KINDS["feet_contact"], 2.0, -0.1
    r = t.rewards[names.index("joint_acc_l2")]
    r.kind, r.weight = REW_KINDS["joint_vel_l2"], -1e-3
    # action_mirror (rewards.py:281-302) in place of joint_mirror - A1's cfg pairs the same joints -, action_sync (rewards.py:305-337)
    # over the reference's three joint groups in place of joint_power
    r = t.rewards[names.index("joint_mirror")]
    r.kind, r.weight = REW_KINDS["action_mirror"], -0.05
    set_action_sync(desc, "joint_power", A1_SYNC_GROUPS).weight = -0.1
    return desc


def test_reward_kinds_without_a_cfg(emu_lib):
    desc, ora, nat = make_pair(TASKS[1], 16, 8, emu_lib, mutate=_switch_kinds)
    ora.reset()
    nat.reset()
    rng = np.random.default_rng(2)
    names = list(desc.reward_names)
    seen = np.zeros(5)
    for s in range(6):
        a = rng.uniform(-1, 1, (16, desc.model.num_dof)).astype(np.float32)
        ora.step(a)
        nat.step(a.ctypes.data)
        got, want = host_view(nat, "REWARD_TERMS")[:, :16], ora.reward_terms
        assert_close(f"terms[{s}]", got, want, 2e-3, 2e-5)
        seen += [np.abs(want[names.index(n)]).max() for n in ("feet_height_body", "feet_contact_without_cmd", "joint_acc_l2", "joint_mirror", "joint_power")]
    assert (seen > 0).all(), seen  # the five swapped terms really produced something
    nat.close()


@pytest.mark.parametrize("task", [TASKS[1], TASKS[3]])
def test_partial_reset_by_env_ids(task, emu_lib):
    """rl_env_reset(env, ids, n): only the named envs are reset (all their draws as in the oracle), the others keep their state bit
    for bit; an id out of range is an error.  (The GPU twin: tests/test_gpu_edge_cases.py.)"""
    N = 16
    desc, ora, nat = make_pair(task, N, 9, emu_lib)
    ora.reset()
    nat.reset()
    rng = np.random.default_rng(0)
    for _ in range(2):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        ora.step(a)
        nat.step(a.ctypes.data)
    nat.export_state()
    before = {k: host_view(nat, k).copy() for k in ("ROOT_STATE", "JOINT_POS", "JOINT_VEL", "OBS_CRITIC", "EPISODE_LENGTH")}
    ids = [3, 4, 12, 15]
    keep = np.setdiff1d(np.arange(N), ids)
    nat.reset(ids)
    o = ora.reset(env_ids=ids)
    nat.export_state()
    for k, v in before.items():
        if k != "OBS_CRITIC":  # (observations are recomputed for everybody: fresh noise on the policy row, same critic row)
            assert np.array_equal(host_view(nat, k)[keep], v[keep]), k
    assert (host_view(nat, "EPISODE_LENGTH")[ids] == 0).all()
    assert_close("root", host_view(nat, "ROOT_STATE")[ids], oracle_root_state(ora)[ids], 1e-4, 1e-5)
    assert_close("q", host_view(nat, "JOINT_POS")[ids], ora.st["q"][ids], 1e-5, 1e-6)
    assert_close("critic", host_view(nat, "OBS_CRITIC"), o[1], 2e-3, 2e-3)
    with pytest.raises(Exception):
        nat.reset([N])
    nat.close()


@pytest.mark.parametrize("sub", ["4", "8"])
def test_a_hosted_spine_link_carries_the_robot(sub, emu_lib, monkeypatch):
    """`rl_model_desc.chain_grp0`: a spine link nothing hangs off - GR1's head - rides in the group 0 of a spared lane.  The
    shipped GR1 URDF has no head geometry, so the descriptor is edited the way model/build.py would compile one that has: a sphere on
    the head body, the second arm lane's group 0 moved from the torso (depth 3) to the head link (depth 6).  Robots dropped on their
    heads: the head sphere carries load, and the lane program (frame, twist and trunk accumulator of a group 0 that is NOT the limb's
    attachment link) agrees with the oracle, which knows nothing of lanes."""
    from helpers import emu_load_state, emu_read_state

    monkeypatch.setenv("RL_EMU_SUB", sub)
    monkeypatch.setenv("RL_EMU_FIBERS", "1")
    task, N = "RobotLab-Isaac-Velocity-Flat-FFTAI-GR1T1-v0", 4

    def add_head_sphere(desc):
        m = desc.model
        head = desc.body_names.index("head_pitch")
        g = m.num_spheres
        m.sphere_body[g], m.sphere_radius[g] = head, 0.10
        m.sphere_center[g][0] = m.sphere_center[g][1] = m.sphere_center[g][2] = 0.0
        m.num_spheres = g + 1
        depth = [int(m.trunk_link[i]) for i in range(m.num_trunk)].index(int(m.body_link[head])) + 1
        assert list(m.chain_attach) == [0, 0, 3, 3] and depth > 3
        m.chain_grp0[3] = depth + 1
        desc.task.term_illegal_contact = 0  # (the cfg terminates on head contact - which is the point of hosting it; here the contact itself is watched)

    desc, ora, nat = make_pair(task, N, 5, emu_lib, mutate=add_head_sphere)
    head = desc.body_names.index("head_pitch")
    ora.reset()
    nat.reset()
    state = emu_read_state(nat)
    rs = state["root_state"].copy()
    rs[:, 2] = 5.0
    rs[:, 3:7] = [0.0, 1.0, 0.0, 0.0]         # upside down (a half turn about x)
    rs[:, 7:13] = 0.0
    state["root_state"] = rs
    ora.load_state(state)
    head_z = ora.phys.body_kinematics(ora.st)[0][:, head, 2]  # where that puts the head: lower the robots until its sphere is 1 - 4 cm in the plane
    rs[:, 2] = 5.0 - (head_z - 0.10) - np.linspace(0.01, 0.04, N)
    state["root_state"] = rs
    emu_load_state(nat, state)
    state = emu_read_state(nat)
    ora.load_state(state)
    a = np.zeros((N, desc.model.num_dof), dtype=np.float32)
    k = 6.0  # (GR1's bands: test_lane_program_matches_oracle)
    for s in range(2):  # (two steps = eight substeps of a 55 kg robot landing on its head: beyond that the free runs drift apart like any stiff contact)
        ora.step(a)
        nat.step(a.ctypes.data)
        assert_close(f"cforce[{s}]", host_view(nat, "CONTACT_FORCE"), ora.contact_force, k * 5e-3, k * 5e-2)
    assert (np.linalg.norm(ora.contact_force[:, head], axis=1) > 20.0).sum() >= 2, ora.contact_force[:, head]  # the head carries load
    nat.export_state()
    assert_close("root", host_view(nat, "ROOT_STATE"), oracle_root_state(ora), k * 1e-3, k * 1e-4)
    assert_close("qd", host_view(nat, "JOINT_VEL"), ora.st["qd"], k * 2e-3, k * 2e-3)
    assert_close("timers", host_view(nat, "CONTACT_TIMERS"), ora.timers, k * 1e-5, k * 1e-6)
    nat.close()


@pytest.mark.parametrize("sub", ["1", "4", "8"])
def test_six_joint_spine_without_contacts_meets_the_standard_bands(sub, emu_lib, monkeypatch):
    """GR1's wider bands (k = 6 above) are calibrated to what its stiff two-foot contacts do to an fp32 step, not to an
    independent estimate - a defect of the NW = 6 code paths (trunk links k and k + 4 sharing an owner, the dealt kinematics with 13
    chain slots, the packed tables) could hide inside them.  So the same robot, same actions, in FREE FALL: kinematics, velocities,
    elimination, trunk pieces and the outward pass are all exercised, the contact conditioning is not - held to the bands every other
    robot gets (k = 1), in every lane mapping."""
    from helpers import emu_load_state, emu_read_state

    monkeypatch.setenv("RL_EMU_SUB", sub)
    monkeypatch.setenv("RL_EMU_FIBERS", "1")
    task, N = "RobotLab-Isaac-Velocity-Flat-FFTAI-GR1T1-v0", 4
    desc, ora, nat = make_pair(task, N, 3, emu_lib)
    ora.reset()
    nat.reset()
    state = emu_read_state(nat)
    rs = state["root_state"].copy()
    rs[:, 2] += 6.0  # six steps of 20 ms: it falls 7 cm
    state["root_state"] = rs
    emu_load_state(nat, state)
    state = emu_read_state(nat)
    ora.load_state(state)
    rng = np.random.default_rng(9)
    for s in range(6):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        ora.step(a)
        nat.step(a.ctypes.data)
        assert np.abs(ora.contact_force).max() == 0.0
        assert_close(f"torque[{s}]", host_view(nat, "JOINT_TORQUE"), ora.applied_torque, 2e-3, 2e-3)
    nat.export_state()
    assert_close("root", host_view(nat, "ROOT_STATE"), oracle_root_state(ora), 1e-3, 1e-4)
    assert_close("q", host_view(nat, "JOINT_POS"), ora.st["q"], 1e-3, 1e-4)
    assert_close("qd", host_view(nat, "JOINT_VEL"), ora.st["qd"], 2e-3, 2e-3)
    nat.close()


def test_tita_on_the_trunk_and_limbs_instance(emu_lib, monkeypatch):
    """RL_ENV_ROTPAD=0: DDT Tita back on the trunk + limbs instance it ran on until round 4 (what a quadruped-shaped robot with self-collision
    pairs or more than three spheres on a link would still need); the default - the rot / pad quadruped instance - is in the lists above."""
    monkeypatch.setenv("RL_ENV_ROTPAD", "0")
    monkeypatch.setenv("RL_EMU_SUB", "8")
    monkeypatch.setenv("RL_EMU_FIBERS", "1")
    N = 4
    desc, ora, nat = make_pair("RobotLab-Isaac-Velocity-Flat-DDTRobot-Tita-v0", N, 21, emu_lib)
    assert nat.envs_per_wavefront() == 2
    o = ora.reset()
    nat.reset()
    rng = np.random.default_rng(5)
    for s in range(3):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        o = ora.step(a)
        nat.step(a.ctypes.data)
        assert_close(f"reward[{s}]", host_view(nat, "REWARD"), ora.reward, 1e-3, 2e-5)
    nat.export_state()
    assert_close("root", host_view(nat, "ROOT_STATE"), oracle_root_state(ora), 1e-3, 1e-4)
    assert_close("qd", host_view(nat, "JOINT_VEL"), ora.st["qd"], 2e-3, 2e-3)
    assert_close("critic", host_view(nat, "OBS_CRITIC"), o[1], 2e-3, 2e-3)
    nat.close()


def test_log_partial_rows_and_inheritance(emu_lib, monkeypatch):
    """The episode log over SEVERAL wavefronts: a ring slot is RL_LOG_PARTS partial rows (wavefront w adds into row w % RL_LOG_PARTS, readers
    sum - include/rl_env.h RL_BUF_LOG), kept by the first RL_LOG_PARTS wavefronts of the next launch: a step that reset nobody inherits its
    predecessor row by row, a step that did starts from zeros.  16 envs per wavefront x 5 wavefronts, a third of the envs timing out in steps
    1 and 3, nobody in steps 2, 4, 5."""
    from robot_lab_amd.desc import RL_LOG_PARTS, RL_LOG_SIZE

    task, N = TASKS[1], 80
    desc, ora, nat = make_pair(task, N, 9, emu_lib)
    ora.reset()
    nat.reset()
    ep = np.zeros(N, dtype=np.int64)
    ep[::3] = ora.max_episode_length - 1
    ep[1::3] = ora.max_episode_length - 3
    ora.episode_length_buf[:] = ep
    host_view(nat, "EPISODE_LENGTH")[:] = ep
    rng = np.random.default_rng(1)
    ring = host_view(nat, "LOG")
    assert ring.shape[1:] == (RL_LOG_PARTS, RL_LOG_SIZE)
    last = None
    for s in range(5):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        ora.step(a)
        nat.step(a.ctypes.data)
        done = ora.terminated | ora.time_outs
        slot = ring[nat.log_slot()]
        assert slot[:, RL_LOG_SIZE - 1].sum() == done.sum()  # the slot's own resets, spread over the rows of the wavefronts that had one
        log = nat.read_log()
        if done.any():
            assert s in (0, 2) and (slot[:5, 0] > 0).sum() >= 4 and not slot[5:].any()  # five wavefronts, five rows
            assert log[0] == done.sum() and log[1] == ora.time_outs_terms[0].sum()
            for i, name in enumerate(desc.reward_names):
                np.testing.assert_allclose(log[8 + i] / log[0] / ora.max_episode_length_s, ora.log["Episode_Reward/" + name], rtol=2e-3, atol=1e-7)
            last = log.copy()
        else:
            np.testing.assert_array_equal(log[:RL_LOG_SIZE - 1], last[:RL_LOG_SIZE - 1])  # the most recent step that reset an env
        if s >= 1:  # the slot of the step before is final: it holds its own log, or its predecessor's - never zeros after the first reset
            prev = ring[(nat.log_slot() - 1) % ring.shape[0]].sum(axis=0)
            assert prev[0] > 0
    nat.close()
