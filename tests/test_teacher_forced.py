"""CPU tier of the teacher-forced parity check (SURVEY.md 8(c) last row): the lane program (CPU lane emulator, the source
hipcc compiles) and the fp64 oracle take ONE step from a SHARED state that the lane program itself reached after a few
random-action steps (contacts, resets, commands in every phase).  Also covers the C-ABI state round trip
(rl_env_export_state / rl_env_commit_state / rl_env_import_state, rl_env_step_count) that the GPU tier uses at the
BASELINE sizes (tests/test_gpu_teacher_forced.py)."""
import numpy as np
import pytest

from helpers import emu_load_state, emu_read_state, host_view, make_pair, teacher_forced_check

RL_TS_CMD_TIME_LEFT, RL_TS_PUSH_TIME_LEFT = 4, 7  # include/rl_env.h rl_task_state_field

CASES = [
    ("RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0", 48, 8),
    ("RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0", 32, 6),
    ("RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0", 32, 6),
    ("RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0", 8, 6),
    ("RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0", 8, 6),  # the six-joint-spine instance, judged by the oracle's own fp32 sensitivity
    ("RobotLab-Isaac-Velocity-Rough-Booster-T1-v0", 8, 6),   # a trunk of two pieces, passive neck joints (no saturation switch: the mask must not flag them)
]


def _outputs(nat, N):
    got = emu_read_state(nat)
    got.update(reward=host_view(nat, "REWARD").copy(), reward_terms=host_view(nat, "REWARD_TERMS")[:, :N].copy(),
               done=host_view(nat, "TERMINATED").astype(bool) | host_view(nat, "TIME_OUT").astype(bool),
               obs_policy=host_view(nat, "OBS_POLICY").copy(), obs_critic=host_view(nat, "OBS_CRITIC").copy())
    return got


@pytest.mark.parametrize("task,N,K", CASES)
def test_one_step_from_shared_state(task, N, K, emu_lib):
    desc, ora, nat = make_pair(task, N, 42, emu_lib)
    nat.reset()
    rng = np.random.default_rng(1)
    ep = rng.integers(0, nat.max_episode_length, N)
    ep[::5] = nat.max_episode_length - 1 - (np.arange(len(ep[::5])) % (K + 2))  # time-outs during the warm-up ...
    ep[1:4] = nat.max_episode_length - 1 - K                                     # ... and on the compared step
    host_view(nat, "EPISODE_LENGTH")[:] = ep
    for _ in range(K):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        nat.step(a.ctypes.data)
    state = emu_read_state(nat)
    assert state["step_count"] == K
    # the interval events of the compared step: the push (every 10 - 15 s, velocity_env_cfg.py:366-371) and the command resampling
    # (every 10 s, :106-117) are due in a few envs - their timers are part of the exchanged state
    ts = state["task_state"].copy()
    ts[2::7, RL_TS_PUSH_TIME_LEFT] = 0.015
    ts[4::9, RL_TS_CMD_TIME_LEFT] = 0.015
    state["task_state"] = ts
    # ... and two envs are moved: one 4.5 m away from its origin on the step its episode times out (terrain_levels_vel: one level up,
    # velocity_env_cfg.py:671), one across the border of the terrain (terrain_out_of_bounds, :656-664: a time-out-type termination)
    td, tk = desc.terrain, desc.task
    rs = state["root_state"].copy()
    up, oob = 1, 6
    rs[up, 0] += 4.5
    rs[oob, 0] = 0.5 * (td.num_rows * td.tile_size + 2 * td.border) - tk.oob_buffer + 0.5
    for i in (up, oob):
        rs[i, 2] = ora.phys.terrain.sample(rs[i, 0:1].astype(np.float64), rs[i, 1:2].astype(np.float64))[0][0] + 0.45
        rs[i, 3:7] = [1.0, 0.0, 0.0, 0.0]
    state["root_state"] = rs
    level_before = state["terrain_level"].copy()
    emu_load_state(nat, state)
    state = emu_read_state(nat)
    a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
    nat.step(a.ctypes.data)
    got = _outputs(nat, N)
    assert (got["task_state"][2::7, RL_TS_PUSH_TIME_LEFT] > 5.0).all()  # fired: redrawn U(10, 15) (the velocity kick is in the comparison below)
    assert (got["task_state"][4::9, RL_TS_CMD_TIME_LEFT] > 5.0).all()   # resampled: 10 s
    if not td.is_plane:
        assert got["done"][oob] and got["done"][up]
        if tk.term_out_of_bounds:
            # (GR1 under random actions has its torso on the ground in that env at that step as well: a termination besides the time-out)
            assert host_view(nat, "TIME_OUT")[oob] and ("GR1" in task or not host_view(nat, "TERMINATED")[oob])
        if td.curriculum and level_before[up] < td.num_rows - 1:
            assert got["terrain_level"][up] == level_before[up] + 1
    # small batches: one env on a switch is already 2 - 12 % of the batch, so the mask-size bound is checked at full size only
    rep = teacher_forced_check(ora, state, a, got, max_mask=0.25)
    assert rep["done_count"] > 0  # the compared step itself resets somebody
    nat.close()


@pytest.mark.parametrize("task,merge", [("RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0", None), ("RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0", "0"),
                                        ("RobotLab-Isaac-Velocity-Rough-Deeprobotics-M20-v0", None), ("RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0", None),
                                        ("RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0", None),  # G1: torso on the ground = illegal_contact -> terminated, is_terminated reward
                                        ("RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0", "sub8"),  # ... and in the 32-lane mapping (group 0 = the trunk share, sub-lane 0)
                                        ("RobotLab-Isaac-Velocity-Rough-Booster-T1-v0", "sub8"),     # the base body itself + the head on the second trunk piece
                                        ("RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0", "sub8")])
def test_one_step_with_the_trunk_on_the_ground(task, merge, emu_lib, monkeypatch):
    """Robots lying on their backs in the 16-lanes-per-env mapping: the TRUNK's collision spheres carry the robot - the contacts
    random-action warm-ups rarely reach.  On the merged 4-joint instance those spheres sit in flagged slots
    of limb link groups (Go2W: hip groups, with the sub-lane that owns the trunk body's slot; M20: wheel groups, with another
    sub-lane - separate base record, twist, friction row and force sum); RL_ENV_MERGE=0 and A1 take the group-0 path."""
    monkeypatch.setenv("RL_EMU_SUB", "4")
    if merge == "sub8":
        monkeypatch.setenv("RL_EMU_SUB", "8")
        monkeypatch.setenv("RL_EMU_FIBERS", "1")
    elif merge is not None:
        monkeypatch.setenv("RL_ENV_MERGE", merge)
    N = 8
    desc, ora, nat = make_pair(task, N, 5, emu_lib)
    nat.reset()
    rng = np.random.default_rng(3)
    for _ in range(2):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        nat.step(a.ctypes.data)
    state = emu_read_state(nat)
    rs = state["root_state"].copy()
    ground = ora.phys.terrain.sample(rs[:, 0].astype(np.float64), rs[:, 1].astype(np.float64))[0]
    rs[:6, 2] = (ground + np.linspace(0.015, 0.04, N))[:6]  # ON ITS BACK (legs in the air, or they would carry it), the trunk's spheres
    rs[:6, 3:7] = [0.0, 1.0, 0.0, 0.0]                       # (r = 0.047 m on Go2W) pressed 1 - 3 cm into the ground
    rs[:6, 7:13] *= 0.1
    state["root_state"] = rs
    emu_load_state(nat, state)
    state = emu_read_state(nat)
    a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
    nat.step(a.ctypes.data)
    # the trunk body (body 0) carries load in most of the lowered envs - or the env was terminated for it (illegal_contact on the
    # base, where the cfg has that term) and reset, which clears the sensor
    trunk = np.linalg.norm(host_view(nat, "CONTACT_FORCE")[:, 0], axis=1) > 1.0
    done = host_view(nat, "TERMINATED").astype(bool)
    assert (trunk | done)[:6].sum() >= 4 and not (trunk | done)[6:].any(), (trunk, done)
    teacher_forced_check(ora, state, a, _outputs(nat, N), max_mask=0.5)
    nat.close()


def test_state_round_trip_is_exact(emu_lib):
    """export -> commit into a second env -> both step bit-identically (every carried field is in the exchange)."""
    task, N = CASES[0][0], 16
    desc, _, nat = make_pair(task, N, 7, emu_lib)
    _, _, nat2 = make_pair(task, N, 7, emu_lib)
    nat.reset()
    nat2.reset()
    rng = np.random.default_rng(2)
    host_view(nat, "EPISODE_LENGTH")[:] = rng.integers(0, nat.max_episode_length, N)
    for _ in range(5):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        nat.step(a.ctypes.data)
    emu_load_state(nat2, emu_read_state(nat))
    assert nat2.step_count == nat.step_count == 5
    for _ in range(2):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        nat.step(a.ctypes.data)
        nat2.step(a.ctypes.data)
        for name in ("OBS_POLICY", "OBS_CRITIC", "REWARD", "TERMINATED", "TIME_OUT", "EPISODE_LENGTH", "EPISODE_SUMS", "COMMAND"):
            assert np.array_equal(host_view(nat, name), host_view(nat2, name)), name
    s1, s2 = emu_read_state(nat), emu_read_state(nat2)
    for k in s1:
        assert np.array_equal(s1[k], s2[k]), k
    nat.close()
    nat2.close()


def test_import_state_host_arrays(emu_lib):
    """rl_env_import_state (host arrays; any part may be NULL) overwrites exactly the given parts."""
    task, N = CASES[0][0], 16
    desc, _, nat = make_pair(task, N, 3, emu_lib)
    nat.reset()
    before = emu_read_state(nat)
    q = (before["joint_pos"] + 0.01).astype(np.float32)
    nat.import_state(0, q.ctypes.data, 0)
    after = emu_read_state(nat)
    assert np.array_equal(after["joint_pos"], q)
    for k in before:
        if k != "joint_pos":
            assert np.array_equal(before[k], after[k]), k
    nat.close()


def test_observation_buffers_alternate(emu_lib):
    """The observations returned by step t are not touched by step t + 1 (include/rl_env.h "Ownership": rsl_rl's PPO.act keeps
    a reference to them across env.step)."""
    task, N = CASES[0][0], 16
    desc, _, nat = make_pair(task, N, 3, emu_lib)
    nat.reset()
    rng = np.random.default_rng(0)
    views, snaps = [], []
    for t in range(3):
        a = rng.uniform(-1, 1, (N, desc.model.num_dof)).astype(np.float32)
        nat.step(a.ctypes.data)
        assert nat.obs_slot() == (t + 1 + 1) % 2  # reset() wrote slot 1, the steps alternate from there
        views.append(host_view(nat, "OBS_CRITIC"))  # a VIEW of the buffer step t wrote
        snaps.append(views[-1].copy())
        if t >= 1:
            assert np.array_equal(views[t - 1], snaps[t - 1])  # step t did not overwrite what step t - 1 returned
            assert not np.array_equal(views[t], views[t - 1])
    ring = host_view(nat, "OBS_CRITIC_RING")
    assert ring.shape[0] == 2 and np.array_equal(ring[nat.obs_slot(), :N], views[-1])
    nat.close()
