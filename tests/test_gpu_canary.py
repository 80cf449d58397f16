"""`-m gpu`: canary for the `__launch_bounds__(64)` miscompile (profiles/r02_launch_bounds64_miscompile.txt, csrc/rl_env.hip).

The symptom of that defect: hipcc parked the env's push timer in an AGPR, lent its VGPR to a block that ran under a narrowed EXEC
mask and reloaded it under the narrower mask, so `push_left < 1e-6` was evaluated on a stale register and the interval push
(velocity_env_cfg.py:366-371) fired in EVERY env on EVERY step.  The full parity tier catches that, but only as "velocities
differ"; this test names the symptom and is cheap enough to run for every lane-program instance x workgroup shape x lane mapping
that the product library ships:

    after one step from reset, the push timer and the command timer of every env that was not reset by that step have
    decreased by exactly step_dt (one fp32 subtraction - bit exact), and nobody's timer was redrawn.

`__graft_entry__.smoke()` runs the same assertion on A1 (both workgroup shapes) and G1."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RL_TS_CMD_TIME_LEFT, RL_TS_PUSH_TIME_LEFT = 4, 7  # include/rl_env.h rl_task_state_field

# (task, RL_ENV_MERGE): one robot per lane-program instance
INSTANCES = [
    ("RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0", None),        # Topo<3,0,3,6>
    ("RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0", None),       # Topo<3,0,3,6>, 21 reward terms (two trips of the term loop)
    ("RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0", None),      # Topo<4,0,3,6,1> merged
    ("RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0", "0"),       # Topo<4,0,3,6>
    ("RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0", None),        # Topo<7,3,4,9>
    ("RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0", None),       # Topo<7,6,4,9>: a six-joint spine
]
# RL_ENV_WG: "" = the shape the launch size selects (single-wavefront workgroups at this size), "-4" = four wavefronts per workgroup
# (what >= 4096 quadruped envs launch), RL_ENV_SUB=1 = the one-lane-per-limb mapping
SHAPES = [("", "4"), ("-4", "4"), ("", "1"), ("-4", "1"), ("", "2"), ("-4", "2"), ("", "8"), ("-4", "8")]  # 8: the 32-lane mapping of the trunk + limbs instances


def timers_tick_exactly(env, torch, steps=3):
    """The assertion itself (shared with __graft_entry__.smoke): returns the number of (env, step) pairs checked."""
    N = env.num_envs
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    dt = np.float32(env.step_dt)
    checked = 0
    for _ in range(steps):
        before = env.read_state()["task_state"]
        assert (before[:, RL_TS_PUSH_TIME_LEFT] > 2 * dt).all() and (before[:, RL_TS_CMD_TIME_LEFT] > 2 * dt).all()  # nothing is due
        _, _, term, tout, _ = env.step(torch.rand(N, env.num_actions, device="cuda", generator=g) * 2 - 1)
        alive = ~(term | tout).cpu().numpy()
        after = env.read_state()["task_state"]
        for f, name in ((RL_TS_PUSH_TIME_LEFT, "push"), (RL_TS_CMD_TIME_LEFT, "command")):
            want = (before[:, f].astype(np.float32) - dt).astype(np.float32)
            wrong = np.nonzero((after[:, f] != want) & alive)[0]
            assert len(wrong) == 0, (f"{name} timer of {len(wrong)} / {int(alive.sum())} live envs did not tick by exactly step_dt "
                                     f"(first: env {wrong[0]}: {before[wrong[0], f]} -> {after[wrong[0], f]}): the interval event fired or "
                                     f"the timer register was clobbered - the launch-bounds miscompile symptom")
        checked += int(alive.sum())
    return checked


@pytest.mark.parametrize("wg,sub", SHAPES)
@pytest.mark.parametrize("task,merge", INSTANCES)
def test_interval_timers_tick_by_exactly_step_dt(task, merge, wg, sub, monkeypatch):
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    if wg:
        monkeypatch.setenv("RL_ENV_WG", wg)
    trunk = "G1" in task or "GR1" in task
    monkeypatch.setenv("RL_ENV_SUB", sub)
    if merge is not None:
        monkeypatch.setenv("RL_ENV_MERGE", merge)
    if sub == "8" and not trunk:
        pytest.skip("eight sub-lanes per limb: trunk + limbs instances only")
    if sub == "4" and wg == "-4" and trunk:
        pytest.skip("the 16-lane mapping of the trunk + limbs instance ships single-wavefront workgroups only")
    if sub == "2" and trunk:
        pytest.skip("the trunk + limbs instance has the 16- and 32-lane mappings only")
    if sub == "1" and ("G1" in task or "GR1" in task):
        # the trunk + limbs instance keeps its kinematics / link records in limb-shared LDS words: with one lane per limb (64 limbs per
        # wavefront) that is 115 KB + 30 KB of sensor rows - it exists on the CPU lane emulator only, rl_env_create refuses it on the GPU
        pytest.skip("the one-lane-per-limb mapping of the trunk + limbs instance does not fit the LDS of a CU (CPU emulator only)")
    env = ManagerBasedRLEnv(task, num_envs=256, seed=3, device="cuda:0")
    assert timers_tick_exactly(env, torch) > 256
    env.close()


def _eventful_state(env, seed):
    """Edit the carried state so that the next steps contain every rare branch: a seventh of the episodes at their last step (time-out
    resets with all their draws), push / command timers about to expire in a ninth / an eleventh of the envs."""
    st = env.read_state()
    N, L = env.num_envs, env.max_episode_length
    rng = np.random.default_rng(seed)
    ep = rng.integers(0, L - 40, N)
    ep[::7] = L - 1 - (np.arange(len(ep[::7])) % 5)
    ts = st["task_state"].copy()
    dt = np.float32(env.step_dt)
    ts[::9, RL_TS_PUSH_TIME_LEFT] = dt * (1 + np.arange(len(ts[::9])) % 4).astype(np.float32)
    ts[::11, RL_TS_CMD_TIME_LEFT] = dt * (1 + np.arange(len(ts[::11])) % 4).astype(np.float32)
    env.load_state({"task_state": ts, "episode_length": ep})


# float fields of a step's trace whose bits may differ by contraction-level round-off between two kernels of one source (see the test);
# everything else - flags, counters, timers' ticks - is compared exactly
ROUNDOFF_REL = 2e-5


def _compare_step(s, a, b):
    """One step of two kernels from the SAME state: exact, or - for float fields - within ROUNDOFF_REL of the field's scale.  Returns the
    float fields that were not bit-equal (and by how much)."""
    soft = {}
    for k in a:
        x, y = np.asarray(a[k]), np.asarray(b[k])
        if x.dtype.kind != "f":
            assert np.array_equal(x, y), f"step {s}: '{k}' (exact field) differs between one- and four-wavefront workgroups"
            continue
        if np.array_equal(x, y, equal_nan=True):
            continue
        bad = np.argwhere(x != y)
        err = np.abs(x.astype(np.float64) - y.astype(np.float64))
        scale = np.maximum(np.abs(y.astype(np.float64)), 1.0)
        worst = float((err / scale).max())
        if k == "contact_timers" or worst > ROUNDOFF_REL:
            raise AssertionError(f"step {s}: '{k}' differs between one- and four-wavefront workgroups in {len(bad)} entries, first at {bad[0].tolist()}: "
                                 f"{x[tuple(bad[0])]} vs {y[tuple(bad[0])]} (worst relative {worst:.2e}) - same arithmetic, different register "
                                 f"allocation: a miscompile")
        soft[k] = (len(bad), worst)
    return soft


@pytest.mark.parametrize("sub", ["4", "1", "2", "8"])
@pytest.mark.parametrize("task,merge", INSTANCES)
def test_kernel_shapes_agree_bit_for_bit(task, merge, sub, monkeypatch):
    """The SAME lane program is compiled into several kernels (workgroup of one / of four wavefronts; step / reset entry): different
    register allocations of identical arithmetic.  Run from the same state with the same actions they must produce the same results - a
    value clobbered by a live-range split under a narrowed EXEC mask (the defect class of profiles/r02_launch_bounds64_miscompile.txt,
    seen again as corrupted commands in profiles/r03d_pin_desc_miscompile.txt) shows up as a difference in whichever state word,
    observation, reward term or flag it touches, without an oracle and at a size that takes seconds.  12 eventful steps: time-out
    resets, interval pushes and command resampling are due in some envs on every one of them.

    "The same results" = the same BITS for every flag, counter and timer, and for every float either the same bits (the usual case:
    then the two runs free-run side by side) or - round 5 - agreement to 2e-5 of the field's scale, step by step from a shared state:
    under -ffp-contract=fast the back end fuses a multiply-add or not depending on what else uses the product, and its unrolling / CSE
    decisions differ between the two kernels (profiles/r05b_canary_contraction.txt: one ulp in a reset env's heading target).  A
    clobbered register is a wrong VALUE (another variable's), orders of magnitude beyond that; the second env adopts the first one's state
    after a step that was not bit-equal, so round-off cannot grow into a false alarm."""
    import torch

    from robot_lab_amd.env import ManagerBasedRLEnv

    if merge is not None:
        monkeypatch.setenv("RL_ENV_MERGE", merge)
    trunk = "G1" in task or "GR1" in task
    if (sub in ("1", "2") and trunk) or (sub == "8" and not trunk):
        pytest.skip("the trunk + limbs instance has the 16- and 32-lane mappings, the quadrupeds the 16-, 8- and 4-lane ones")
    monkeypatch.setenv("RL_ENV_SUB", sub)
    N = 512
    envs = []
    for wg in ("1", "-4"):
        monkeypatch.setenv("RL_ENV_WG", wg)
        env = ManagerBasedRLEnv(task, num_envs=N, seed=5, device="cuda:0")
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        for _ in range(3):
            env.step(torch.rand(N, env.num_actions, device="cuda", generator=g) * 2 - 1)
        envs.append((env, g))
    (ea, ga), (eb, gb) = envs
    eb.load_state(ea.read_state())  # (three steps of the two kernels may already differ in an ulp)
    _eventful_state(ea, 0)
    _eventful_state(eb, 0)
    dones, soft_steps = 0, {}
    for s in range(12):
        act = torch.rand(N, ea.num_actions, device="cuda", generator=ga) * 2 - 1
        out = []
        for env in (ea, eb):
            obs, rew, term, tout, _ = env.step(act)
            out.append(dict(policy=obs["policy"].cpu().numpy().copy(), critic=obs["critic"].cpu().numpy().copy(), reward=rew.cpu().numpy().copy(),
                            terms=env.reward_terms().cpu().numpy().copy(), done=(term | tout).cpu().numpy().copy(), **env.read_state()))
        dones += int(out[0]["done"].sum())
        soft = _compare_step(s, out[0], out[1])
        if soft:
            soft_steps[s] = soft
            eb.load_state(ea.read_state())
    assert dones > N // 8  # the window is eventful
    if soft_steps:
        print(f"\n[canary] {task} sub {sub}: float fields not bit-equal between the kernel shapes (within {ROUNDOFF_REL:g}): {soft_steps}")
    # The round-off allowance is for shape pairs whose floating-point opcode counts DIFFER (another contraction of a multiply-add); the
    # build lists those (csrc/build_info.json, tools/isa_shape_arith.py).  A build that lists none compiled every pair to the same
    # arithmetic, and then anything short of bit equality is a clobbered register, not round-off: no soft step is accepted.
    import json
    import os

    info = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robot_lab_amd", "csrc", "build_info.json")
    if os.path.isfile(info) and not os.environ.get("RL_ENV_LIB"):
        with open(info) as f:
            differing = json.load(f).get("shape_pairs_with_other_float_arithmetic")
        if differing == []:
            assert not soft_steps, f"{task} sub {sub}: the build compiled every kernel-shape pair to the same arithmetic, yet {soft_steps}"
    ea.close()
    eb.close()
