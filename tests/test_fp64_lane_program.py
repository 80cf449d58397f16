"""The lane program in DOUBLE precision against the fp64 oracle (VERDICT r5 "next round" item 1).

The fp32 tiers (tests/test_teacher_forced.py, tests/test_gpu_teacher_forced.py) bound HIP-vs-oracle by an envelope of the oracle's own
fp32 sensitivity, because only 6 - 60 % of the joint-velocity entries sit inside a flat 1e-5.  That leaves a question open: is the gap
round-off of a stiff step evaluated in fp32, or a small ALGORITHMIC difference between the articulated-body recursion in base coordinates
(csrc/env_step.h) and the oracle's dense solve in link coordinates (oracle/physics.py) hiding inside the envelope?  Here the same lane
program source, retyped float -> double by tests/emu/make_f64.py and run on the CPU lane emulator, steps once from a shared eventful
state: every field must agree with the oracle to 1e-9 relative (measured: <= 3e-12), dones / episode lengths / terrain levels / contact
timers exactly, with NO switch mask needed at that precision beyond the oracle's own margins.  So the fp32 gap is round-off.

What this tier found when it was written (round 6), all invisible inside the fp32 envelopes:
* the oracle did not normalise the fp32 link quaternions / joint axes of the descriptor, the host tables did: 1e-8 on G1 / GR1;
* a real defect of the trunk + limbs lane program (csrc/env_step.h substep_aba_trunk): the record of a trunk-link share with an active
  contact carried the rotational inertia of limb link 0 onto the trunk link - ~1 % of the contact forces of a humanoid lying on its
  torso, hidden from the eventful-step comparison because such an env terminates (illegal contact) and is reset inside the compared
  step: `test_fp64_robot_on_the_ground` switches the termination off and compares the state after the physics."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
T = "RobotLab-Isaac-Velocity-Rough-%s-v0"

# (task, envs, warm-up steps, lanes per limb, RL_ENV_SPEC): every lane-program instance in the mapping production runs it in, the four
# BASELINE tasks through their task-specialised term stack (spec_id > 0 is asserted) and through the interpreter
CASES = [
    ("Unitree-A1", 48, 8, 4, 1), ("Unitree-A1", 32, 6, 1, 0), ("Unitree-A1", 32, 6, 2, 0),
    ("Unitree-Go2", 32, 6, 4, 1),
    ("Unitree-Go2W", 32, 6, 4, 1), ("Unitree-Go2W", 16, 6, 1, 0),
    ("Unitree-G1", 8, 6, 8, 1), ("Unitree-G1", 8, 6, 1, 0),
    ("FFTAI-GR1T1", 8, 6, 8, 0),        # six trunk joints, tilted joint axes
    ("Booster-T1", 8, 6, 8, 0),         # a trunk of two pieces
    ("Unitree-B2W", 16, 6, 4, 0),       # unmerged 4-joint instance
    ("Deeprobotics-M20", 16, 6, 4, 0),  # merged: trunk spheres in the wheel groups
    ("DDTRobot-Tita", 16, 4, 4, 0),          # rotated frames + padding joints on the quadruped instance
]
FLOAT_FIELDS = ("root_state", "joint_pos", "joint_vel", "task_state", "gains", "episode_sums", "obs_policy", "obs_critic", "reward", "reward_terms")


@pytest.fixture(scope="module")
def f64_lib():
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import make_f64

    return make_f64.build()


def run_case(robot, n, k, sub, spec):
    env = dict(os.environ, RL_ABI_REAL="f64", RL_ENV_SPEC=str(spec))
    out = subprocess.run([sys.executable, os.path.join(HERE, "fp64_lane_program.py"), T % robot, str(n), str(k), str(sub)], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("FP64_REPORT ")][-1]
    return json.loads(line[len("FP64_REPORT "):])


@pytest.mark.parametrize("robot,n,k,sub,spec", CASES)
def test_fp64_lane_program_matches_oracle(robot, n, k, sub, spec, f64_lib):
    rep = run_case(robot, n, k, sub, spec)
    assert rep["real_bytes"] == 8
    assert rep["done_count"] > 0  # the compared step resets somebody
    assert (rep["spec_id"] > 0) == bool(spec), rep["spec_id"]
    assert rep["masked"] <= max(1, n // 8), rep
    worst = {f: rep["fields"][f]["max_err"] for f in FLOAT_FIELDS}
    assert all(v <= 1e-9 for v in worst.values()), worst
    assert rep["fields"]["contact_timers"]["max_err"] <= 1e-12
    assert rep["done_equal"] and rep["episode_length_equal"] and rep["terrain_level_equal"], rep


GROUND = [("Booster-T1", 16, 8), ("Unitree-G1", 16, 8), ("Unitree-G1", 16, 1), ("FFTAI-GR1T1", 16, 8), ("Unitree-A1", 16, 4), ("Unitree-Go2W", 16, 4),
          ("Deeprobotics-M20", 16, 2), ("DDTRobot-Tita", 16, 1)]


@pytest.mark.parametrize("robot,n,sub", GROUND)
def test_fp64_robot_on_the_ground(robot, n, sub, f64_lib):
    """Robots on their backs / faces, terminations off (tests/fp64_lane_program.py on_the_ground): three steps, every field and the
    contact sensor's net forces per body."""
    rep = run_case(robot, n, -1, sub, 0)
    assert rep["envs_with_a_trunk_body_loaded"] >= 2, rep  # the scenario does load the trunk
    worst = {f: rep["fields"][f]["max_err"] for f in FLOAT_FIELDS}
    assert all(v <= 1e-9 for v in worst.values()), worst
    assert rep["fields"]["contact_force_abs"]["max_err"] <= 1e-6, rep["fields"]["contact_force_abs"]  # N, forces of 10^2 - 10^3 N
    assert rep["fields"]["contact_timers"]["max_err"] <= 1e-12
    assert rep["done_equal"] and rep["episode_length_equal"] and rep["terrain_level_equal"], rep
