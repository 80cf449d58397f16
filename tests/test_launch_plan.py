"""The launch geometry rl_env_create arrives at (include/rl_env.h rl_env_plan: no device needed) for the BASELINE robots at the sizes
the bench line and the sweeps run: the lane mapping, the workgroup shape and the LDS footprint against a CU's 160 KiB.

The one-lane-per-limb mapping's four-wavefront workgroup sits within a few KiB of that limit; in round 4 a 192-byte growth of the table
image pushed Go2 off it at 16 384 envs (99.7 -> 119 us) and only a manual sweep noticed.  This test is that sweep's assertion: table
growth that evicts a mapping fails HERE."""
import pytest

from robot_lab_amd import capi
from robot_lab_amd.scene import build_world, load_bundle

LDS_PER_CU = 160 * 1024
# task -> {envs: (lanes per limb, wavefronts per workgroup)} on 256 CUs
QUADRUPED = {64: (4, 1), 2048: (4, 1), 4096: (4, 4), 8192: (2, 4), 16384: (1, 4), 65536: (1, 4)}
TRUNK = {64: (8, 1), 1024: (8, 1), 2048: (8, 4), 4096: (8, 4), 16384: (8, 4)}
PLANS = {
    "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0": QUADRUPED,
    "RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0": QUADRUPED,
    "RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0": QUADRUPED,
    "RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0": QUADRUPED,
    "RobotLab-Isaac-Velocity-Rough-Deeprobotics-M20-v0": QUADRUPED,
    "RobotLab-Isaac-Velocity-Rough-DDTRobot-Tita-v0": QUADRUPED,  # the rot / pad quadruped instance
    "RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0": TRUNK,
    "RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0": TRUNK,
}


@pytest.mark.parametrize("task", sorted(PLANS))
def test_mapping_choice_and_lds_budget(task, monkeypatch):
    for k in ("RL_ENV_SUB", "RL_ENV_WG", "RL_ENV_MERGE", "RL_ENV_COST8"):
        monkeypatch.delenv(k, raising=False)
    desc, extra = load_bundle(task)
    build_world(desc, extra, 16, 0)
    for n, (lanes, waves) in PLANS[task].items():
        p = capi.plan(desc, n, 256)
        assert (p["lanes_per_limb"], p["wavefronts_per_workgroup"]) == (lanes, waves), (task, n, p)
        assert p["lds_bytes_per_workgroup"] <= LDS_PER_CU, (task, n, p)
        # a CU holds four wavefronts of the mapping (one per SIMD: 300 - 500 registers each): a four-wavefront workgroup as a whole, or four
        # single-wavefront ones - except the 16-lane mapping of small launches, which spread over more CUs than they fill
        if waves == 1 and n >= 4096:
            assert 4 * p["lds_bytes_per_wavefront"] <= LDS_PER_CU, (task, n, p)


def test_forced_mappings_are_reported(monkeypatch):
    desc, extra = load_bundle("RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0")
    build_world(desc, extra, 16, 0)
    for sub in (4, 2, 1):
        monkeypatch.setenv("RL_ENV_SUB", str(sub))
        monkeypatch.setenv("RL_ENV_WG", "-4")
        p = capi.plan(desc, 512, 256)
        assert p["lanes_per_limb"] == sub and p["wavefronts_per_workgroup"] == 4 and p["lds_bytes_per_workgroup"] <= LDS_PER_CU
