"""Host logic of the hipGraph protocol (include/rl_env.h `rl_env_graph_*`): the kernels take the step count as
*device word + launch literal, a capture rolls the host mirrors back and appends the node that advances the word.

On the CPU lane emulator a "captured" launch simply runs, i.e. a capture behaves like the capture plus ONE replay - which is
enough to pin the counter algebra: an env driven through begin / steps / end / launching must stay bit-identical to one that
took the same steps directly (every random stream is keyed by the step count).  The real capture + replay is
tests/test_gpu_collect.py."""
import numpy as np
import pytest

from helpers import host_view, make_pair

TASK = "RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0"


def _native(emu_lib, seed=5, N=8):
    desc, ora, nat = make_pair(TASK, N, seed, emu_lib)
    nat.reset()
    return nat


def _snap(nat):
    nat.export_state()
    slot = nat.obs_slot()
    return {k: host_view(nat, k).copy() for k in ("ROOT_STATE", "JOINT_POS", "JOINT_VEL", "REWARD", "COMMAND", "EPISODE_LENGTH")} | {
        "obs": host_view(nat, "OBS_POLICY_RING")[slot].copy(), "count": nat.step_count, "log_slot": nat.log_slot(), "obs_slot": slot}


def test_capture_and_direct_steps_agree(emu_lib):
    a, b = _native(emu_lib), _native(emu_lib)
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, (9, 8, a.num_actions)).astype(np.float32)
    for nat in (a, b):  # a few direct steps first: the anchor is not zero
        for k in range(3):
            nat.step(acts[k].ctypes.data)
    # a: steps 3, 4 inside a "capture" (the emulator runs them), then the replay bookkeeping
    a.graph_begin()
    a.step(acts[3].ctypes.data)
    a.step(acts[4].ctypes.data)
    mid = a.step_count
    assert a.graph_end() == 2
    assert mid == 5 and a.step_count == 3, "the capture ran nothing as far as the host mirrors go"
    a.graph_launching()
    assert a.step_count == 5
    b.step(acts[3].ctypes.data)
    b.step(acts[4].ctypes.data)
    sa, sb = _snap(a), _snap(b)
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    # direct steps after a replay take their literal from the moved anchor
    for nat in (a, b):
        nat.step(acts[5].ctypes.data)
        nat.step(acts[6].ctypes.data)
    sa, sb = _snap(a), _snap(b)
    for k in sa:
        np.testing.assert_array_equal(sa[k], sb[k], err_msg=k)
    # a second "replay" after direct steps re-anchors the device word first
    a.graph_launching()
    assert a.step_count == 9
    a.close()
    b.close()


def test_protocol_errors(emu_lib):
    from robot_lab_amd.capi import RlEnvError

    nat = _native(emu_lib)
    act = np.zeros((8, nat.num_actions), dtype=np.float32)
    with pytest.raises(RlEnvError, match="without rl_env_graph_begin"):
        nat.graph_end()
    with pytest.raises(RlEnvError, match="no captured loop"):
        nat.graph_launching()
    nat.graph_begin()
    with pytest.raises(RlEnvError, match="already open"):
        nat.graph_begin()
    nat.step(act.ctypes.data)
    with pytest.raises(RlEnvError, match="even, positive number"):
        nat.graph_end()  # one step: the observation buffers would not line up on replay
    assert nat.step_count == 0
    nat.graph_begin()
    nat.step(act.ctypes.data)
    nat.step(act.ctypes.data)
    assert nat.graph_end() == 2
    nat.step(act.ctypes.data)  # odd number of direct steps since the capture: the graph's literal pointers are the other slot's
    with pytest.raises(RlEnvError, match="observation slot parity"):
        nat.graph_launching()
    nat.step(act.ctypes.data)
    nat.graph_launching()
    nat.close()
