"""Symmetry data augmentation (include/rl_rollout.h rl_symmetry_*): oracle and host tables against the reference's own
outputs (tests/golden/symmetry_anymal.npz, generated from mdp/symmetry/anymal.py by tools/gen_golden_symmetry.py) on
CPU; the HIP kernel against the same fixture, bit for bit, on the GPU."""
import ctypes
import os

import numpy as np
import pytest

from oracle.symmetry import compute_symmetric_states as oracle_css
from robot_lab_amd.rollout import ROLLOUT_LIB
from robot_lab_amd.symmetry import ACTION_LAYOUT, ANYMAL_JOINTS, POLICY_LAYOUT, SYMMETRY_EXPORTS, joint_tables, layout_tables

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "symmetry_anymal.npz"))


def test_oracle_matches_reference_outputs():
    o, a = oracle_css(GOLD["obs"], GOLD["actions"])
    np.testing.assert_array_equal(o.astype(np.float32), GOLD["obs_aug"])
    np.testing.assert_array_equal(a.astype(np.float32), GOLD["actions_aug"])


def test_host_tables_reproduce_reference_outputs():
    for layout, x, want in ((POLICY_LAYOUT, GOLD["obs"], GOLD["obs_aug"]), (ACTION_LAYOUT, GOLD["actions"], GOLD["actions_aug"])):
        perm, sign = layout_tables(layout)
        assert perm.shape == sign.shape == (4, x.shape[1]) and perm.dtype == np.int32 and sign.dtype == np.float32
        got = np.concatenate([sign[s][None] * x[:, perm[s]] for s in range(4)], 0)
        np.testing.assert_array_equal(got, want)
        for s in range(4):  # every copy is an involution (a mirror, or the product of two commuting mirrors)
            np.testing.assert_array_equal(sign[s] * (sign[s][None] * x[:, perm[s]])[:, perm[s]], x)


def test_joint_tables_from_names():
    (lrp, lrs), (fbp, fbs) = joint_tables(ANYMAL_JOINTS)
    assert [ANYMAL_JOINTS[i] for i in lrp[:4]] == ["RF_HAA", "RH_HAA", "LF_HAA", "LH_HAA"] and list(lrs[:4]) == [-1] * 4 and list(lrs[4:]) == [1] * 8
    assert [ANYMAL_JOINTS[i] for i in fbp[4:8]] == ["LH_HFE", "LF_HFE", "RH_HFE", "RF_HFE"] and list(fbs[:4]) == [1] * 4 and list(fbs[4:]) == [-1] * 8
    with pytest.raises(ValueError):
        joint_tables(["FR_hip_joint"])


def test_symmetry_exports():
    lib = ctypes.CDLL(ROLLOUT_LIB)
    for name in SYMMETRY_EXPORTS:
        assert hasattr(lib, name), name


@pytest.mark.gpu
def test_hip_symmetry_matches_reference_outputs_bitwise():
    import torch

    from robot_lab_amd.symmetry import SymmetryAugmentation, compute_symmetric_states

    obs, act = torch.from_numpy(GOLD["obs"]).cuda(), torch.from_numpy(GOLD["actions"]).cuda()
    critic = torch.randn(obs.shape[0], 48, device="cuda:0")
    o, a = compute_symmetric_states(None, {"policy": obs, "critic": critic}, act)
    np.testing.assert_array_equal(o["policy"].cpu().numpy(), GOLD["obs_aug"])
    np.testing.assert_array_equal(a.cpu().numpy(), GOLD["actions_aug"])
    assert torch.equal(o["critic"], critic.repeat(4, 1))
    assert compute_symmetric_states(None, None, None) == (None, None)
    # a mini-batch of a real iteration: 24 x 4096 / 4 rows
    big = torch.randn(24576, 45, device="cuda:0")
    perm, sign = layout_tables(POLICY_LAYOUT)
    aug = SymmetryAugmentation(perm, sign)
    want = torch.cat([torch.from_numpy(sign[s]).cuda()[None] * big[:, torch.from_numpy(perm[s]).long().cuda()] for s in range(4)], 0)
    assert torch.equal(aug(big), want)
    aug.close()
