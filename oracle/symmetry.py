"""TEST INFRASTRUCTURE ONLY (oracle).  numpy restatement of the reference's symmetry data augmentation for ANYmal,
`source/robot_lab/robot_lab/tasks/manager_based/locomotion/velocity/mdp/symmetry/anymal.py` - pinned to that file's own
outputs by tests/golden/symmetry_anymal.npz (tools/gen_golden_symmetry.py imports it unchanged).

Joint order of the ANYmal articulation (anymal.py:216-229): LF, LH, RF, RH for HAA (0..3), HFE (4..7), KFE (8..11).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module."""
import numpy as np

LF, LH, RF, RH = (0, 4, 8), (1, 5, 9), (2, 6, 10), (3, 7, 11)


def switch_left_right(j):  # anymal.py:232-244
    out = np.zeros_like(j)
    out[..., LF + LH] = j[..., RF + RH]
    out[..., RF + RH] = j[..., LF + LH]
    out[..., [0, 1, 2, 3]] *= -1.0  # HAA
    return out


def switch_front_back(j):  # anymal.py:247-259
    out = np.zeros_like(j)
    out[..., LF + RF] = j[..., LH + RH]
    out[..., LH + RH] = j[..., LF + RF]
    out[..., 4:] *= -1.0  # HFE, KFE
    return out


def policy_obs(obs, which):
    """anymal.py:97-127 (left-right) / :130-160 (front-back) on the 45-column policy observation
    [ang vel 3 | projected gravity 3 | velocity command 3 | joint pos 12 | joint vel 12 | last action 12]."""
    o = np.array(obs, dtype=np.float64, copy=True)
    sw = switch_left_right if which == "lr" else switch_front_back
    sg = dict(lr=([-1, 1, -1], [1, -1, 1], [1, -1, -1]), fb=([1, -1, -1], [-1, 1, 1], [-1, 1, -1]))[which]
    o[:, 0:3] *= sg[0]
    o[:, 3:6] *= sg[1]
    o[:, 6:9] *= sg[2]
    for a in (9, 21, 33):
        o[:, a:a + 12] = sw(o[:, a:a + 12])
    return o


def compute_symmetric_states(obs=None, actions=None):
    """anymal.py:27-87: [original, left-right, front-back, front-back of left-right] stacked on the batch axis."""
    o = a = None
    if obs is not None:
        lr = policy_obs(obs, "lr")
        o = np.concatenate([np.asarray(obs, dtype=np.float64), lr, policy_obs(obs, "fb"), policy_obs(lr, "fb")], 0)
    if actions is not None:
        x = np.asarray(actions, dtype=np.float64)
        lr = switch_left_right(x)
        a = np.concatenate([x, lr, switch_front_back(x), switch_front_back(lr)], 0)
    return o, a
