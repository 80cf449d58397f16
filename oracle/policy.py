"""TEST INFRASTRUCTURE ONLY (oracle).  fp64 numpy restatement of the MLP the reference's PPO configs describe
(`.../unitree_a1/agents/rsl_rl_ppo_cfg.py:15-22`: Linear + ELU stacks; the arithmetic itself is rsl_rl / torch,
third-party and absent from /root/reference).  Pinned in tests/test_policy.py against `torch.nn.Sequential` of the
same layers - torch is the library the reference would run."""
import numpy as np


def mlp_forward(x, weights, biases, activation="elu"):
    """weights[l]: [out, in] (nn.Linear layout); activation after every layer but the last."""
    h = np.asarray(x, dtype=np.float64)
    for l, (w, b) in enumerate(zip(weights, biases)):
        h = h @ np.asarray(w, dtype=np.float64).T + np.asarray(b, dtype=np.float64)
        if l < len(weights) - 1:
            if activation == "elu":
                h = np.where(h > 0, h, np.expm1(np.minimum(h, 0.0)))
            elif activation == "relu":
                h = np.maximum(h, 0.0)
            else:
                h = np.tanh(h)
    return h
