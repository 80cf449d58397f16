#!/usr/bin/env python
"""tests/golden/terms_extra.npz: the REFERENCE's reward functions that no shipped robot cfg gives a weight - `feet_contact`
(VEL/mdp/rewards.py:399-413), `feet_height` (rewards.py:507-524), `action_mirror` (rewards.py:281-302) and `action_sync`
(rewards.py:305-337; both with the parameters velocity_env_cfg.py:478-498 declares) - evaluated on the recorded Go2 state of
terms_go2.npz with explicit parameters, so that the oracle's restatement of them is pinned like the others (tests/test_terms_golden.py).

Run in the build container (needs /root/reference):  python tools/gen_golden_extra_terms.py
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_golden_terms as gen  # noqa: E402  (installs the shims)
from isaaclab.managers import SceneEntityCfg  # noqa: E402
from robot_lab.tasks.manager_based.locomotion.velocity import mdp  # noqa: E402
from test_terms_golden import _load  # noqa: E402


def main():
    g, desc, ora = _load("go2")
    env = gen.duck_env(ora, desc)
    feet_sensor = gen.resolve(dict(c=SceneEntityCfg("contact_forces", body_names=".*_foot")), desc)["c"]
    feet_asset = gen.resolve(dict(c=SceneEntityCfg("robot", body_names=".*_foot")), desc)["c"]
    p = dict(expect_contact_num=2, target_height=0.05, tanh_mult=2.0)
    mirror_joints = [["FR.*", "RL.*"], ["FL.*", "RR.*"]]  # velocity_env_cfg.py:483
    joint_groups = [["FR_hip_joint", "FL_hip_joint", "RL_hip_joint", "RR_hip_joint"], ["FR_thigh_joint", "FL_thigh_joint", "RL_thigh_joint", "RR_thigh_joint"],
                    ["FR_calf_joint", "FL_calf_joint", "RL_calf_joint", "RR_calf_joint"]]  # velocity_env_cfg.py:492-496
    vals = {
        "action_mirror": mdp.action_mirror(env, asset_cfg=SceneEntityCfg("robot"), mirror_joints=mirror_joints),
        "action_sync": mdp.action_sync(env, asset_cfg=SceneEntityCfg("robot"), joint_groups=joint_groups),
        "feet_contact": mdp.feet_contact(env, command_name="base_velocity", expect_contact_num=p["expect_contact_num"], sensor_cfg=feet_sensor),
        "feet_height": mdp.feet_height(env, command_name="base_velocity", asset_cfg=feet_asset, target_height=p["target_height"], tanh_mult=p["tanh_mult"]),
    }
    out = os.path.join(os.environ.get("RL_GOLDEN_DIR", os.path.join(ROOT, "tests", "golden")), "terms_extra.npz")
    np.savez(out, source="terms_go2.npz", names=np.array(list(vals)), values=np.stack([v.double().numpy() for v in vals.values()]),
             expect_contact_num=p["expect_contact_num"], target_height=p["target_height"], tanh_mult=p["tanh_mult"],
             mirror_joints=np.array(mirror_joints), joint_groups=np.array(joint_groups))
    for k, v in vals.items():
        print(k, "nonzero in", int((v != 0).sum()), "of", len(v), "envs; max", float(v.abs().max()))


if __name__ == "__main__":
    main()
