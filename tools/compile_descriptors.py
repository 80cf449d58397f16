#!/usr/bin/env python
"""Compile the reference's task cfg classes into descriptor bundles (robot_lab_amd/data/*.json).

Run in the build container (needs /root/reference):  python tools/compile_descriptors.py
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from robot_lab_amd import shims  # noqa: E402

shims.install(shims.REFERENCE_SOURCE)
import gymnasium as gym  # noqa: E402
import robot_lab.tasks  # noqa: E402,F401
from isaaclab_tasks.utils import parse_env_cfg  # noqa: E402

from robot_lab_amd.model.cfg_compile import UnsupportedTerm, compile_cfg  # noqa: E402
from robot_lab_amd.scene import DATA_DIR, save_bundle  # noqa: E402

# the BASELINE.json robots + the other quadrupeds of the reference whose cfg compiles to the same lane-program
# instances as data: 3-joint legs -> Topo<3,0,3,6>, wheeled 4-joint legs -> Topo<4,0,3,6>
# humanoids / bipeds that fit "trunk of <= 3 joints + <= 4 limb chains of <= 7 joints" run on the G1 instance
# (Topo<7,3,4,9>) with inert padding joints / empty limbs: ATOM01 (23 DoF, 1 waist joint), Xbot (28 DoF, legs hang off
# a 2-joint trunk), MagicLab Bot-Gen1 (14 DoF), Openloong Loong (12-DoF biped, two empty limbs)
# ... MagicLab Z1 (14 DoF, hip joints listed out of tree order), DDT Tita (2 wheeled legs, 2 empty limbs)
ROBOTS = ("Unitree-A1", "Unitree-Go2", "Unitree-Go2W", "Unitree-G1", "Unitree-B2", "Deeprobotics-Lite3", "Deeprobotics-M20",
          "Zsibot-ZSL1", "Zsibot-ZSL1W", "RoboParty-ATOM01", "RobotEra-Xbot", "MagicLab-Bot-Gen1", "Openloong-Loong",
          "Unitree-B2W", "MagicLab-Dog-W", "MagicLab-Dog", "MagicLab-Bot-Z1", "DDTRobot-Tita", "HandStand-Unitree-A1", "Agibot-D1",
          "FFTAI-GR1T1", "FFTAI-GR1T2",  # GR1: a six-joint spine (waist + head) with the arms leaving it at depth 3: Topo<7,6,4,9>
          "Booster-T1")                  # T1: waist (carrying the legs) and a two-joint neck both on the trunk body - a trunk of two pieces
# not compiled: Unitree-H1 (asset lives in isaaclab_assets, not in the reference), MagicLab-Dog Rough (its registration names a class that does not exist)
TASKS = sys.argv[1:] or [f"RobotLab-Isaac-Velocity-{t}-{r}-v0" for r in ROBOTS for t in ("Flat", "Rough")]
os.makedirs(DATA_DIR, exist_ok=True)
for task in TASKS:
    try:
        cfg = parse_env_cfg(task, device="cpu")
        desc, spec = compile_cfg(cfg)
    except (UnsupportedTerm, NotImplementedError, AttributeError) as e:  # AttributeError: a registration that names a class the module does not have (MagicLab-Dog Rough upstream)
        print(f"{task}: NOT COMPILED ({e})")
        continue
    if desc.model.num_chains == 0:
        print(f"{task}: NOT COMPILED (topology is not a trunk of <= 6 joints + <= 4 serial limbs of <= 7 joints)")
        continue
    save_bundle(os.path.join(DATA_DIR, task + ".json"), desc, spec)
    m = desc.model
    print(f"{task}: links={m.num_links} dof={m.num_dof} bodies={m.num_bodies} spheres={m.num_spheres} "
          f"chains={m.num_chains}x{m.chain_len} rewards={desc.task.n_rewards} obs={desc.obs_dim(0)}/{desc.obs_dim(1)}")
