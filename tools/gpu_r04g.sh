#!/bin/bash
# Round 4, A/B call: the 16-lane A1 kernel with its 3-slot contact stash in LDS (shipped) vs in registers (-DRL_STASH_REG_QUAD), one call.
TAG=r04g
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 --num-envs 4096 --rounds 3 $V/base_34.so $V/stashreg_34.so 2>&1 | grep -v amdgpu.ids | tee $OUT/a1_stash_ab.txt
