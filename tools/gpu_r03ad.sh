#!/bin/bash
# Round 3: self-collision test moved ahead of the contact stage (overlapping its terrain gathers, no third sync) vs behind it
OUT=gpurun_out/r03ad
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
timeout 400 python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 --rounds 2 $V/self_old_74.so $V/self_new_74.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_self_placement.txt
RL_ENV_SELF=0 timeout 400 python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 --rounds 1 $V/self_new_74.so 2>&1 | grep -v amdgpu.ids | sed 's/^/RL_ENV_SELF=0 /' | tee -a $OUT/ab_self_placement.txt
RL_ENV_LIB=$V/self_new_74.so timeout 300 python -m pytest tests/test_gpu_self_collision.py -m gpu -q 2>&1 | tail -2
