#!/bin/bash
# Round 4, call m: the per-joint constants of the lane tables packed as 16-byte vectors (jc_*) against the tree before (base_* = 4f0c1a5), one call.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/gpu_r04m.sh'
TAG=r04m
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 --rounds 3 $V/base_78.so $V/jc_78.so 2>&1 | grep -v amdgpu.ids | tee $OUT/jc_ab.txt
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0 --num-envs 2048 --rounds 3 $V/base_2078.so $V/jc_2078.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/jc_ab.txt
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 --num-envs 4096 --rounds 3 $V/base_34.so $V/jc_34.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/jc_ab.txt
