#!/bin/bash
# Round 4, call o: the lane scratchpad in 16-byte granules (gran_*: sensor rows and contact stash read / written as ds_read_b128 / ds_write_b128) against the tree before (base_*), one call.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/gpu_r04o.sh'
TAG=r04o
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 $V/base_34.so $V/gran_34.so 2>&1 | grep -v amdgpu.ids | tee $OUT/gran_ab.txt
RL_ENV_SUB=2 python tools/ab_bench.py --task $A1 --num-envs 8192 --rounds 3 $V/base_32.so $V/gran_32.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gran_ab.txt
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 --rounds 3 $V/base_78.so $V/gran_78.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gran_ab.txt
