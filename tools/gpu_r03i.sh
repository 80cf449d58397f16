#!/bin/bash
# Round 3, ninth GPU call: one lane per limb - link records built where the recursion consumes them (serial) against all records up front (prev)
OUT=gpurun_out/r03i
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
GO2W=RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0
for n in 16384 65536; do
  RL_ENV_SUB=1 python tools/ab_bench.py --task $A1 --num-envs $n --rounds 2 $V/prev_31.so $V/serial_31.so $V/serialb_31.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
RL_ENV_SUB=1 python tools/ab_bench.py --task $GO2W --num-envs 16384 --rounds 1 $V/prev_1041.so $V/serial_1041.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 1 $V/prev_34.so $V/cur_34.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
