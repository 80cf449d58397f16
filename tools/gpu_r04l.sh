#!/bin/bash
# Round 4, call l (after the closing call): in-kernel phase clocks of the final A1 / G1 step kernels and of the G1 kernel with the
# row-distributed elimination (where do the 1.2 k saved instructions go?), and the collection loop on the final tree.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/gpu_r04l.sh'
TAG=r04l
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
G1=RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0
RL_ENV_LIB=$V/clock_34.so python tools/phase_clock.py $A1 4096 2>&1 | grep -v amdgpu.ids | tee $OUT/phase_clock_a1.txt
RL_ENV_LIB=$V/clock_78.so python tools/phase_clock.py $G1 2048 2>&1 | grep -v amdgpu.ids | tee $OUT/phase_clock_g1.txt
RL_ENV_LIB=$V/clockrows_78.so python tools/phase_clock.py $G1 2048 2>&1 | grep -v amdgpu.ids | tee $OUT/phase_clock_g1_rows.txt
python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 1 robot_lab_amd/csrc/librl_env_hip.so $V/clock_78.so $V/clockrows_78.so 2>&1 | grep -v amdgpu.ids | tee $OUT/clock_overhead.txt
python tools/bench_collect.py $A1 4096 40 2>&1 | grep -v amdgpu.ids | tee $OUT/collect.txt
python tools/bench_collect.py $G1 2048 20 2>&1 | grep -v amdgpu.ids | tee -a $OUT/collect.txt
