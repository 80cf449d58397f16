#!/bin/bash
# Round 4, tenth call: one-call A/Bs on the tree where the one-lane-per-limb mapping addresses its state tiles as buffers:
#   bufall = buffers in every mapping (-DRL_STATE_BUF_ALL; divergent field indices in the vector offset - no waterfall loops this time),
#   iregs  = per-step rigid-body constants held in registers over the substeps (-DRL_INERTIA_REGS).
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r04j.sh'
TAG=r04j
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
G1=RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0
python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 $V/base_34.so $V/bufall_34.so $V/iregs_34.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
RL_ENV_SUB=2 python tools/ab_bench.py --task $A1 --num-envs 8192 --rounds 3 $V/base_32.so $V/bufall_32.so $V/iregs_32.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 3 $V/base_78.so $V/bufall_78.so $V/iregs_78.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
