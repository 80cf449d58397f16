#!/bin/bash
# Round 3: G1 step kernel under the back end's other scheduling strategies (one-call A/B against the shipped build)
OUT=gpurun_out/r03y
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
timeout 600 python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 --rounds 2 $V/kin_dealt_74.so $V/g1_ilp_74.so $V/g1_memclause_74.so $V/g1_metric0_74.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_g1_sched.txt
