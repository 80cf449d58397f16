#!/bin/bash
OUT=gpurun_out/r03r
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python tools/train_demo.py --task RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0 --num-envs 4096 --iterations 3000 --print-every 100 --out $OUT 2>&1 | grep -v amdgpu.ids | tee $OUT/train_g1_flat_3000.txt | tail -36
