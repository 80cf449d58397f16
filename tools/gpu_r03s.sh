#!/bin/bash
OUT=gpurun_out/r03s
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/train_demo.py --task RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0 --num-envs 4096 --iterations 600 --print-every 50 --out $OUT 2>&1 | grep -v amdgpu.ids | tee $OUT/train_g1_diag.txt | tail -16
timeout 300 python tools/train_demo.py --iterations 200 --print-every 50 --out $OUT 2>&1 | grep -v amdgpu.ids | tail -7
