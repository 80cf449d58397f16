#!/bin/bash
# Round 3, GPU call: longer PPO runs (A1 Flat 1500 iterations, A1 Rough 1000)
OUT=gpurun_out/r03n
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/train_demo.py --iterations 1500 --print-every 50 --out $OUT 2>&1 | grep -v amdgpu.ids | tee $OUT/train_a1_flat.txt | tail -36
timeout 900 python tools/train_demo.py --task RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 --iterations 1000 --print-every 50 --out $OUT 2>&1 | grep -v amdgpu.ids | tee $OUT/train_a1_rough.txt | tail -26
