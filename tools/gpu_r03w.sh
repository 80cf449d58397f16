#!/bin/bash
OUT=gpurun_out/r03w
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/train_demo.py --task RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0 --num-envs 4096 --iterations 1200 --print-every 100 --out $OUT --also-terminate-on "pelvis|.*hip.*|.*knee.*|.*shoulder.*|.*elbow.*|waist.*" 2>&1 | grep -v amdgpu.ids | tee $OUT/train_g1_diag_terminate.txt | tail -18 | cut -c1-250
