#!/bin/bash
# Round 3: G1 Velocity-Flat PPO with the reference's cfg again, now WITH the self-collision pass (profiles/r03r: without it the policy sits down)
OUT=gpurun_out/r03ac
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/train_demo.py --task RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0 --num-envs 4096 --iterations 1500 --print-every 100 --out $OUT 2>&1 | grep -v amdgpu.ids | tee $OUT/train_g1_flat_selfcol.txt | tail -22 | cut -c1-250
