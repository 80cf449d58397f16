#!/bin/bash
# Round 3: every compiled task id through env.step() (table), and PPO runs on the other lane-program instances (Go2W: merged wheeled; G1: trunk + limbs)
OUT=gpurun_out/r03p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/bench_every_task.py 2>&1 | grep -v amdgpu.ids | tee $OUT/all_tasks.txt | tail -45
timeout 600 python tools/train_demo.py --task RobotLab-Isaac-Velocity-Flat-Unitree-Go2W-v0 --iterations 500 --print-every 50 --out $OUT 2>&1 | grep -v amdgpu.ids | tee $OUT/train_go2w_flat.txt | tail -14
timeout 600 python tools/train_demo.py --task RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0 --num-envs 2048 --iterations 500 --print-every 50 --out $OUT 2>&1 | grep -v amdgpu.ids | tee $OUT/train_g1_flat.txt | tail -14
