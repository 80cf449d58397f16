#!/bin/bash
# Round 4, call r: mapping choice and LDS need of the quadruped instances at 8192 / 16384 envs on the final tree (RL_ENV_DEBUG prints the
# cost model's inputs), with the step time of each.
#   /usr/local/graft/bin/gpurun --timeout 300 -- 'bash tools/gpu_r04r.sh'
TAG=r04r
OUT=gpurun_out/$TAG
mkdir -p $OUT
for t in Unitree-A1 Unitree-Go2 Unitree-Go2W Unitree-B2 Unitree-B2W Deeprobotics-Lite3 Deeprobotics-M20 DDTRobot-Tita; do
  RL_ENV_DEBUG=1 timeout 60 python tools/sweep_envs.py RobotLab-Isaac-Velocity-Rough-$t-v0 8192,16384 2>&1 | grep -v amdgpu.ids | grep "rl_env: .*envs:\|^ *[0-9]\|^Robot" | tee -a $OUT/quadruped_mappings.txt
done
