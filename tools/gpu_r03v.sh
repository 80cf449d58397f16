#!/bin/bash
# Round 3: soak of the final build - 3000 random-action steps of the BASELINE robots (and of the 8-lane / one-lane-per-limb mappings)
OUT=gpurun_out/r03v
mkdir -p $OUT
export TMPDIR=/tmp
for t in A1 Go2 Go2W; do echo "== $t Rough 4096" | tee -a $OUT/soak.txt; timeout 300 python tools/soak.py RobotLab-Isaac-Velocity-Rough-Unitree-$t-v0 3000 random 4096 2>&1 | grep -v amdgpu | tail -2 | tee -a $OUT/soak.txt; done
echo "== G1 Rough 2048" | tee -a $OUT/soak.txt; timeout 400 python tools/soak.py RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 3000 random 2048 2>&1 | grep -v amdgpu | tail -2 | tee -a $OUT/soak.txt
echo "== A1 Rough 8192 (8 lanes per env)" | tee -a $OUT/soak.txt; timeout 300 python tools/soak.py RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 3000 random 8192 2>&1 | grep -v amdgpu | tail -2 | tee -a $OUT/soak.txt
echo "== A1 Rough 16384 (one lane per limb)" | tee -a $OUT/soak.txt; timeout 300 python tools/soak.py RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 3000 random 16384 2>&1 | grep -v amdgpu | tail -2 | tee -a $OUT/soak.txt
echo "== A1 Rough 4096 zero actions" | tee -a $OUT/soak.txt; timeout 300 python tools/soak.py RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 1500 zero 4096 2>&1 | grep -v amdgpu | tail -1 | tee -a $OUT/soak.txt
