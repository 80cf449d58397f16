#!/bin/bash
# Round 3, GPU call: two sub-lanes per limb (8 envs per wavefront) - throughput by launch size against the other two mappings, and its
# canary / cross-shape / teacher-forced / parity tests
OUT=gpurun_out/r03j
mkdir -p $OUT
export TMPDIR=/tmp
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
for sub in 2 4 1; do
  echo "== RL_ENV_SUB=$sub" | tee -a $OUT/sweep.txt
  RL_ENV_SUB=$sub timeout 400 python tools/sweep_envs.py $A1 2048,4096,5120,6144,8192,10240,12288,16384,24576 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep.txt
done
echo "== RL_ENV_SUB=2 Go2W" | tee -a $OUT/sweep.txt
RL_ENV_SUB=2 timeout 300 python tools/sweep_envs.py RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0 4096,8192,16384 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep.txt
RL_ENV_SUB=4 timeout 300 python tools/sweep_envs.py RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0 8192 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep.txt
timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_teacher_forced.py -m gpu -q -rf -k "2] or sub2" > $OUT/pytest_sub2.log 2>&1; grep -E "passed|failed|FAILED" $OUT/pytest_sub2.log | cut -c1-300
RL_ENV_SUB=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -q -rf -k "not G1 and not Xbot and not ATOM and not Loong and not Bot and not Z1" > $OUT/pytest_sub2_parity.log 2>&1; grep -E "passed|failed|FAILED" $OUT/pytest_sub2_parity.log | cut -c1-300 | tail -12
