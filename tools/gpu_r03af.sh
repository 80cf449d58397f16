#!/bin/bash
# Round 3: the six-joint-spine instance Topo<7,6,4,9> (FFTAI GR1T1 / GR1T2) on the GPU: parity, canaries, every GR1 bundle stepped, timing
OUT=gpurun_out/r03af
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_canary.py tests/test_gpu_all_tasks.py tests/test_gpu_self_collision.py -m gpu -q -rf -k "GR1 or self" > $OUT/pytest_gr1.log 2>&1; echo "rc=$?" >> $OUT/pytest_gr1.log
grep -E "passed|failed|FAILED|rc=|^E " $OUT/pytest_gr1.log | cut -c1-300 | tail -12
RL_ENV_DEBUG=1 timeout 200 python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0 --num-envs 2048 --rounds 1 robot_lab_amd/csrc/librl_env_hip.so 2>&1 | grep -v amdgpu.ids | tee $OUT/gr1_timing.txt | tail -3
