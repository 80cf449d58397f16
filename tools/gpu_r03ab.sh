#!/bin/bash
# Round 3: self-collision pass after the pair cull (48 pairs = 3 trips, batched bounding tests): cost on / off, its GPU test
OUT=gpurun_out/r03ab
mkdir -p $OUT
export TMPDIR=/tmp
LIB=robot_lab_amd/csrc/librl_env_hip.so
for t in RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0; do
  echo "# self-collision pass ON"; timeout 300 python tools/ab_bench.py --task $t --num-envs 2048 --rounds 2 $LIB 2>&1 | grep -v amdgpu.ids
  echo "# RL_ENV_SELF=0"; RL_ENV_SELF=0 timeout 300 python tools/ab_bench.py --task $t --num-envs 2048 --rounds 2 $LIB 2>&1 | grep -v amdgpu.ids
done | tee $OUT/ab_self_collision.txt
timeout 600 python -m pytest tests/test_gpu_self_collision.py -m gpu -q -rf > $OUT/pytest_self.log 2>&1; echo "rc=$?" >> $OUT/pytest_self.log
grep -E "passed|failed|FAILED|rc=|^E " $OUT/pytest_self.log | cut -c1-300 | tail -12
