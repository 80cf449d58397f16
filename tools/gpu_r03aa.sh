#!/bin/bash
# Round 3: self-collision pass of the trunk + limbs instance - cost (on / off in one call), its GPU parity test, the G1 / ATOM01 suites
OUT=gpurun_out/r03aa
mkdir -p $OUT
export TMPDIR=/tmp
LIB=robot_lab_amd/csrc/librl_env_hip.so
for t in RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 RobotLab-Isaac-Velocity-Rough-RoboParty-ATOM01-v0; do
  echo "# self-collision pass ON"; timeout 300 python tools/ab_bench.py --task $t --num-envs 2048 --rounds 2 $LIB 2>&1 | grep -v amdgpu.ids
  echo "# RL_ENV_SELF=0"; RL_ENV_SELF=0 timeout 300 python tools/ab_bench.py --task $t --num-envs 2048 --rounds 2 $LIB 2>&1 | grep -v amdgpu.ids
done | tee $OUT/ab_self_collision.txt
RL_ENV_DEBUG=1 timeout 120 python tools/pcs_run.py RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 2048 5 2>&1 | grep -i "rl_env\|lds" | head -5
timeout 900 python -m pytest tests/test_gpu_self_collision.py tests/test_gpu_canary.py tests/test_gpu_teacher_forced.py tests/test_gpu_episode_stats.py tests/test_gpu_parity.py tests/test_gpu_all_tasks.py -m gpu -q -rf -k "G1 or ATOM01 or self or canary or shapes or all_tasks or every" > $OUT/pytest_self.log 2>&1; echo "rc=$?" >> $OUT/pytest_self.log
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_self.log | cut -c1-300 | tail -12
