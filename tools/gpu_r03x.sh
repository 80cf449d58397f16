#!/bin/bash
# Round 3: G1 kinematics with the local joint transforms dealt to the limb's sub-lanes - A/B against the replicated form in one call, then
# the trunk + limbs instance's parity suites on the new build
OUT=gpurun_out/r03x
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
timeout 400 python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 --rounds 2 $V/kin_repl_74.so $V/kin_dealt_74.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_g1_kinematics.txt
timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_teacher_forced.py tests/test_gpu_episode_stats.py tests/test_gpu_parity.py -m gpu -q -rf -k "G1 or Xbot or ATOM01 or Tita or Loong or Gen1 or Z1 or canary or shapes" > $OUT/pytest_trunk.log 2>&1; echo "rc=$?" >> $OUT/pytest_trunk.log
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_trunk.log | cut -c1-300 | tail -12
