#!/bin/bash
# Round 4, eighth call: the row-distributed elimination of the trunk + limbs instances (each lane of a DPP quad carries rows of the
# 6 x 6 link records) against the replicated form (-DRL_ELIM_REPLICATED, later renamed: the rows are -DRL_ELIM_ROWS) and the tree before it, one call; then the trunk-robot parity
# subset and timings on the shipped library.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r04h.sh'
TAG=r04h
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 --rounds 3 $V/base_78.so $V/rep_78.so $V/dist_78.so 2>&1 | grep -v amdgpu.ids | tee $OUT/elim_ab.txt
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0 --num-envs 2048 --rounds 3 $V/rep_2078.so $V/dist_2078.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/elim_ab.txt
RL_ENV_DEBUG=1 python tools/sweep_envs.py RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0 2048 2>&1 | grep "rl_env:" | head -2 | tee $OUT/lds_choice.txt
RL_ENV_DEBUG=1 python tools/sweep_envs.py RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 2048 2>&1 | grep "rl_env:" | head -2 | tee -a $OUT/lds_choice.txt
timeout 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_canary.py tests/test_gpu_self_collision.py tests/test_gpu_teacher_forced.py tests/test_gpu_lane_mapping.py -m gpu -q -x > $OUT/pytest_subset.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -6 $OUT/pytest_subset.log
for t in Rough-Unitree-G1 Rough-FFTAI-GR1T1 Rough-Booster-T1 Rough-RobotEra-Xbot; do python tools/sweep_envs.py RobotLab-Isaac-Velocity-$t-v0 2048,4096 2>&1 | grep -v amdgpu.ids | tee -a $OUT/trunk_sweep.txt; done
