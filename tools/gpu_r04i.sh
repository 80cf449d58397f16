#!/bin/bash
# Round 4, ninth call: the HBM state tiles addressed as buffers (wave-uniform descriptor + one vector offset + the field as scalar
# offset) against per-lane column pointers (-DRL_LANE_PTR), one call, three instances; then the parity subset on the shipped library.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r04i.sh'
TAG=r04i
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 --num-envs 4096 --rounds 3 $V/ptr_34.so $V/buf_34.so 2>&1 | grep -v amdgpu.ids | tee $OUT/state_buf_ab.txt
RL_ENV_SUB=1 python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 --num-envs 16384 --rounds 3 $V/ptr_31.so $V/buf_31.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/state_buf_ab.txt
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 --rounds 3 $V/ptr_78.so $V/buf_78.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/state_buf_ab.txt
timeout 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_canary.py tests/test_gpu_self_collision.py tests/test_gpu_teacher_forced.py tests/test_gpu_lane_mapping.py tests/test_gpu_edge_cases.py -m gpu -q > $OUT/pytest_subset.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -6 $OUT/pytest_subset.log
for t in Rough-Unitree-G1 Rough-FFTAI-GR1T1 Rough-Booster-T1 Rough-RobotEra-Xbot; do python tools/sweep_envs.py RobotLab-Isaac-Velocity-$t-v0 2048,4096 2>&1 | grep -v amdgpu.ids | tee -a $OUT/trunk_sweep.txt; done
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
