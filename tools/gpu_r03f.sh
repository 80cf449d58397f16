#!/bin/bash
# Round 3, sixth GPU call: the env library built with -mllvm -amdgpu-remove-redundant-endcf=false (the root cause of the reload-under-a-
# stale-EXEC miscompile, tools/isa_exec_hazard.py) against the default build: speed (one-call A/B on three instances) and the GPU
# canaries + parity of the instances the static check flags in the default build (Go2W four-wavefront shape).
OUT=gpurun_out/r03f
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
GO2W=RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0
G1=RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0
python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 2 $V/base_34.so $V/endcf_34.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
python tools/ab_bench.py --task $GO2W --num-envs 4096 --rounds 2 $V/base_1044.so $V/endcf_1044.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 2 $V/base_74.so $V/endcf_74.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
for v in base endcf; do
  echo "== $v (Go2W, merged instance)" | tee -a $OUT/variants.txt
  RL_ENV_LIB=$PWD/$V/${v}_1044.so timeout 600 python -m pytest tests/test_gpu_canary.py tests/test_gpu_teacher_forced.py "tests/test_gpu_parity.py::test_short_horizon_parity" -m gpu -q -rf -k "Go2W and Rough and not 0-" > $OUT/pytest_go2w_$v.log 2>&1
  grep -E "passed|failed|FAILED" $OUT/pytest_go2w_$v.log | cut -c1-300 | tee -a $OUT/variants.txt
done
echo "== endcf (A1 / Go2 / HandStand)" | tee -a $OUT/variants.txt
RL_ENV_LIB=$PWD/$V/endcf_34.so timeout 600 python -m pytest tests/test_gpu_canary.py tests/test_gpu_teacher_forced.py "tests/test_gpu_parity.py::test_short_horizon_parity" -m gpu -q -rf -k "(A1 or Go2-v0 or HandStand) and not None--1" > $OUT/pytest_a1_endcf.log 2>&1
grep -E "passed|failed|FAILED" $OUT/pytest_a1_endcf.log | cut -c1-300 | tee -a $OUT/variants.txt
echo "== endcf (G1)" | tee -a $OUT/variants.txt
RL_ENV_LIB=$PWD/$V/endcf_74.so timeout 600 python -m pytest tests/test_gpu_canary.py tests/test_gpu_teacher_forced.py tests/test_gpu_episode_stats.py "tests/test_gpu_parity.py::test_short_horizon_parity" -m gpu -q -rf -s -k "G1 and Rough" > $OUT/pytest_g1_endcf.log 2>&1
grep -E "episode-stats|passed|failed|FAILED" $OUT/pytest_g1_endcf.log | cut -c1-600 | tee -a $OUT/variants.txt
cp gpurun_out/episode_stats_*.json $OUT/ 2>/dev/null
