#!/bin/bash
# Round 3, fifth GPU call: which build of the 3-joint instance is correct in the four-wavefront shape (the descriptor pinning of the
# last commit broke it): variants x {cross-shape bit-equality canary, timer canary, short-horizon parity at RL_ENV_WG=-4}.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_r03e.sh'
OUT=gpurun_out/r03e
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
for v in nopin pin pinlr nopinlr; do
  echo "== $v" | tee -a $OUT/variants.txt
  RL_ENV_LIB=$PWD/$V/${v}_34.so timeout 400 python -m pytest tests/test_gpu_canary.py "tests/test_gpu_parity.py::test_short_horizon_parity" -m gpu -q -rf -k "(A1 or Go2-v0) and Rough" > $OUT/pytest_$v.log 2>&1
  grep -E "passed|failed|FAILED" $OUT/pytest_$v.log | cut -c1-300 | tee -a $OUT/variants.txt
done
python tools/ab_bench.py --num-envs 4096 --rounds 2 $V/nopin_34.so $V/nopinlr_34.so $V/pin_34.so $V/pinlr_34.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_a1.txt
# the command-range curricula on the GPU: split step (head launch, decision, tail launch)
timeout 300 python -m pytest tests/test_gpu_command_levels.py tests/test_gpu_edge_cases.py -m gpu -q -rf > $OUT/pytest_levels.log 2>&1; tail -3 $OUT/pytest_levels.log
