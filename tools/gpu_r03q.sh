#!/bin/bash
OUT=gpurun_out/r03q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_all_tasks.py -m gpu -q -rf > $OUT/pytest_all_tasks.log 2>&1; grep -E "passed|failed|FAILED" $OUT/pytest_all_tasks.log | cut -c1-250 | tail -12
timeout 900 python tools/bench_every_task.py 2>&1 | grep -v amdgpu.ids | tee $OUT/all_tasks.txt | tail -42
python tools/ab_bench.py --num-envs 4096 --rounds 1 robot_lab_amd/csrc/variants/cur_34.so robot_lab_amd/csrc/librl_env_hip.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_clamp.txt
