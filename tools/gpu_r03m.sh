#!/bin/bash
# Round 3, GPU call: PPO on the env end to end (collect = one hipGraph launch of the HIP kernels, update = torch, parameters pushed in place)
OUT=gpurun_out/r03m
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_policy.py -m gpu -q -rf -s > $OUT/pytest_train.log 2>&1; grep -E "\[train\]|passed|failed|FAILED|Error" $OUT/pytest_train.log | cut -c1-300 | tail -8
timeout 900 python tools/train_demo.py --iterations 300 --out $OUT 2>&1 | grep -v amdgpu.ids | tee $OUT/train_a1_flat.txt | tail -40
