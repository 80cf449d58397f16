#!/bin/bash
OUT=gpurun_out/r03u
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_train.py tests/test_gpu_collect.py -m gpu -q -rf > $OUT/pytest.log 2>&1; grep -E "passed|failed|FAILED|Error" $OUT/pytest.log | cut -c1-300 | tail; grep -n "^E " $OUT/pytest.log | head -20
