#!/bin/bash
OUT=gpurun_out/r03l
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_lane_mapping.py -m gpu -q -rf > $OUT/pytest_mapping.log 2>&1; tail -3 $OUT/pytest_mapping.log
timeout 900 python -m pytest "tests/test_gpu_parity.py::test_short_horizon_parity" -m gpu -q -rf -s > $OUT/pytest_parity.log 2>&1; grep -E "parity-small|passed|failed" $OUT/pytest_parity.log | cut -c1-200
