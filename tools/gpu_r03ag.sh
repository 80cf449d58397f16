#!/bin/bash
# Round 3, last call: the build with the six-joint-spine instance (host tables sized for 6 trunk joints, 33 links / 32 DoF) on the suites that
# exercise every instance and the C-ABI surface; bench line
OUT=gpurun_out/r03ag
mkdir -p $OUT
export TMPDIR=/tmp
timeout 150 python -m pytest tests/test_gpu_canary.py tests/test_gpu_dropin.py tests/test_gpu_edge_cases.py tests/test_gpu_collect.py tests/test_gpu_teacher_forced.py -m gpu -q -rf -x -k "not (Go2 or M20 or B2W or Xbot or sub1 or sub2)" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|rc=|^E " $OUT/pytest.log | cut -c1-300 | tail -8
timeout 60 python bench.py --steps 300 --warmup 50 > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print('value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
