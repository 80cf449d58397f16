#!/bin/bash
# Round 3, closing GPU call: whole GPU tier on the final tree, the edge-case / drop-in suites with each lane mapping forced, smoke, bench line
OUT=gpurun_out/r03ae
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rf > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_gpu.log | cut -c1-300 | tail -12
for sub in 1 2; do
  RL_ENV_SUB=$sub timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_dropin.py tests/test_gpu_collect.py -m gpu -q -rf > $OUT/pytest_sub$sub.log 2>&1
  echo "RL_ENV_SUB=$sub: $(grep -E 'passed|failed' $OUT/pytest_sub$sub.log | tail -1)"; grep FAILED $OUT/pytest_sub$sub.log | cut -c1-200
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -2
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; python -c "
import json; d=json.load(open('$OUT/bench.json')); print('value', d['value'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac']); print(d.get('mid_batch')); print(d.get('large_batch')); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
