#!/bin/bash
# Round 3, GPU call: the launch-size selection over the three lane mappings (sweep with the default choice), then the whole GPU tier, smoke and the bench line
OUT=gpurun_out/r03k
mkdir -p $OUT
export TMPDIR=/tmp
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
RL_ENV_DEBUG=1 timeout 400 python tools/sweep_envs.py $A1 1024,4096,5120,8192,10240,12288,16384,20480,24576,32768,65536 2>&1 | grep -v "amdgpu.ids\|lane program" | tee $OUT/sweep_auto.txt
timeout 1500 python -m pytest tests -m gpu -q -rf -s > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_gpu.log | cut -c1-400 | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -3
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 300 $OUT/bench.json; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['roofline']['kernel_ms'], d.get('mid_batch'), d.get('large_batch'))"
