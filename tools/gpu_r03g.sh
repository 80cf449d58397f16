#!/bin/bash
# Round 3, seventh GPU call: the one-lane-per-limb mapping after its LDS diet (variable-length body-table rows, critic row straight to
# HBM, four-wavefront workgroups) and the launch-size selection of the mapping; then the WHOLE -m gpu tier on the new build.
OUT=gpurun_out/r03g
mkdir -p $OUT
export TMPDIR=/tmp
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
T0=$SECONDS
lap() { echo "[lap] $1 at $((SECONDS - T0)) s" | tee -a $OUT/laps.txt; }
for sub in auto 1 4; do
  echo "== RL_ENV_SUB=$sub" | tee -a $OUT/sweep.txt
  if [ $sub = auto ]; then RL_ENV_DEBUG=1 timeout 400 python tools/sweep_envs.py $A1 4096,8192,12288,16384,24576,32768,65536 2>&1 | grep -v "amdgpu.ids\|lane program" | tee -a $OUT/sweep.txt
  else RL_ENV_SUB=$sub timeout 400 python tools/sweep_envs.py $A1 4096,12288,16384,32768,65536 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep.txt; fi
done
lap sweep
timeout 1500 python -m pytest tests -m gpu -q -rf -s > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/pytest_gpu.log
grep -E "episode-stats|passed|failed|FAILED|rc=" $OUT/pytest_gpu.log | cut -c1-400 | tail -30
cp gpurun_out/episode_stats_*.json gpurun_out/teacher_forced_*.json $OUT/ 2>/dev/null
lap pytest
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 600 $OUT/bench.json
lap bench
