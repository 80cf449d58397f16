#!/bin/bash
# Round 4, call q: rocprofv3 kernel traces of the bench command on the FINAL tree (A1 incl. the mid- and large-batch legs, G1).
#   /usr/local/graft/bin/gpurun --timeout 400 -- 'bash tools/gpu_r04q.sh'
TAG=r04q
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
prof() {  # name, cmd, rocprofv3 args...
  local name=$1; local cmd=$2; shift; shift
  ( cd /tmp && timeout 300 rocprofv3 "$@" -d $GRAFT_REPO_ROOT/$OUT/prof_$name -- $cmd > $GRAFT_REPO_ROOT/$OUT/under_$name.json 2> $GRAFT_REPO_ROOT/$OUT/$name.err )
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/$name.txt 2>&1 || true
  rm -rf $OUT/prof_$name
}
A1="python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline"
G1="python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --large-batch-envs 0 --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048"
prof a1_kernel_stats "$A1" --kernel-trace --stats
prof g1_kernel_stats "$G1" --kernel-trace --stats
head -9 $OUT/a1_kernel_stats.txt; head -6 $OUT/g1_kernel_stats.txt
python -c "
import json
for n in ('a1','g1'):
    d=json.load(open('$OUT/under_%s_kernel_stats.json'%n)); print(n, 'under rocprofv3: value %.2f M kernel_ms %.4f'%(d['value']/1e6, d['roofline']['kernel_ms']))"
