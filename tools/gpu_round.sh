#!/bin/bash
# One gpurun call of a development round: GPU test tier, the default bench line, a rocprofv3 kernel trace of the same
# command.  Everything lands under gpurun_out/$TAG/ (copied into profiles/ by hand when it is to be judged).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a [tests|notests]'
TAG=${1:-r02}
MODE=${2:-tests}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host_cores.txt
if [ "$MODE" = "tests" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/rocprof.err )
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB > $OUT/kernel_stats.txt 2>&1 || true
rm -rf $OUT/prof  # the database itself is large; the summary is what is kept
head -20 $OUT/kernel_stats.txt
