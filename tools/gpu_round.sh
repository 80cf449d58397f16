#!/bin/bash
# One gpurun call of a development round: GPU test tier, the default bench line, a rocprofv3 kernel trace of the same
# command, SQ instruction counters.  Everything lands under gpurun_out/$TAG/ (copied into profiles/ by hand when it is to be judged).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_round.sh r02a [tests|notests]'
TAG=${1:-r02}
MODE=${2:-tests}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host_cores.txt
if [ "$MODE" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -s > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -8 $OUT/pytest_gpu.log
fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -c 900 $OUT/bench.json
prof() {  # name, rocprofv3 args...
  local name=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 "$@" -d $GRAFT_REPO_ROOT/$OUT/prof_$name -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/bench_under_$name.json 2> $GRAFT_REPO_ROOT/$OUT/$name.err )
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/$name.txt 2>&1 || true
  rm -rf $OUT/prof_$name  # the database itself is large; the summary is what is kept
}
prof kernel_stats --kernel-trace --stats
prof pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
prof pmc_wait --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
prof pmc_fetch --pmc FETCH_SIZE
prof pmc_write --pmc WRITE_SIZE
head -12 $OUT/kernel_stats.txt
grep -A3 "env_kernel" $OUT/pmc_sq.txt | head -12
