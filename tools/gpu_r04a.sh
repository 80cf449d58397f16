#!/bin/bash
# Round 4, first call: GPU test tier (incl. the two-rank train.py body), the default bench line (pre-rolled window, new cpu_baseline),
# kernel trace of the same command, and the G1 counters on the CURRENT build (profiles/traffic.json carried round-2 numbers for G1).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_r04a.sh'
TAG=r04a
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host_cores.txt; lscpu | grep -i "model name\|thread\|core(s)\|socket" >> $OUT/host_cores.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench_driver_flags.json
prof() {  # name, cmd, rocprofv3 args...
  local name=$1; local cmd=$2; shift; shift
  ( cd /tmp && timeout 600 rocprofv3 "$@" -d $GRAFT_REPO_ROOT/$OUT/prof_$name -- $cmd > $GRAFT_REPO_ROOT/$OUT/under_$name.json 2> $GRAFT_REPO_ROOT/$OUT/$name.err )
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/$name.txt 2>&1 || true
  rm -rf $OUT/prof_$name
}
A1="python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline"
G1="python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --large-batch-envs 0 --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048"
prof a1_kernel_stats "$A1" --kernel-trace --stats
prof g1_kernel_stats "$G1" --kernel-trace --stats
prof g1_pmc_fetch "$G1" --pmc FETCH_SIZE
prof g1_pmc_write "$G1" --pmc WRITE_SIZE
prof g1_pmc_sq "$G1" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
prof g1_pmc_wait "$G1" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
head -8 $OUT/a1_kernel_stats.txt; head -8 $OUT/g1_kernel_stats.txt; grep -A3 env_kernel $OUT/g1_pmc_fetch.txt | head; grep -A3 env_kernel $OUT/g1_pmc_write.txt | head
