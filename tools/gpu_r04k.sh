#!/bin/bash
# Round 4, closing call on the final tree: the whole GPU tier, the default and the driver-flag bench lines, G1 / trunk-robot timings,
# kernel traces (A1 incl. the mid- and large-batch legs, G1) and the counter passes behind profiles/traffic.json (A1 and G1: FETCH_SIZE,
# WRITE_SIZE, SQ instruction mix, SQ wait shares; separate --pmc passes, no other trace domain).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_r04k.sh'
TAG=r04k
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
G1T=RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0
G1ARGS="--no-cpu-baseline --large-batch-envs 0 --task $G1T --num-envs 2048"
S=$(date +%s.%N); python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; E=$(date +%s.%N)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_driver_flags.json 2> /dev/null
python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_1000.json 2> /dev/null
python bench.py --steps 300 --warmup 50 $G1ARGS > $OUT/g1_bench.json 2> /dev/null
python - <<PY | tee $OUT/summary.txt
import json
print("default bench.py wall time %.1f s" % ($E - $S))
for n in ("bench_default", "bench_driver_flags", "bench_1000", "g1_bench"):
    d = json.load(open("$OUT/%s.json" % n))
    print(n, "value %.2f M" % (d["value"] / 1e6), "ms_per_step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d["window"]["envs_reset_in_window"], d["window"]["mean_bodies_in_contact_at_end"])
d = json.load(open("$OUT/bench_default.json"))
print({k: d["cpu_baseline"][k] for k in ("value", "cores", "per_core", "repeats")})
for leg in ("large_batch", "mid_batch"):
    print(leg, {k: d.get(leg, {}).get(k) for k in ("envs_per_gpu", "value", "ms_per_step", "roofline_frac")})
PY
for t in Rough-Unitree-G1 Rough-FFTAI-GR1T1 Rough-Booster-T1 Rough-RobotEra-Xbot; do python tools/sweep_envs.py RobotLab-Isaac-Velocity-$t-v0 2048,4096 2>&1 | grep -v amdgpu.ids | tee -a $OUT/trunk_sweep.txt; done
prof() {  # name, cmd, rocprofv3 args...
  local name=$1; local cmd=$2; shift; shift
  ( cd /tmp && timeout 600 rocprofv3 "$@" -d $GRAFT_REPO_ROOT/$OUT/prof_$name -- $cmd > $GRAFT_REPO_ROOT/$OUT/under_$name.json 2> $GRAFT_REPO_ROOT/$OUT/$name.err )
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/$name.txt 2>&1 || true
  rm -rf $OUT/prof_$name
}
A1="python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline"
A1S="python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0"
G1="python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 $G1ARGS"
prof a1_kernel_stats "$A1" --kernel-trace --stats
prof g1_kernel_stats "$G1" --kernel-trace --stats
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
WAIT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
prof a1_pmc_fetch "$A1S" --pmc FETCH_SIZE
prof a1_pmc_write "$A1S" --pmc WRITE_SIZE
prof a1_pmc_sq "$A1S" --pmc $SQ
prof a1_pmc_wait "$A1S" --pmc $WAIT
prof g1_pmc_fetch "$G1" --pmc FETCH_SIZE
prof g1_pmc_write "$G1" --pmc WRITE_SIZE
prof g1_pmc_sq "$G1" --pmc $SQ
prof g1_pmc_wait "$G1" --pmc $WAIT
head -9 $OUT/a1_kernel_stats.txt; head -6 $OUT/g1_kernel_stats.txt
grep "env_kernel" $OUT/a1_pmc_*.txt $OUT/g1_pmc_*.txt | grep mean | cut -c1-30,90-220
