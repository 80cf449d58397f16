#!/bin/bash
# Round 4, sixth call: the tree with the register contact stash (32-lane mapping), one 64-bit product per Philox multiplier, Booster T1's
# mask fix: trunk-robot parity subset + T1, G1 / GR1 / T1 timings, the default bench line timed end to end, A1 kernel trace.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_r04f.sh'
TAG=r04f
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_canary.py tests/test_gpu_self_collision.py tests/test_gpu_teacher_forced.py tests/test_gpu_episode_stats.py tests/test_gpu_edge_cases.py tests/test_gpu_lane_mapping.py -m gpu -q > $OUT/pytest_subset.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -6 $OUT/pytest_subset.log
for t in Rough-Unitree-G1 Rough-FFTAI-GR1T1 Rough-Booster-T1 Rough-RobotEra-Xbot; do python tools/sweep_envs.py RobotLab-Isaac-Velocity-$t-v0 2048,4096 2>&1 | grep -v amdgpu.ids | tee -a $OUT/trunk_sweep.txt; done
G1ARGS="--no-cpu-baseline --large-batch-envs 0 --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048"
python bench.py --steps 300 --warmup 50 $G1ARGS > $OUT/g1_bench.json 2> /dev/null
S=$(date +%s.%N); python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; E=$(date +%s.%N)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_driver_flags.json 2> /dev/null
python - <<PY | tee $OUT/summary.txt
import json
print("default bench.py wall time %.1f s" % ($E - $S))
for n in ("bench_default", "bench_driver_flags", "g1_bench"):
    d = json.load(open("$OUT/%s.json" % n))
    print(n, "value %.2f M" % (d["value"] / 1e6), "ms_per_step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d["window"]["envs_reset_in_window"], d["window"]["mean_bodies_in_contact_at_end"])
d = json.load(open("$OUT/bench_default.json"))
print({k: d["cpu_baseline"][k] for k in ("value", "cores", "per_core", "repeats")}, d.get("large_batch", {}).get("value"), d.get("mid_batch", {}).get("value"))
PY
prof() {  # name, cmd, rocprofv3 args...
  local name=$1; local cmd=$2; shift; shift
  ( cd /tmp && timeout 600 rocprofv3 "$@" -d $GRAFT_REPO_ROOT/$OUT/prof_$name -- $cmd > $GRAFT_REPO_ROOT/$OUT/under_$name.json 2> $GRAFT_REPO_ROOT/$OUT/$name.err )
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/$name.txt 2>&1 || true
  rm -rf $OUT/prof_$name
}
A1="python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline"
G1="python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 $G1ARGS"
prof a1_kernel_stats "$A1" --kernel-trace --stats
prof g1_kernel_stats "$G1" --kernel-trace --stats
head -8 $OUT/a1_kernel_stats.txt; head -6 $OUT/g1_kernel_stats.txt
