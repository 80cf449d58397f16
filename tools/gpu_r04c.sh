#!/bin/bash
# Round 4, third call: the whole GPU tier on the build with (i) the 32-lane mapping as the trunk + limbs default and (ii) the reset
# uniforms computed once per env by its lanes together; then the headline and G1 bench lines in steady state and their kernel traces.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_r04c.sh'
TAG=r04c
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_1000.json 2> /dev/null
G1ARGS="--no-cpu-baseline --large-batch-envs 0 --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048"
python bench.py --steps 300 --warmup 50 $G1ARGS > $OUT/g1_bench.json 2> /dev/null
RL_ENV_SUB=4 python bench.py --steps 300 --warmup 50 $G1ARGS > $OUT/g1_bench_sub4.json 2> /dev/null
python bench.py --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 4096 > $OUT/g1_bench_4096.json 2> /dev/null
python - <<PY | tee $OUT/summary.txt
import json
for n in ("bench_driver_flags", "bench_1000", "g1_bench", "g1_bench_sub4", "g1_bench_4096"):
    d = json.load(open("$OUT/%s.json" % n))
    print(n, "value %.2f M" % (d["value"] / 1e6), "ms_per_step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d["window"]["envs_reset_in_window"], d["window"]["mean_bodies_in_contact_at_end"])
d = json.load(open("$OUT/bench_driver_flags.json"))
print({k: d["cpu_baseline"][k] for k in ("value", "cores", "per_core", "repeats", "cgroup_cpu_quota")})
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
