#!/bin/bash
# Round 6: what each gpurun call of the round ran (one section per call; results land in gpurun_out/r06<call>/, the summaries that are
# to be judged are copied into profiles/ by hand).
#   /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/gpu_r06.sh a'
CALL=${1:-a}
TAG=r06$CALL
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
GO2=RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0
GO2W=RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0
G1=RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0
GR1=RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0
prof() {  # name, cmd, rocprofv3 args...
  local name=$1; local cmd=$2; shift; shift
  ( cd /tmp && timeout 300 rocprofv3 "$@" -d $GRAFT_REPO_ROOT/$OUT/prof_$name -- $cmd > $GRAFT_REPO_ROOT/$OUT/under_$name.json 2> $GRAFT_REPO_ROOT/$OUT/$name.err )
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/$name.txt 2>&1 || true
  rm -rf $OUT/prof_$name
}

case $CALL in
a)
  # the critic off the collection loop's critical path (robot_lab_amd/collect.py overlap=True) against the serial loop, one call;
  # the actor / critic alone; then the GPU tier on the tree with the trunk-link-share fix (csrc/env_step.h substep_aba_trunk)
  for cfg in "$A1 4096" "$G1 2048"; do
    set -- $cfg
    for ov in 0 1 0 1; do RL_OVERLAP=$ov timeout 300 python tools/bench_collect.py $1 $2 40 >> $OUT/collect.txt 2>&1; done
  done
  cat $OUT/collect.txt
  timeout 300 python tools/bench_policy.py > $OUT/policy.txt 2>&1; cat $OUT/policy.txt
  timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
  timeout 300 python bench.py --task $G1 --num-envs 2048 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_G1.json 2> $OUT/bench_G1.err; tail -c 600 $OUT/bench_G1.json
  timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
  tail -15 $OUT/pytest_gpu.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  ;;
b)
  # the critic through the small-footprint launch (fits beside the env kernel's workgroup) under env step t; the Flat twins' Specs;
  # then the whole GPU tier
  for cfg in "$A1 4096" "$G1 2048"; do
    set -- $cfg
    for mode in "0 0" "1 1" "1 0" "0 0" "1 1"; do set -- $1 $2 $mode; RL_OVERLAP=$3 RL_CRITIC_SMALL=$4 timeout 300 python tools/bench_collect.py $1 $2 40 >> $OUT/collect.txt 2>&1; done
  done
  grep -v amdgpu.ids $OUT/collect.txt
  timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json
  timeout 300 python bench.py --task RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_A1_Flat.json 2> $OUT/bench_A1_Flat.err; head -c 900 $OUT/bench_A1_Flat.json
  RL_ENV_SPEC=0 timeout 300 python bench.py --task RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_A1_Flat_interpreter.json 2> /dev/null; head -c 900 $OUT/bench_A1_Flat_interpreter.json
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?" >> $OUT/pytest_gpu.log
  tail -15 $OUT/pytest_gpu.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  mv gpurun_out/spec_vs_interpreter.jsonl $OUT/ 2>/dev/null
  ;;
c)
  # scan rays dealt to the lanes by yaw quadrant (csrc/env_terms.h scan_ray_of_slot) against the reference's fixed order: kernel time and
  # FETCH_SIZE, one call; then the host-side cost of enqueueing a step with 8 ranks alive
  for cfg in "$A1 4096" "$GO2 4096"; do
    set -- $cfg
    timeout 400 python tools/ab_bench.py --task $1 --num-envs $2 --rounds 3 --steady $V/scanfixed_34.so $V/scanyaw_34.so >> $OUT/scan_lanes_ab.txt 2>&1
  done
  cat $OUT/scan_lanes_ab.txt
  for v in scanfixed scanyaw; do
    RL_ENV_LIB=$V/${v}_34.so prof ${v}_fetch "python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0" --pmc FETCH_SIZE
    grep -A2 "env_kernel" $OUT/${v}_fetch.txt | head -6
  done
  timeout 600 python tools/host_enqueue.py --ranks 8 --out $OUT/host_enqueue_8ranks.json > $OUT/host_enqueue.log 2>&1; tail -c 1500 $OUT/host_enqueue.log
  timeout 300 python tools/host_enqueue.py --ranks 1 --out $OUT/host_enqueue_1rank.json > $OUT/host_enqueue1.log 2>&1; tail -c 600 $OUT/host_enqueue1.log
  ;;
d)
  # the scan-lane A/B again on the SPECIALISED A1 kernel (call c's variants carried the interpreter only), with FETCH_SIZE of both
  timeout 400 python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 --steady $V/scanfixed_34.so $V/scanyaw_34.so >> $OUT/scan_lanes_ab.txt 2>&1
  grep -v amdgpu $OUT/scan_lanes_ab.txt
  for v in scanfixed scanyaw; do
    RL_ENV_LIB=$GRAFT_REPO_ROOT/$V/${v}_34.so prof ${v}_fetch "python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0" --pmc FETCH_SIZE
    grep -A2 "env_kernel" $OUT/${v}_fetch.txt | head -6
  done
  ;;
e)
  # step kernels specialised at run time: the GPU tests of the plugin path, then every task with and without
  timeout 900 python -m pytest tests/test_gpu_specs.py -m gpu -q -x -k "jit or Flat" > $OUT/pytest_specs_jit.log 2>&1; echo "rc=$?" >> $OUT/pytest_specs_jit.log; tail -5 $OUT/pytest_specs_jit.log
  timeout 2400 python tools/bench_every_task.py --jit > $OUT/all_tasks_jit.txt 2> $OUT/all_tasks_jit.err; cat $OUT/all_tasks_jit.txt | cut -c1-250
  ;;
esac
