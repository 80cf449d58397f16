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
esac
