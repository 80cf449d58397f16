#!/bin/bash
# Second evidence call of a round: the other BASELINE configs, the G1 profile, the collection loop's kernel table, the env-count
# sweep and the all-task table.   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_evidence.sh r02'
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_evidence
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/bench_all.sh 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_all.txt
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 2> /dev/null | tail -1 > $OUT/g1_bench.json
prof() {  # name, cmd..., rocprofv3 args after --
  local name=$1; local cmd=$2; shift; shift
  ( cd /tmp && timeout 600 rocprofv3 "$@" -d $GRAFT_REPO_ROOT/$OUT/prof_$name -- $cmd > $GRAFT_REPO_ROOT/$OUT/under_$name.log 2> $GRAFT_REPO_ROOT/$OUT/$name.err )
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/$name.txt 2>&1 || true
  rm -rf $OUT/prof_$name
}
G1="python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048"
prof g1_kernel_stats "$G1" --kernel-trace --stats
prof g1_pmc_fetch "$G1" --pmc FETCH_SIZE
prof g1_pmc_write "$G1" --pmc WRITE_SIZE
RL_GRAPH=0 prof collect_kernel_stats "python $GRAFT_REPO_ROOT/tools/bench_collect.py" --kernel-trace --stats
python tools/bench_collect.py 2>&1 | grep -v amdgpu.ids | tee $OUT/collect.txt
python tools/bench_collect.py RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0 2>&1 | grep -v amdgpu.ids | tee -a $OUT/collect.txt
python tools/bench_collect.py RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 2048 2>&1 | grep -v amdgpu.ids | tee -a $OUT/collect.txt
python tools/bench_pair.py 2>&1 | grep -v amdgpu.ids | tee $OUT/policy.txt
python tools/bench_policy.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/policy.txt
python tools/sweep_envs.py 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep_a1.txt
python tools/sweep_envs.py RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 2>&1 | grep -v amdgpu.ids | head -8 | tee $OUT/sweep_g1.txt
python tools/bench_every_task.py 2>&1 | grep -v amdgpu.ids > $OUT/all_tasks.txt
tail -5 $OUT/all_tasks.txt
head -8 $OUT/g1_kernel_stats.txt; head -10 $OUT/collect_kernel_stats.txt
