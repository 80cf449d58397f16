#!/bin/bash
# Round 3, evidence of the final build: whole GPU tier, bench line, rocprofv3 kernel trace + SQ / HBM counters of the bench command
# (A1 Rough 4096 + its large-batch leg), G1 kernel trace, collection loop, lane-mapping sweep, all BASELINE configs.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_r03final.sh'
TAG=${1:-r03final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_round.sh $TAG tests
G1=RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_g1 -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 --task $G1 --num-envs 2048 > $GRAFT_REPO_ROOT/$OUT/g1_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/g1.err )
python tools/rocpd_summary.py $(find $OUT/prof_g1 -name "*.db" | head -1) > $OUT/g1_kernel_stats.txt 2>&1; rm -rf $OUT/prof_g1
head -6 $OUT/g1_kernel_stats.txt
for t in Go2 Go2W; do timeout 200 python bench.py --steps 500 --warmup 100 --no-cpu-baseline --large-batch-envs 0 --task RobotLab-Isaac-Velocity-Rough-Unitree-$t-v0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['metric'], '%.1f M env-steps/s' % (d['value']/1e6), 'kernel %.1f us' % (d['roofline']['kernel_ms']*1e3))" | tee -a $OUT/baseline_configs.txt; done
timeout 200 python bench.py --steps 500 --warmup 100 --no-cpu-baseline --large-batch-envs 0 --task $G1 --num-envs 2048 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['metric'], '%.1f M env-steps/s' % (d['value']/1e6), 'kernel %.1f us' % (d['roofline']['kernel_ms']*1e3))" | tee -a $OUT/baseline_configs.txt
timeout 200 python tools/bench_collect.py 2>&1 | grep -v amdgpu.ids | tee $OUT/collect.txt
timeout 200 python tools/bench_pair.py 4096 2>&1 | grep -v amdgpu.ids | tee $OUT/policy.txt
timeout 300 python tools/sweep_envs.py RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.txt
RL_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_share_gpu_2ranks.json; head -c 400 $OUT/bench_share_gpu_2ranks.json
