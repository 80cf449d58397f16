#!/bin/bash
# Round 3, fourth GPU call (first of the second session): what the third one was meant to measure - the 32-row split-bf16 MLP
# kernel, the collection loop, the env kernel at HEAD against the round-2 source, the distributional parity test, the bench line -
# plus the one-lane-per-limb mapping (RL_ENV_SUB=1) at large env counts and the SQ counters of the G1 kernel.
#   /usr/local/graft/bin/gpurun --timeout 1300 -- 'bash tools/gpu_r03d.sh'
OUT=gpurun_out/r03d
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
G1=RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0
T0=$SECONDS
lap() { echo "[lap] $1 at $((SECONDS - T0)) s" | tee -a $OUT/laps.txt; }
timeout 500 python -m pytest tests/test_policy.py tests/test_gpu_collect.py -m gpu -q -rf > $OUT/pytest_mlp.log 2>&1; echo "pytest mlp rc=$?" >> $OUT/pytest_mlp.log
tail -4 $OUT/pytest_mlp.log; lap pytest_mlp
for rt in 1 2; do
  echo "== RL_MLP_SPLIT_RT=$rt" | tee -a $OUT/policy.txt
  RL_MLP_SPLIT_RT=$rt timeout 200 python tools/bench_pair.py 4096 2>&1 | grep -v amdgpu.ids | tee -a $OUT/policy.txt
  echo "== RL_MLP_SPLIT_RT=$rt" | tee -a $OUT/collect.txt
  RL_MLP_SPLIT_RT=$rt timeout 200 python tools/bench_collect.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/collect.txt
done
timeout 200 python tools/bench_collect.py $G1 2048 2>&1 | grep -v amdgpu.ids | tee -a $OUT/collect.txt
lap mlp_collect
python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 2 $V/r02_34.so $V/head_34.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_a1.txt
python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 1 $V/r02_74.so $V/head_74.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_g1.txt
lap ab
for sub in 4 1; do
  echo "== RL_ENV_SUB=$sub" | tee -a $OUT/sweep_sub.txt
  RL_ENV_SUB=$sub timeout 300 python tools/sweep_envs.py $A1 4096,8192,16384,32768,65536 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sweep_sub.txt
done
lap sweep
timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_teacher_forced.py tests/test_gpu_episode_stats.py "tests/test_gpu_parity.py::test_short_horizon_parity" -m gpu -q -rf -s -k "canary or episode or ((A1 or G1 or Go2) and not Flat) or HandStand" > $OUT/pytest_env.log 2>&1; echo "pytest env rc=$?" >> $OUT/pytest_env.log
grep -E "episode-stats|passed|failed|FAILED|rc=" $OUT/pytest_env.log | cut -c1-700 | tail -12
cp gpurun_out/episode_stats_*.json $OUT/ 2>/dev/null
lap pytest_env
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1800 $OUT/bench.json
lap bench
prof() {  # name, task, envs, rocprofv3 args...
  local name=$1 task=$2 envs=$3; shift 3
  ( cd /tmp && timeout 300 rocprofv3 "$@" -d $GRAFT_REPO_ROOT/$OUT/prof_$name -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 30 --no-cpu-baseline --task $task --num-envs $envs > $GRAFT_REPO_ROOT/$OUT/under_$name.json 2> $GRAFT_REPO_ROOT/$OUT/$name.err )
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/$name.txt 2>&1 || true
  rm -rf $OUT/prof_$name
}
prof g1_pmc_sq $G1 2048 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
grep -A10 "PMC counters" $OUT/g1_pmc_sq.txt | grep env_kernel | head -6
prof g1_pmc_wait $G1 2048 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
grep -A10 "PMC counters" $OUT/g1_pmc_wait.txt | grep env_kernel | head -6
lap pmc
