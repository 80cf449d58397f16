#!/bin/bash
# Round 3, second GPU call: split-bf16 MLP (correctness + timing), the env kernel after the reward schedule / terrain-base / med3
# changes (parity subset + A/B against the round-2 source in ONE call), scheduler-strategy builds, stage ablations, G1 SQ counters.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r03b.sh'
OUT=gpurun_out/r03b
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
G1=RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0
timeout 600 python -m pytest tests/test_policy.py tests/test_gpu_collect.py tests/test_rollout.py -m gpu -q -s -rf > $OUT/pytest_mlp.log 2>&1; echo "pytest mlp rc=$?" >> $OUT/pytest_mlp.log
grep -E "^\[mlp|passed|failed|FAILED|rc=" $OUT/pytest_mlp.log | tail -40
timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_teacher_forced.py "tests/test_gpu_parity.py::test_short_horizon_parity" -m gpu -q -rf -k "canary or (A1 or G1 or Go2W or Tita or HandStand) or teacher" > $OUT/pytest_env.log 2>&1; echo "pytest env rc=$?" >> $OUT/pytest_env.log
tail -12 $OUT/pytest_env.log
python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 2 $V/r02_34.so $V/new_34.so $V/ilp_34.so $V/bias0_34.so $V/noterms_34.so $V/norew_34.so $V/noobs_34.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_a1.txt
python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 2 $V/r02_74.so $V/new_74.so $V/ilp_74.so $V/bias0_74.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_g1.txt
for p in "" f32; do
  echo "== RL_MLP_PRECISION=$p" | tee -a $OUT/policy.txt
  RL_MLP_PRECISION=$p timeout 300 python tools/bench_pair.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/policy.txt
  RL_MLP_PRECISION=$p timeout 300 python tools/bench_collect.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/collect.txt
done
tools/micro/halfwave > $OUT/halfwave.txt 2>&1; cat $OUT/halfwave.txt
prof() {  # name, lib, task, envs, rocprofv3 args...
  local name=$1 lib=$2 task=$3 envs=$4; shift 4
  ( cd /tmp && RL_ENV_LIB=$GRAFT_REPO_ROOT/$lib timeout 400 rocprofv3 "$@" -d $GRAFT_REPO_ROOT/$OUT/prof_$name -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 30 --no-cpu-baseline --task $task --num-envs $envs > $GRAFT_REPO_ROOT/$OUT/under_$name.json 2> $GRAFT_REPO_ROOT/$OUT/$name.err )
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/$name.txt 2>&1 || true
  rm -rf $OUT/prof_$name
}
prof g1_pmc_sq $V/new_74.so $G1 2048 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
prof g1_pmc_wait $V/new_74.so $G1 2048 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
grep -A10 "PMC counters" $OUT/g1_pmc_sq.txt | grep env_kernel | head -10
grep -A10 "PMC counters" $OUT/g1_pmc_wait.txt | grep env_kernel | head -10
