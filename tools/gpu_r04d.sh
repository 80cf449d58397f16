#!/bin/bash
# Round 4, fourth call: the trunk + limbs instances on limb-major LDS records (16-byte vector access) with the link-velocity pass fused
# into the kinematics - parity tier for the trunk robots, then the one-call A/B of the three builds of the G1 32-lane kernel, the G1
# bench line and its SQ counters.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_r04d.sh'
TAG=r04d
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_canary.py tests/test_gpu_self_collision.py tests/test_gpu_teacher_forced.py tests/test_gpu_all_tasks.py tests/test_gpu_episode_stats.py tests/test_gpu_edge_cases.py tests/test_gpu_lane_mapping.py -m gpu -q -k "G1 or GR1 or ATOM01 or Xbot or Gen1 or Loong or Z1 or self_contact or stays_finite or statistics or mapping" > $OUT/pytest_trunk.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_trunk.log
tail -6 $OUT/pytest_trunk.log
V=robot_lab_amd/csrc/variants
G1=RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0
python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 2 $V/base_78.so $V/limbmajor_78.so $V/fusedvel_78.so 2>&1 | grep -v amdgpu.ids | tee $OUT/g1_ab.txt
python tools/ab_bench.py --task $G1 --num-envs 512 --rounds 1 $V/base_78.so $V/fusedvel_78.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/g1_ab.txt
G1ARGS="--no-cpu-baseline --large-batch-envs 0 --task $G1 --num-envs 2048"
python bench.py --steps 300 --warmup 50 $G1ARGS > $OUT/g1_bench.json 2> /dev/null
python -c "
import json; d=json.load(open('$OUT/g1_bench.json')); print('g1_bench value %.2f M kernel_ms %.4f frac %.4f' % (d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['frac']), d['window'])" | tee $OUT/summary.txt
prof() {  # name, cmd, rocprofv3 args...
  local name=$1; local cmd=$2; shift; shift
  ( cd /tmp && timeout 600 rocprofv3 "$@" -d $GRAFT_REPO_ROOT/$OUT/prof_$name -- $cmd > $GRAFT_REPO_ROOT/$OUT/under_$name.json 2> $GRAFT_REPO_ROOT/$OUT/$name.err )
  local DB=$(find $OUT/prof_$name -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/$name.txt 2>&1 || true
  rm -rf $OUT/prof_$name
}
G1C="python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 $G1ARGS"
prof g1_kernel_stats "$G1C" --kernel-trace --stats
prof g1_pmc_sq "$G1C" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
prof g1_pmc_wait "$G1C" --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
prof g1_pmc_lds "$G1C" --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES
prof g1_pmc_fetch "$G1C" --pmc FETCH_SIZE
prof g1_pmc_write "$G1C" --pmc WRITE_SIZE
head -6 $OUT/g1_kernel_stats.txt; grep "env_kernel.*0, 8" $OUT/g1_pmc_*.txt | grep mean | cut -c60-200
