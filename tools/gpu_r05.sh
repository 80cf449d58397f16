#!/bin/bash
# Round 5: what each gpurun call of the round ran (one script, one section per call; results land in gpurun_out/r05<call>/ and the
# summaries that are to be judged are copied into profiles/ by hand).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r05.sh a'
CALL=${1:-a}
TAG=r05$CALL
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
  # 1. the step kernel specialised on the task against the term-stack interpreter of the same library, steady-state windows, one call
  for cfg in "$A1 4096" "$GO2 4096" "$GO2W 4096" "$G1 2048" "$A1 8192" "$A1 65536"; do
    set -- $cfg
    timeout 300 python tools/ab_bench.py --task $1 --num-envs $2 --rounds 2 --steady interpreter:RL_ENV_SPEC=0 specialised:RL_ENV_SPEC=1 >> $OUT/spec_ab.txt 2>&1
  done
  cat $OUT/spec_ab.txt
  # 2. the specialised kernels under the parity tiers: kernel-vs-kernel, canaries, teacher-forced at the BASELINE sizes
  timeout 900 python -m pytest tests/test_gpu_specs.py tests/test_gpu_canary.py -m gpu -q -x > $OUT/pytest_specs_canary.log 2>&1; echo "rc=$?" >> $OUT/pytest_specs_canary.log
  tail -4 $OUT/pytest_specs_canary.log
  timeout 900 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -x -k "(A1-v0-4096-None or Go2-v0-4096-None or G1-v0-2048-None or Go2W-v0-4096-None or A1-v0-4096-sub1 or A1-v0-4096-sub2)" > $OUT/pytest_teacher_forced.log 2>&1; echo "rc=$?" >> $OUT/pytest_teacher_forced.log
  tail -4 $OUT/pytest_teacher_forced.log
  mkdir -p $OUT/tf_spec && mv gpurun_out/teacher_forced_*.json $OUT/tf_spec/ 2>/dev/null
  # 3. the tolerance floor: the same test on the exact-math build of the library (IEEE divide / sqrt, libm sin / cos / exp on the device)
  RL_ENV_LIB=$V/exact_34.so RL_REPORT_TAG=exact_ timeout 300 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "A1-v0-4096-None" > $OUT/pytest_exact_a1.log 2>&1; echo "rc=$?" >> $OUT/pytest_exact_a1.log
  RL_ENV_LIB=$V/exact_78.so RL_REPORT_TAG=exact_ timeout 300 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "G1-v0-2048-None" > $OUT/pytest_exact_g1.log 2>&1; echo "rc=$?" >> $OUT/pytest_exact_g1.log
  RL_ENV_LIB=$V/exact_2078.so RL_REPORT_TAG=exact_ timeout 300 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "GR1T1-v0-1024-sub8" > $OUT/pytest_exact_gr1.log 2>&1; echo "rc=$?" >> $OUT/pytest_exact_gr1.log
  tail -2 $OUT/pytest_exact_*.log
  mv gpurun_out/teacher_forced_exact_*.json $OUT/ 2>/dev/null
  # 4. eight ranks on the one GPU: bench.py and the reference's train.py --distributed body
  RL_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 8 --num-envs 512 --steps 50 --warmup 10 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_share_gpu_8ranks.json 2> $OUT/bench_share_gpu_8ranks.err
  tail -c 600 $OUT/bench_share_gpu_8ranks.json
  timeout 600 python -m pytest tests/test_gpu_distributed_train.py -m gpu -q > $OUT/pytest_distributed_train.log 2>&1; echo "rc=$?" >> $OUT/pytest_distributed_train.log
  tail -3 $OUT/pytest_distributed_train.log
  mv gpurun_out/train_distributed_8ranks.json $OUT/ 2>/dev/null
  # 5. the default bench line + how long the host side of one env.step() is (64 envs: the kernel is ~10 us, the loop is host-bound)
  timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err
  timeout 120 python bench.py --num-envs 64 --steps 2000 --warmup 100 --preroll 0 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_host_bound_64envs.json 2>/dev/null
  python -c "
import json
for n in ('bench_default','bench_host_bound_64envs'):
    d=json.load(open('$OUT/%s.json'%n)); print(n, 'value %.2f M  ms_per_step %.4f  kernel_ms %.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']), d.get('large_batch',{}).get('value'), d.get('mid_batch',{}).get('value'))"
  ;;
esac
