#!/bin/bash
# Round 5: what each gpurun call of the round ran (one script, one section per call; results land in gpurun_out/r05<call>/ and the
# summaries that are to be judged are copied into profiles/ by hand).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r05.sh a'
# Calls: a, b the specialised term stack and the round's first evidence; c, d canaries after the explicit-fma draws, the observation stage on a Spec;
# e Tita's quadruped instance; f, g, j what a reset costs (phase clocks of a resetting wavefront); h, i, k the reset path rewritten; l - o the episode
# log (deferred adds, partial rows); p, s, t more batched reads; q the noise-pass interleave (dropped); r the GPU tier; u, v one limb joint per
# sub-lane (kinematics, actuators); z the final evidence; w G1 after v; y the closing tier.  Variant libraries: tools/build_variant.sh.
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
TITA=RobotLab-Isaac-Velocity-Rough-DDTRobot-Tita-v0
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
b)
  # 1. one-call A/Bs on the specialised A1 kernel (steady-state windows): the scan's loads issued before the command update / state write-back,
  #    and the stage ablations (a stage's cost = the difference to the full kernel); the same switch on G1
  timeout 600 python tools/ab_bench.py --steady --rounds 2 $V/ablfull_34.so $V/scanearly_34.so $V/ablnorewards_34.so $V/ablnoobs_34.so $V/ablnoscan_34.so $V/ablsub0_34.so $V/ablskeleton_34.so > $OUT/a1_spec_stage_ablation.txt 2>&1
  cat $OUT/a1_spec_stage_ablation.txt
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $G1 --num-envs 2048 $V/full_78.so $V/scanearly_78.so > $OUT/g1_scan_early_ab.txt 2>&1
  cat $OUT/g1_scan_early_ab.txt
  # 2. in-kernel phase clock of the specialised kernels (a clock build: every stamp drains the memory pipelines - shares, not ticks)
  RL_ENV_LIB=$V/clockspec_34.so timeout 200 python tools/phase_clock.py $A1 4096 > $OUT/phase_clock_a1.txt 2>&1
  RL_ENV_LIB=$V/clockspec_78.so timeout 200 python tools/phase_clock.py $G1 2048 > $OUT/phase_clock_g1.txt 2>&1
  cat $OUT/phase_clock_a1.txt $OUT/phase_clock_g1.txt
  # 3. bench lines with `roofline` for the BASELINE configs 2 - 5 (A1 default line incl. the large-batch legs and the CPU baseline)
  timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_driver_flags.json 2> /dev/null
  for t in $GO2 $GO2W; do timeout 200 python bench.py --task $t --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_$(echo $t | cut -d- -f6).json 2>/dev/null; done
  timeout 200 python bench.py --task $G1 --num-envs 2048 --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_G1.json 2>/dev/null
  python - <<PY | tee $OUT/baseline_configs.txt
import json
for n in ("bench_default", "bench_driver_flags", "bench_Go2", "bench_Go2W", "bench_G1"):
    d = json.load(open("$OUT/%s.json" % n))
    print("%-20s %-62s value %7.2f M env-steps/s  ms_per_step %.4f  kernel_ms %.4f  roofline.frac %.4f  resets in window %d" % (n, d["config"]["workload"].split(",")[0], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["window"]["envs_reset_in_window"]))
d = json.load(open("$OUT/bench_default.json"))
for leg in ("mid_batch", "large_batch"):
    print(leg, {k: d.get(leg, {}).get(k) for k in ("envs_per_gpu", "value", "ms_per_step", "roofline_frac", "envs_per_wavefront")})
print("cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value", "cores", "kind", "per_core")})
PY
  # 4. kernel traces + counter passes on the final tree (separate --pmc passes, no other trace domain)
  A1S="python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0"
  A1FULL="python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline"
  G1S="python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --large-batch-envs 0 --task $G1 --num-envs 2048"
  A8K="python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --large-batch-envs 0 --num-envs 8192"
  A64K="python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --preroll 100 --no-cpu-baseline --large-batch-envs 0 --num-envs 65536"
  SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  WAIT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
  prof a1_kernel_stats "$A1FULL" --kernel-trace --stats
  prof g1_kernel_stats "$G1S" --kernel-trace --stats
  for cfg in "a1:$A1S" "g1:$G1S" "a1_8192:$A8K" "a1_65536:$A64K"; do
    name=${cfg%%:*}; cmd=${cfg#*:}
    prof ${name}_pmc_fetch "$cmd" --pmc FETCH_SIZE
    prof ${name}_pmc_write "$cmd" --pmc WRITE_SIZE
    prof ${name}_pmc_sq "$cmd" --pmc $SQ
    prof ${name}_pmc_wait "$cmd" --pmc $WAIT
  done
  head -9 $OUT/a1_kernel_stats.txt; head -6 $OUT/g1_kernel_stats.txt
  grep "env_kernel" $OUT/*_pmc_fetch.txt $OUT/*_pmc_write.txt | grep mean | cut -c1-40,100-240
  # 5. the collection phase of a PPO iteration as one hipGraph (24 x (actor + critic, act, env) + GAE)
  timeout 300 python tools/bench_collect.py $A1 4096 40 2>/dev/null | tail -2 > $OUT/collect.txt
  timeout 300 python tools/bench_collect.py $G1 2048 20 2>/dev/null | tail -1 >> $OUT/collect.txt
  cat $OUT/collect.txt
  # 6. the tolerance floor again, now with the oracle's own fp32-solve twin in the report (tests/helpers.py `oracle_with_fp32_solves`)
  RL_ENV_LIB=$V/exact_34.so RL_REPORT_TAG=exact_ timeout 300 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "A1-v0-4096-None" > $OUT/pytest_exact_a1.log 2>&1; echo "rc=$?" >> $OUT/pytest_exact_a1.log
  RL_ENV_LIB=$V/exact_78.so RL_REPORT_TAG=exact_ timeout 300 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "G1-v0-2048-None" > $OUT/pytest_exact_g1.log 2>&1; echo "rc=$?" >> $OUT/pytest_exact_g1.log
  RL_ENV_LIB=$V/exact_2078.so RL_REPORT_TAG=exact_ timeout 300 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "GR1T1-v0-1024-sub8" > $OUT/pytest_exact_gr1.log 2>&1; echo "rc=$?" >> $OUT/pytest_exact_gr1.log
  tail -n 2 $OUT/pytest_exact_*.log
  mv gpurun_out/teacher_forced_exact_*.json $OUT/ 2>/dev/null
  # 7. the whole GPU tier on the final tree
  timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -8 $OUT/pytest_gpu.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  mv gpurun_out/spec_vs_interpreter.jsonl gpurun_out/train_distributed_8ranks.json $OUT/ 2>/dev/null
  ;;
c)
  # after the draws' multiply-adds became explicit fmas and the shape canaries learnt to tell contraction round-off from a clobbered register:
  # the canaries (verbose: which shapes are bit-equal), the kernel-vs-kernel test, smoke(), and the bench line (no regression)
  timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_specs.py -m gpu -q -s > $OUT/pytest_canary_specs.log 2>&1; echo "rc=$?" >> $OUT/pytest_canary_specs.log
  grep -E "canary\]|passed|failed|rc=" $OUT/pytest_canary_specs.log | cut -c1-400 | tail -30
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log
  grep -E "smoke|rc=|Error" $OUT/smoke.log | tail -12
  timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> /dev/null
  python -c "
import json
d=json.load(open('$OUT/bench_default.json')); print('bench_default value %.2f M  ms_per_step %.4f  kernel_ms %.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']), d.get('large_batch',{}).get('value'), d.get('mid_batch',{}).get('value'))"
  ;;
d)
  # the observation stage on a Spec (owner-writes from registers) + hardware sin / cos in the reset pose: one-call A/B against the build before
  # them (ablfull_34 / full_78: call b's kernels), then the parity tiers that cover the change
  timeout 300 python tools/ab_bench.py --steady --rounds 2 $V/ablfull_34.so $V/specobs_34.so > $OUT/spec_obs_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $G1 --num-envs 2048 $V/full_78.so $V/specobs_78.so >> $OUT/spec_obs_ab.txt 2>&1
  cat $OUT/spec_obs_ab.txt
  timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_specs.py tests/test_gpu_edge_cases.py -m gpu -q -s > $OUT/pytest_canary_specs.log 2>&1; echo "rc=$?" >> $OUT/pytest_canary_specs.log
  grep -E "canary\]|passed|failed|rc=" $OUT/pytest_canary_specs.log | cut -c1-400 | tail -12
  timeout 900 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "(A1-v0-4096-None or Go2-v0-4096-None or G1-v0-2048-None or Go2W-v0-4096-None or A1-v0-4096-sub2)" > $OUT/pytest_teacher_forced.log 2>&1; echo "rc=$?" >> $OUT/pytest_teacher_forced.log
  tail -3 $OUT/pytest_teacher_forced.log
  timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> /dev/null
  python -c "
import json
d=json.load(open('$OUT/bench_default.json')); print('bench_default value %.2f M  ms_per_step %.4f  kernel_ms %.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']), d.get('large_batch',{}).get('value'), d.get('mid_batch',{}).get('value'))"
  ;;
e)
  # 1. the observation stage on a Spec, on / off (the same tree, -DRL_SPEC_OBS_OFF), quadrupeds: A1 and Go2W (merged instance)
  timeout 300 python tools/ab_bench.py --steady --rounds 2 $V/obsoff_34.so $V/obson_34.so > $OUT/spec_obs_onoff.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $GO2W $V/obsoff_1044.so $V/obson_1044.so >> $OUT/spec_obs_onoff.txt 2>&1
  cat $OUT/spec_obs_onoff.txt
  # 2. DDT Tita on the rot / pad quadruped instance Topo<4,0,3,6,0,1> (until now: the trunk + limbs instance, RL_ENV_ROTPAD=0): parity in
  #    every shape, one step from a shared state at 4096 envs, and the two instances timed in one call
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "Tita" > $OUT/pytest_tita_parity.log 2>&1; echo "rc=$?" >> $OUT/pytest_tita_parity.log
  tail -4 $OUT/pytest_tita_parity.log
  timeout 900 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "Tita" > $OUT/pytest_tita_teacher_forced.log 2>&1; echo "rc=$?" >> $OUT/pytest_tita_teacher_forced.log
  tail -4 $OUT/pytest_tita_teacher_forced.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  for n in 4096 16384; do
    timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $TITA --num-envs $n trunk_limbs:RL_ENV_ROTPAD=0 rot_pad_quadruped:RL_ENV_ROTPAD=1 >> $OUT/tita_instances.txt 2>&1
  done
  cat $OUT/tita_instances.txt
  ;;
f)
  # what the reset path costs a launch (a launch ends with its slowest wavefront, and a wavefront with a resetting env runs reset_env +
  # the observation stage's kinematics of the new pose): steady-state windows with and without termination terms (RL_ENV_TERMS=0: nothing
  # ever resets), interpreter on both sides.  Tita falls under random actions: ~0.29 resets per env-step; G1 ~0.02; A1 time-outs only
  RL_ENV_LIB=$V/clock_4044.so timeout 200 python tools/phase_clock.py $TITA 4096 > $OUT/phase_clock_tita.txt 2>&1
  cat $OUT/phase_clock_tita.txt | tail -26
  for cfg in "$TITA 4096" "$G1 2048" "$A1 4096" "$GO2W 4096"; do
    set -- $cfg
    timeout 300 python tools/ab_bench.py --task $1 --num-envs $2 --rounds 2 --steady resets:RL_ENV_SPEC=0 no_resets:RL_ENV_SPEC=0,RL_ENV_TERMS=0 >> $OUT/reset_cost.txt 2>&1
  done
  cat $OUT/reset_cost.txt
  ;;
g)
  # the reset path phase by phase: env 0 of every wavefront times out on every step (tools/phase_clock.py --reset-env0), against the same
  # kernel without forced resets; and what the interval events (push, command resampling) cost a launch
  for cfg in "clockspec_34 $A1 4096" "clockspec_78 $G1 2048" "clock_4044 $TITA 4096"; do
    set -- $cfg
    RL_ENV_LIB=$V/$1.so RL_ENV_TERMS=0 timeout 200 python tools/phase_clock.py $2 $3 2>&1 | grep -v amdgpu.ids > $OUT/phase_clock_$1_no_resets.txt
    RL_ENV_LIB=$V/$1.so timeout 200 python tools/phase_clock.py $2 $3 --reset-env0 2>&1 | grep -v amdgpu.ids > $OUT/phase_clock_$1_reset_env0.txt
    paste $OUT/phase_clock_$1_no_resets.txt $OUT/phase_clock_$1_reset_env0.txt | cut -c1-60,61-200 | head -40
  done
  for cfg in "$A1 4096" "$G1 2048"; do
    set -- $cfg
    timeout 300 python tools/ab_bench.py --task $1 --num-envs $2 --rounds 2 --steady all:RL_ENV_SPEC=0 no_resets:RL_ENV_SPEC=0,RL_ENV_TERMS=0 no_resets_no_intervals:RL_ENV_SPEC=0,RL_ENV_TERMS=0,RL_ENV_INTERVALS=0 >> $OUT/event_cost.txt 2>&1
  done
  cat $OUT/event_cost.txt
  ;;
h)
  # the reset path as straight-line code + the episode sums' log in the reward write-back + the scanner pose of a reset env from the trunk
  # joints alone: the kernels of the commit before (pre_*) against the tree's, steady-state windows of one call; then the phase table of a
  # resetting wavefront again, and the parity tiers that cover resets
  timeout 300 python tools/ab_bench.py --steady --rounds 3 $V/pre_34.so new:RL_ENV_SPEC=1 > $OUT/reset_path_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 3 --task $G1 --num-envs 2048 $V/pre_78.so new:RL_ENV_SPEC=1 >> $OUT/reset_path_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $TITA --num-envs 4096 new:RL_ENV_SPEC=1 no_resets:RL_ENV_TERMS=0 >> $OUT/reset_path_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $GO2W --num-envs 4096 new:RL_ENV_SPEC=1 no_resets:RL_ENV_TERMS=0 >> $OUT/reset_path_ab.txt 2>&1
  cat $OUT/reset_path_ab.txt
  for cfg in "clockspec_34 $A1 4096" "clockspec_78 $G1 2048"; do
    set -- $cfg
    RL_ENV_LIB=$V/$1.so timeout 200 python tools/phase_clock.py $2 $3 --reset-env0 2>&1 | grep -v amdgpu.ids > $OUT/phase_clock_$1_reset_env0.txt
    grep -E "total|reset|resets|obs.kin|rewards.writeback" $OUT/phase_clock_$1_reset_env0.txt
  done
  timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_specs.py tests/test_gpu_edge_cases.py -m gpu -q > $OUT/pytest_canary_specs.log 2>&1; echo "rc=$?" >> $OUT/pytest_canary_specs.log
  tail -3 $OUT/pytest_canary_specs.log
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "reset or log or G1 or Tita or Flat" > $OUT/pytest_parity_resets.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity_resets.log
  tail -3 $OUT/pytest_parity_resets.log
  timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> /dev/null
  python -c "
import json
d=json.load(open('$OUT/bench_default.json')); print('bench_default value %.2f M  ms_per_step %.4f  kernel_ms %.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms']), d.get('large_batch',{}).get('value'), d.get('mid_batch',{}).get('value'))"
  ;;
i)
  # the same change on the other lane mappings (8192 envs: two sub-lanes per limb; 65 536: one lane per limb, which has no table of uniforms),
  # and whether Loong Flat's envelope outlier (env 22, 1.14 x its six-twin envelope) is the commit before's too
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --num-envs 8192 $V/pre_32.so $V/new_32.so > $OUT/reset_path_ab_other_mappings.txt 2>&1
  timeout 400 python tools/ab_bench.py --steady --rounds 2 --num-envs 65536 $V/pre_31.so $V/new_31.so >> $OUT/reset_path_ab_other_mappings.txt 2>&1
  cat $OUT/reset_path_ab_other_mappings.txt
  RL_ENV_LIB=$V/pre_78.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "Loong and sub8-None" 2>&1 | grep -E "bad|passed|failed" | cut -c1-300 | tail -3
  ;;
j)
  # what a reset still costs the wavefront that carries it, without contention: every 64th wavefront resets its env 0 on every step
  for cfg in "clockspec_34 $A1 4096" "clockspec_78 $G1 2048"; do
    set -- $cfg
    RL_ENV_LIB=$V/$1.so timeout 200 python tools/phase_clock.py $2 $3 --reset-env0=64 2>&1 | grep -v amdgpu.ids > $OUT/phase_clock_$1_reset_every64.txt
    cut -c1-150 $OUT/phase_clock_$1_reset_every64.txt
  done
  ;;
k)
  # the reset as ONE batch of pinned reads (rl_pin): the commit before the reset work (pre_*), steady state, one call; the uncontended phase
  # table of a resetting wavefront; parity of what resets touch
  timeout 300 python tools/ab_bench.py --steady --rounds 3 $V/pre_34.so new:RL_ENV_SPEC=1 > $OUT/reset_batch_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 3 --task $G1 --num-envs 2048 $V/pre_78.so new:RL_ENV_SPEC=1 >> $OUT/reset_batch_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --num-envs 8192 $V/pre_32.so $V/new_32.so >> $OUT/reset_batch_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $G1 --num-envs 2048 new:RL_ENV_SPEC=1 no_resets:RL_ENV_SPEC=1,RL_ENV_TERMS=0 >> $OUT/reset_batch_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 new:RL_ENV_SPEC=1 no_resets:RL_ENV_SPEC=1,RL_ENV_TERMS=0 >> $OUT/reset_batch_ab.txt 2>&1
  cat $OUT/reset_batch_ab.txt
  for cfg in "clockspec_34 $A1 4096" "clockspec_78 $G1 2048"; do
    set -- $cfg
    RL_ENV_LIB=$V/$1.so timeout 200 python tools/phase_clock.py $2 $3 --reset-env0=64 2>&1 | grep -v amdgpu.ids > $OUT/phase_clock_$1_reset_every64.txt
    grep -E "reset|total|writeback|obs.kin|terminations|observations" $OUT/phase_clock_$1_reset_every64.txt | cut -c1-150
  done
  timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_specs.py tests/test_gpu_edge_cases.py -m gpu -q > $OUT/pytest_canary_specs.log 2>&1; echo "rc=$?" >> $OUT/pytest_canary_specs.log
  tail -3 $OUT/pytest_canary_specs.log
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "reset or log or G1 or Tita or Flat" > $OUT/pytest_parity_resets.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity_resets.log
  tail -3 $OUT/pytest_parity_resets.log
  ;;
l)
  # the episode log's atomic adds deferred to the end of the step: A1 / G1 against the commit before the reset work, Tita against itself
  # without resets; then the tiers that read the log
  timeout 300 python tools/ab_bench.py --steady --rounds 3 $V/pre_34.so new:RL_ENV_SPEC=1 no_resets:RL_ENV_SPEC=1,RL_ENV_TERMS=0 > $OUT/deferred_log_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 3 --task $G1 --num-envs 2048 $V/pre_78.so new:RL_ENV_SPEC=1 no_resets:RL_ENV_SPEC=1,RL_ENV_TERMS=0 >> $OUT/deferred_log_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $TITA --num-envs 4096 new:RL_ENV_SPEC=1 no_resets:RL_ENV_TERMS=0 >> $OUT/deferred_log_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $GO2 --num-envs 4096 new:RL_ENV_SPEC=1 no_resets:RL_ENV_TERMS=0 >> $OUT/deferred_log_ab.txt 2>&1
  cat $OUT/deferred_log_ab.txt
  timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_specs.py tests/test_gpu_edge_cases.py tests/test_gpu_dropin.py -m gpu -q > $OUT/pytest_canary_specs.log 2>&1; echo "rc=$?" >> $OUT/pytest_canary_specs.log
  tail -3 $OUT/pytest_canary_specs.log
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "reset or log or G1-v0 or Tita or A1-v0" > $OUT/pytest_parity_resets.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity_resets.log
  tail -3 $OUT/pytest_parity_resets.log
  ;;
m)
  # Tita under random actions (0.29 resets per env-step): the launch with the log's atomic adds, without them (-DRL_ABL_NO_LOG), without resets
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $TITA --num-envs 4096 $V/full_4044.so $V/nolog_4044.so no_resets:RL_ENV_TERMS=0@$V/full_4044.so > $OUT/tita_log_atomics.txt 2>&1
  cat $OUT/tita_log_atomics.txt
  ;;
n)
  # the log ring's slots as 32 partial rows (wavefront w adds into row w % 32, readers sum): Tita under random actions, and the headline kernels
  # against the commit before the reset work; then every tier that reads a log
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $TITA --num-envs 4096 new:RL_ENV_SPEC=1 no_resets:RL_ENV_TERMS=0 > $OUT/log_rows_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 3 $V/pre_34.so new:RL_ENV_SPEC=1 no_resets:RL_ENV_SPEC=1,RL_ENV_TERMS=0 >> $OUT/log_rows_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 3 --task $G1 --num-envs 2048 $V/pre_78.so new:RL_ENV_SPEC=1 no_resets:RL_ENV_SPEC=1,RL_ENV_TERMS=0 >> $OUT/log_rows_ab.txt 2>&1
  cat $OUT/log_rows_ab.txt
  timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_specs.py tests/test_gpu_edge_cases.py tests/test_gpu_dropin.py tests/test_gpu_multirank.py tests/test_gpu_collect.py -m gpu -q > $OUT/pytest_log_readers.log 2>&1; echo "rc=$?" >> $OUT/pytest_log_readers.log
  tail -3 $OUT/pytest_log_readers.log
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "reset or log or Tita" > $OUT/pytest_parity_resets.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity_resets.log
  tail -3 $OUT/pytest_parity_resets.log
  ;;
o)
  # the log upkeep's reads in one batch with the state loads
  timeout 300 python tools/ab_bench.py --steady --rounds 3 $V/pre_34.so new:RL_ENV_SPEC=1 no_resets:RL_ENV_SPEC=1,RL_ENV_TERMS=0 > $OUT/log_upkeep_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 3 --task $G1 --num-envs 2048 $V/pre_78.so new:RL_ENV_SPEC=1 no_resets:RL_ENV_SPEC=1,RL_ENV_TERMS=0 >> $OUT/log_upkeep_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $TITA --num-envs 4096 new:RL_ENV_SPEC=1 no_resets:RL_ENV_TERMS=0 >> $OUT/log_upkeep_ab.txt 2>&1
  cat $OUT/log_upkeep_ab.txt
  timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_dropin.py tests/test_gpu_canary.py -m gpu -q > $OUT/pytest_log_readers.log 2>&1; echo "rc=$?" >> $OUT/pytest_log_readers.log
  tail -3 $OUT/pytest_log_readers.log
  ;;
p)
  # the action loads in one batch with the state's, the illegal-contact test over the owned slots: against the commit before
  timeout 300 python tools/ab_bench.py --steady --rounds 3 --task $G1 --num-envs 2048 $V/prev_78.so new:RL_ENV_SPEC=1 > $OUT/action_batch_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 3 $V/prev_34.so new:RL_ENV_SPEC=1 >> $OUT/action_batch_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $GR1 --num-envs 1024 prev:RL_ENV_SPEC=1@$V/prev_2078.so new:RL_ENV_SPEC=1 >> $OUT/action_batch_ab.txt 2>&1
  cat $OUT/action_batch_ab.txt
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "G1-v0 or GR1 or Xbot" > $OUT/pytest_parity.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity.log
  tail -3 $OUT/pytest_parity.log
  ;;
q)
  # the noise pass: a lane's Philox blocks with their rounds side by side.  prev_* = HEAD (before the action batch, the owned-slot illegal
  # test and this)
  timeout 300 python tools/ab_bench.py --steady --rounds 3 $V/prev_34.so new:RL_ENV_SPEC=1 > $OUT/noise_interleave_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 3 --task $G1 --num-envs 2048 $V/prev_78.so new:RL_ENV_SPEC=1 >> $OUT/noise_interleave_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $GO2W --num-envs 4096 prev:RL_ENV_SPEC=1@$V/prev_1044.so new:RL_ENV_SPEC=1 >> $OUT/noise_interleave_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --num-envs 65536 prev:RL_ENV_SPEC=1@$V/new_31.so new:RL_ENV_SPEC=1 >> $OUT/noise_interleave_ab.txt 2>&1
  cat $OUT/noise_interleave_ab.txt
  timeout 600 python -m pytest tests/test_gpu_canary.py tests/test_gpu_specs.py -m gpu -q > $OUT/pytest_canary_specs.log 2>&1; echo "rc=$?" >> $OUT/pytest_canary_specs.log
  tail -3 $OUT/pytest_canary_specs.log
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "A1-v0 or Go2W or G1-v0" > $OUT/pytest_parity.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity.log
  tail -3 $OUT/pytest_parity.log
  ;;
r)
  # the whole GPU tier + smoke() on the tree after the reset / log work
  timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -8 $OUT/pytest_gpu.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  mv gpurun_out/spec_vs_interpreter.jsonl gpurun_out/train_distributed_8ranks.json $OUT/ 2>/dev/null
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log
  grep -E "smoke|rc=|Error" $OUT/smoke.log | tail -12
  ;;
s)
  # the actuator constants of all joints in one batch of LDS reads
  timeout 300 python tools/ab_bench.py --steady --rounds 3 --task $G1 --num-envs 2048 $V/prev2_78.so new:RL_ENV_SPEC=1 > $OUT/actuator_batch_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $GR1 --num-envs 1024 prev:RL_ENV_SPEC=1@$V/prev2_2078.so new:RL_ENV_SPEC=1 >> $OUT/actuator_batch_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 prev:RL_ENV_SPEC=1@$V/prev_34.so new:RL_ENV_SPEC=1 >> $OUT/actuator_batch_ab.txt 2>&1
  cat $OUT/actuator_batch_ab.txt
  ;;
t)
  # two more batches inside the substep (A/B switches): the recursion's armature / limit words, the contact slots' friction rows
  timeout 400 python tools/ab_bench.py --steady --rounds 3 $V/base_34.so $V/jcbatch_34.so $V/fricpre_34.so $V/both_34.so > $OUT/substep_batches_ab.txt 2>&1
  cat $OUT/substep_batches_ab.txt
  ;;
u)
  # kinematics with one limb joint per sub-lane (prefix product / prefix sums over a limb's eight lanes) against the dealt chain of the commit before
  timeout 300 python tools/ab_bench.py --steady --rounds 3 --task $G1 --num-envs 2048 $V/prev3_78.so new:RL_ENV_SPEC=1 > $OUT/kin_scan_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $GR1 --num-envs 1024 prev:RL_ENV_SPEC=1@$V/prev3_2078.so new:RL_ENV_SPEC=1 >> $OUT/kin_scan_ab.txt 2>&1
  cat $OUT/kin_scan_ab.txt
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sub8" > $OUT/pytest_parity_sub8.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity_sub8.log
  tail -3 $OUT/pytest_parity_sub8.log
  timeout 900 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "G1-v0-2048-None or Xbot or GR1T1-v0-1024-sub8" > $OUT/pytest_teacher_forced_trunk.log 2>&1; echo "rc=$?" >> $OUT/pytest_teacher_forced_trunk.log
  tail -3 $OUT/pytest_teacher_forced_trunk.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  timeout 300 python -m pytest tests/test_gpu_canary.py tests/test_gpu_self_collision.py -m gpu -q > $OUT/pytest_canary.log 2>&1; echo "rc=$?" >> $OUT/pytest_canary.log
  tail -2 $OUT/pytest_canary.log
  ;;
z)
  # FINAL TREE: the whole GPU tier, smoke(), the bench lines of the BASELINE configs, kernel traces + counter passes, phase clocks, the collection loop
  timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -4 $OUT/pytest_gpu.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  mv gpurun_out/spec_vs_interpreter.jsonl gpurun_out/train_distributed_8ranks.json $OUT/ 2>/dev/null
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log
  grep -E "smoke|rc=|Error" $OUT/smoke.log | tail -8
  timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_driver_flags.json 2> /dev/null
  for t in $GO2 $GO2W $TITA; do timeout 200 python bench.py --task $t --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_$(echo $t | cut -d- -f6).json 2>/dev/null; done
  timeout 200 python bench.py --task $G1 --num-envs 2048 --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_G1.json 2>/dev/null
  python - <<PY | tee $OUT/baseline_configs.txt
import json
for n in ("bench_default", "bench_driver_flags", "bench_Go2", "bench_Go2W", "bench_G1", "bench_Tita"):
    try:
        d = json.load(open("$OUT/%s.json" % n))
    except Exception as ex:
        print(n, "unreadable", ex); continue
    print("%-20s %-62s value %7.2f M env-steps/s  ms_per_step %.4f  kernel_ms %.4f  roofline.frac %.4f  resets in window %s" % (n, d["config"]["workload"].split(",")[0], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("window", {}).get("envs_reset_in_window")))
d = json.load(open("$OUT/bench_default.json"))
for leg in ("mid_batch", "large_batch"):
    print(leg, {k: d.get(leg, {}).get(k) for k in ("envs_per_gpu", "value", "ms_per_step", "roofline_frac", "envs_per_wavefront")})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind", "per_core")})
PY
  A1S="python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0"
  A1FULL="python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 50 --no-cpu-baseline"
  G1S="python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --large-batch-envs 0 --task $G1 --num-envs 2048"
  SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  WAIT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
  prof a1_kernel_stats "$A1FULL" --kernel-trace --stats
  prof g1_kernel_stats "$G1S" --kernel-trace --stats
  for cfg in "a1:$A1S" "g1:$G1S"; do
    name=${cfg%%:*}; cmd=${cfg#*:}
    prof ${name}_pmc_fetch "$cmd" --pmc FETCH_SIZE
    prof ${name}_pmc_write "$cmd" --pmc WRITE_SIZE
    prof ${name}_pmc_sq "$cmd" --pmc $SQ
    prof ${name}_pmc_wait "$cmd" --pmc $WAIT
  done
  head -9 $OUT/a1_kernel_stats.txt; head -6 $OUT/g1_kernel_stats.txt
  RL_ENV_LIB=$V/clockspec_34.so timeout 200 python tools/phase_clock.py $A1 4096 2>&1 | grep -v amdgpu.ids > $OUT/phase_clock_a1.txt
  RL_ENV_LIB=$V/clockspec_78.so timeout 200 python tools/phase_clock.py $G1 2048 2>&1 | grep -v amdgpu.ids > $OUT/phase_clock_g1.txt
  head -30 $OUT/phase_clock_g1.txt
  timeout 300 python tools/bench_collect.py $A1 4096 40 2>/dev/null | tail -2 > $OUT/collect.txt
  timeout 300 python tools/bench_collect.py $G1 2048 20 2>/dev/null | tail -1 >> $OUT/collect.txt
  cat $OUT/collect.txt
  ;;
v)
  # eight sub-lanes per limb: a limb joint's actuator and joint-local terms by the sub-lane that owns the joint (+ limb broadcasts), against call z's tree
  timeout 300 python tools/ab_bench.py --steady --rounds 3 --task $G1 --num-envs 2048 $V/prev4_78.so new:RL_ENV_SPEC=1 > $OUT/actuators_owned_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $GR1 --num-envs 1024 prev:RL_ENV_SPEC=1@$V/prev4_2078.so new:RL_ENV_SPEC=1 >> $OUT/actuators_owned_ab.txt 2>&1
  cat $OUT/actuators_owned_ab.txt
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sub8" > $OUT/pytest_parity_sub8.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity_sub8.log
  tail -3 $OUT/pytest_parity_sub8.log
  timeout 900 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "G1-v0-2048-None or Xbot or GR1T1-v0-1024-sub8" > $OUT/pytest_teacher_forced_trunk.log 2>&1; echo "rc=$?" >> $OUT/pytest_teacher_forced_trunk.log
  tail -3 $OUT/pytest_teacher_forced_trunk.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  timeout 300 python -m pytest tests/test_gpu_canary.py tests/test_gpu_self_collision.py tests/test_gpu_specs.py -m gpu -q > $OUT/pytest_canary.log 2>&1; echo "rc=$?" >> $OUT/pytest_canary.log
  tail -2 $OUT/pytest_canary.log
  ;;
w)
  # G1 / the trunk + limbs instances on the final tree (after call z: the owner-computes actuators): bench line, kernel trace, counter passes, phase clock,
  # collection loop, and the other humanoids' step times
  timeout 200 python bench.py --task $G1 --num-envs 2048 --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_G1.json 2>/dev/null
  python -c "
import json
d=json.load(open('$OUT/bench_G1.json')); print('bench_G1 value %.2f M  ms_per_step %.4f  kernel_ms %.4f  roofline.frac %.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))" | tee $OUT/bench_G1.txt
  G1S="python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --large-batch-envs 0 --task $G1 --num-envs 2048"
  SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  WAIT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
  prof g1_kernel_stats "$G1S" --kernel-trace --stats
  prof g1_pmc_fetch "$G1S" --pmc FETCH_SIZE
  prof g1_pmc_write "$G1S" --pmc WRITE_SIZE
  prof g1_pmc_sq "$G1S" --pmc $SQ
  prof g1_pmc_wait "$G1S" --pmc $WAIT
  head -6 $OUT/g1_kernel_stats.txt
  RL_ENV_LIB=$V/clockspec_78.so timeout 200 python tools/phase_clock.py $G1 2048 2>&1 | grep -v amdgpu.ids > $OUT/phase_clock_g1.txt
  head -24 $OUT/phase_clock_g1.txt
  timeout 300 python tools/bench_collect.py $G1 2048 20 2>/dev/null | tail -1 > $OUT/collect_g1.txt
  cat $OUT/collect_g1.txt
  for t in RobotLab-Isaac-Velocity-Rough-RobotEra-Xbot-v0 RobotLab-Isaac-Velocity-Rough-Booster-T1-v0 $GR1 RobotLab-Isaac-Velocity-Rough-RoboParty-ATOM01-v0; do
    timeout 200 python tools/ab_bench.py --steady --rounds 1 --task $t --num-envs 2048 final:RL_ENV_SPEC=1 >> $OUT/trunk_sweep.txt 2>&1
  done
  cat $OUT/trunk_sweep.txt
  ;;
y)
  # the whole GPU tier + smoke() + the default bench line on the LAST tree of the round (after call v's change to the trunk + limbs kernels)
  timeout 1700 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -4 $OUT/pytest_gpu.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  mv gpurun_out/spec_vs_interpreter.jsonl gpurun_out/train_distributed_8ranks.json $OUT/ 2>/dev/null
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log
  grep -E "smoke|rc=|Error" $OUT/smoke.log | tail -8
  timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  python -c "
import json
d=json.load(open('$OUT/bench_default.json')); print('bench_default value %.2f M  ms_per_step %.4f  kernel_ms %.4f  roofline.frac %.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), d.get('large_batch',{}).get('value'), d.get('mid_batch',{}).get('value'), d['cpu_baseline']['value'])"
  ;;
x)
  # last tree: step time vs envs per GPU (A1, Go2W, G1; cold windows), every compiled task id, and a short PPO run (the end-to-end check)
  timeout 200 python tools/sweep_envs.py $A1 > $OUT/sweep_envs_a1.txt 2>/dev/null
  timeout 200 python tools/sweep_envs.py $GO2W 4096,8192,16384,65536 > $OUT/sweep_envs_go2w.txt 2>/dev/null
  timeout 200 python tools/sweep_envs.py $G1 512,1024,2048,4096,8192 > $OUT/sweep_envs_g1.txt 2>/dev/null
  cat $OUT/sweep_envs_a1.txt $OUT/sweep_envs_go2w.txt $OUT/sweep_envs_g1.txt
  timeout 120 python tools/train_demo.py --iterations 300 2>/dev/null | tail -4 > $OUT/train_demo_a1_flat_300.txt
  cat $OUT/train_demo_a1_flat_300.txt
  timeout 300 python tools/bench_every_task.py > $OUT/all_tasks.txt 2>/dev/null
  tail -48 $OUT/all_tasks.txt | cut -c1-120
  ;;
aa)
  # quadrupeds, 16 lanes per env: a limb joint's actuator and joint-local terms by the sub-lane that owns the joint, against the same tree with
  # -DRL_ACT_REPLICATED; then the quadruped parity tiers
  timeout 300 python tools/ab_bench.py --steady --rounds 3 $V/repl_34.so owned:RL_ENV_SPEC=1 > $OUT/quad_act_owned_ab.txt 2>&1
  timeout 300 python tools/ab_bench.py --steady --rounds 2 --task $GO2W repl:RL_ENV_SPEC=1@$V/repl_1044.so owned:RL_ENV_SPEC=1 >> $OUT/quad_act_owned_ab.txt 2>&1
  cat $OUT/quad_act_owned_ab.txt
  timeout 300 python -m pytest tests/test_gpu_canary.py tests/test_gpu_specs.py tests/test_gpu_lane_mapping.py -m gpu -q > $OUT/pytest_canary_specs.log 2>&1; echo "rc=$?" >> $OUT/pytest_canary_specs.log
  tail -2 $OUT/pytest_canary_specs.log
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "not sub8 and not sub4" > $OUT/pytest_parity_quadrupeds.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity_quadrupeds.log
  tail -3 $OUT/pytest_parity_quadrupeds.log
  timeout 600 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "A1-v0-4096-None or Go2-v0-4096-None or Go2W-v0-4096-None or B2W or M20 or Tita-v0-4096-None" > $OUT/pytest_teacher_forced_quadrupeds.log 2>&1; echo "rc=$?" >> $OUT/pytest_teacher_forced_quadrupeds.log
  tail -3 $OUT/pytest_teacher_forced_quadrupeds.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  ;;
ab)
  # the bench lines after call aa (quadruped actuators by the owning sub-lane) + the tiers that consume the env's outputs end to end
  timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  for t in $GO2 $GO2W; do timeout 100 python bench.py --task $t --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_$(echo $t | cut -d- -f6).json 2>/dev/null; done
  python - <<PY | tee $OUT/baseline_configs.txt
import json
for n in ("bench_default", "bench_Go2", "bench_Go2W"):
    d = json.load(open("$OUT/%s.json" % n))
    print("%-16s %-50s value %7.2f M env-steps/s  ms_per_step %.4f  kernel_ms %.4f  roofline.frac %.4f" % (n, d["config"]["workload"].split(",")[0], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
d = json.load(open("$OUT/bench_default.json"))
for leg in ("mid_batch", "large_batch"):
    print(leg, {k: d.get(leg, {}).get(k) for k in ("envs_per_gpu", "value", "ms_per_step", "roofline_frac")})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind")})
PY
  timeout 200 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_collect.py tests/test_gpu_edge_cases.py tests/test_gpu_episode_stats.py -m gpu -q -x > $OUT/pytest_consumers.log 2>&1; echo "rc=$?" >> $OUT/pytest_consumers.log
  tail -2 $OUT/pytest_consumers.log
  timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log
  grep -E "smoke|rc=" $OUT/smoke.log | tail -6
  ;;
ac)
  # the rest of the GPU tier on the last tree (what calls aa / ab did not run after the quadruped change)
  timeout 330 python -m pytest tests/test_gpu_all_tasks.py tests/test_gpu_command_levels.py tests/test_gpu_train.py tests/test_gpu_multirank.py tests/test_gpu_distributed_train.py tests/test_gpu_self_collision.py -m gpu -q > $OUT/pytest_rest.log 2>&1; echo "rc=$?" >> $OUT/pytest_rest.log
  tail -3 $OUT/pytest_rest.log
  ;;
ad)
  # last tree: the teacher-forced configs of the other quadruped lane mappings (their kernels share aba_solve with the changed 16-lane one) and the
  # trunk + limbs parity shapes
  timeout 200 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -k "sub1 or sub2" > $OUT/pytest_teacher_forced_sub12.log 2>&1; echo "rc=$?" >> $OUT/pytest_teacher_forced_sub12.log
  tail -2 $OUT/pytest_teacher_forced_sub12.log
  timeout 90 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sub8 or sub4" > $OUT/pytest_parity_trunk.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity_trunk.log
  tail -2 $OUT/pytest_parity_trunk.log
  ;;
esac
