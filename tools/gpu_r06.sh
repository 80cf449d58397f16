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
f)
  # the reward kinds added to the specialised evaluation (hand-stand terms, base_height_l2, wheel_vel_penalty, feet_distance_*): plugin kernels vs interpreter
  timeout 900 python -m pytest tests/test_gpu_specs.py -m gpu -q -k "jit" > $OUT/pytest_specs_jit.log 2>&1; echo "rc=$?" >> $OUT/pytest_specs_jit.log; tail -25 $OUT/pytest_specs_jit.log
  timeout 600 python tools/bench_every_task.py --jit 2>/dev/null | grep -i "tita\|handstand" | cut -c1-250
  ;;
g)
  # soak: humanoids that are never reset (RL_ENV_TERMS=0: they fall and lie on the ground for thousands of steps - the trunk-link contacts of
  # csrc/env_step.h substep_aba_trunk under load), and the default A1 / G1 runs; finiteness and envelopes
  for t in $G1 RobotLab-Isaac-Velocity-Rough-Booster-T1-v0 $GR1; do
    RL_ENV_TERMS=0 timeout 400 python tools/soak.py $t 2000 random 2048 2>&1 | grep -v amdgpu | tail -5 >> $OUT/soak_no_terminations.txt
    RL_ENV_TERMS=0 timeout 400 python tools/soak.py $t 1000 zero 2048 2>&1 | grep -v amdgpu | tail -3 >> $OUT/soak_no_terminations.txt
  done
  cat $OUT/soak_no_terminations.txt | cut -c1-260
  timeout 400 python tools/soak.py $A1 3000 random 4096 2>&1 | grep -v amdgpu | tail -3 > $OUT/soak.txt
  timeout 400 python tools/soak.py $G1 3000 random 2048 2>&1 | grep -v amdgpu | tail -3 >> $OUT/soak.txt
  cat $OUT/soak.txt | cut -c1-260
  ;;
h)
  # G1 Velocity-Flat under the reference's cfg with the stand-in learner, after the trunk-link-share fix: does it still settle into the crouch of
  # rounds 3 - 5 (open finding)?  600 iterations, the posture of the batch every 50
  timeout 900 python tools/train_demo.py --task RobotLab-Isaac-Velocity-Flat-Unitree-G1-v0 --iterations 600 --print-every 50 2>/dev/null | tail -16 > $OUT/train_g1_flat_600.txt
  cat $OUT/train_g1_flat_600.txt | cut -c1-260
  ;;
i)
  # stress of the run-time specialisation: the oracle parity tiers with RL_ENV_JIT=1, i.e. EVERY task of the free-run / teacher-forced / all-tasks
  # tests on a step kernel specialised at create (24 ids x kernel shapes against the fp64 oracle, not only against the interpreter)
  export RL_ENV_JIT=1 RL_ENV_JIT_CACHE=/tmp/jit_cache_tier
  timeout 3000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_teacher_forced.py tests/test_gpu_all_tasks.py tests/test_gpu_edge_cases.py tests/test_gpu_episode_stats.py -m gpu -q > $OUT/pytest_jit_tier.log 2>&1; echo "rc=$?" >> $OUT/pytest_jit_tier.log
  tail -12 $OUT/pytest_jit_tier.log
  ls /tmp/jit_cache_tier | wc -l
  grep -c "compiling a step kernel" $OUT/pytest_jit_tier.log
  ;;
zz2)
  # after the run-time specialisation became the default: the whole GPU tier again (the tiers pin it off: same run time), smoke(), the default
  # bench line, and a task without a built-in Spec through bench.py with no flag at all
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -4 $OUT/pytest_gpu.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log
  grep -E "smoke|rc=|Error|jit" $OUT/smoke.log | tail -8
  timeout 400 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> $OUT/bench_default.err; grep jit $OUT/bench_default.err
  timeout 400 python bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-B2-v0 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_B2.json 2> $OUT/bench_B2.err; grep jit $OUT/bench_B2.err
  RL_ENV_JIT=0 timeout 400 python bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-B2-v0 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_B2_nojit.json 2> /dev/null
  python -c "
import json
for n in ('bench_default','bench_B2','bench_B2_nojit'):
    d=json.load(open('$OUT/%s.json' % n)); print(n, 'value %.2f M  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']), d['config'].get('step_kernel'))"
  ;;
j)
  # the captured collection loop around a run-time specialised env
  timeout 600 python -m pytest tests/test_gpu_collect.py -m gpu -q > $OUT/pytest_collect.log 2>&1; echo "rc=$?" >> $OUT/pytest_collect.log; tail -5 $OUT/pytest_collect.log
  ;;
k)
  # axis-aligned joint rotations composed in their sparse form by the Specs (env_spec.h spec_axis_kind): the specialised A1 / Go2W kernels
  # of the commit before against the tree's, one call; then the spec / canary / teacher-forced tiers of the quadrupeds
  timeout 400 python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 --steady $V/old_34.so $V/new_34.so > $OUT/axis_kind_ab.txt 2>&1
  timeout 400 python tools/ab_bench.py --task $GO2W --num-envs 4096 --rounds 3 --steady $V/old_1044.so $V/new_1044.so >> $OUT/axis_kind_ab.txt 2>&1
  grep -v amdgpu $OUT/axis_kind_ab.txt
  timeout 900 python -m pytest tests/test_gpu_specs.py tests/test_gpu_canary.py -m gpu -q -x > $OUT/pytest_specs_canary.log 2>&1; echo "rc=$?" >> $OUT/pytest_specs_canary.log; tail -4 $OUT/pytest_specs_canary.log
  timeout 900 python -m pytest tests/test_gpu_teacher_forced.py -m gpu -q -x -k "A1 or Go2 or Go2W or M20 or B2W" > $OUT/pytest_tf.log 2>&1; echo "rc=$?" >> $OUT/pytest_tf.log; tail -4 $OUT/pytest_tf.log
  timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_default.json 2> /dev/null; python -c "
import json
d=json.load(open('$OUT/bench_default.json')); print('bench_default value %.2f M  kernel_ms %.4f frac %.4f' % (d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['frac']), d['config'].get('step_kernel'))"
  ;;
l)
  timeout 900 python -m pytest tests/test_gpu_specs.py -m gpu -q > $OUT/pytest_specs.log 2>&1; echo "rc=$?" >> $OUT/pytest_specs.log; tail -5 $OUT/pytest_specs.log
  mv gpurun_out/spec_vs_interpreter.jsonl $OUT/ 2>/dev/null
  ;;
m)
  # packed fp32 arithmetic written by hand in the joint elimination (-DRL_PK: env_step.h eliminate_pk) and -fno-signed-zeros (x + 0 of a zeroed
  # link record folds away), each and both against the tree, one call: the specialised A1 and G1 kernels
  timeout 600 python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 --steady $V/base_34.so $V/nsz_34.so $V/pk_34.so $V/pknsz_34.so > $OUT/a1_pk_ab.txt 2>&1
  timeout 600 python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 3 --steady $V/base_78.so $V/nsz_78.so $V/pk_78.so $V/pknsz_78.so > $OUT/g1_pk_ab.txt 2>&1
  grep -v amdgpu $OUT/a1_pk_ab.txt $OUT/g1_pk_ab.txt
  ;;
n)
  # the tree's defaults (packed elimination + contact blocks + the pair-aligned LDS record of the trunk + limbs instances, -fno-signed-zeros)
  # against the flags and sources of call z2 (base_*: -DRL_NO_PK -fsigned-zeros) and against call m's pknsz_* (elimination only)
  timeout 600 python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 --steady $V/base_34.so $V/pknsz_34.so $V/pk2_34.so > $OUT/a1_pk2_ab.txt 2>&1
  timeout 600 python tools/ab_bench.py --task $GO2W --num-envs 4096 --rounds 3 --steady $V/base_1044.so $V/pk2_1044.so > $OUT/go2w_pk2_ab.txt 2>&1
  timeout 600 python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 3 --steady $V/base_78.so $V/pknsz_78.so $V/pk2_78.so > $OUT/g1_pk2_ab.txt 2>&1
  grep -v amdgpu $OUT/a1_pk2_ab.txt $OUT/go2w_pk2_ab.txt $OUT/g1_pk2_ab.txt
  ;;
o)
  # the packed contact blocks alone: pk3 = without them (-DRL_NO_PK_CONTACT on G1; the quadrupeds' default) against pk2 = with them
  timeout 600 python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 3 --steady $V/pk3_78.so $V/pk2_78.so > $OUT/g1_pk_contact_ab.txt 2>&1
  timeout 600 python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 --steady $V/pk3_34.so $V/pk2_34.so > $OUT/a1_pk_contact_ab.txt 2>&1
  grep -v amdgpu $OUT/g1_pk_contact_ab.txt $OUT/a1_pk_contact_ab.txt
  ;;
p)
  # -ffinite-math-only on top of the tree's flags (x * 0 and the products with the identity frame at the root of every kinematic chain fold,
  # fmin / fmax without the canonicalising copy): ~150 of a substep's 2.3 k vector instructions on A1
  timeout 600 python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 --steady $V/pk3_34.so $V/fin_34.so > $OUT/a1_finite_ab.txt 2>&1
  timeout 600 python tools/ab_bench.py --task $GO2W --num-envs 4096 --rounds 3 --steady $V/pk2_1044.so $V/fin_1044.so > $OUT/go2w_finite_ab.txt 2>&1
  timeout 600 python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 3 --steady $V/pk2_78.so $V/fin_78.so > $OUT/g1_finite_ab.txt 2>&1
  grep -v amdgpu $OUT/a1_finite_ab.txt $OUT/go2w_finite_ab.txt $OUT/g1_finite_ab.txt
  ;;
q)
  # the two test files whose bounds moved after call z3 (GR1's factor on the free-running rewards; the reward-term floor of spec-vs-interpreter)
  timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_specs.py -m gpu -q > $OUT/pytest_parity_specs.log 2>&1; echo "rc=$?" >> $OUT/pytest_parity_specs.log; tail -5 $OUT/pytest_parity_specs.log
  mv gpurun_out/spec_vs_interpreter.jsonl $OUT/ 2>/dev/null
  ;;
r)
  # the scheduler: LLVM's max-ILP strategy (-mllvm -amdgpu-sched-strategy=max-ilp) instead of the default max-occupancy one - these kernels
  # run one wavefront per SIMD whatever the register count, so there is no occupancy to protect
  timeout 600 python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 --steady $V/cur_34.so $V/ilp_34.so > $OUT/a1_sched_ab.txt 2>&1
  timeout 600 python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 3 --steady $V/cur_78.so $V/ilp_78.so > $OUT/g1_sched_ab.txt 2>&1
  grep -v amdgpu $OUT/a1_sched_ab.txt $OUT/g1_sched_ab.txt
  ;;
s)
  # the heightfield with rows ix, ix + 1 interleaved (env_step.h RL_TERRAIN_PAIRS: one 16-byte load per query) against the plain row-major grid
  # (two 8-byte loads): the specialised A1, Go2W and G1 kernels, one call
  timeout 600 python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 --steady $V/cur_34.so $V/hf2_34.so > $OUT/a1_hf2_ab.txt 2>&1
  timeout 600 python tools/ab_bench.py --task $GO2W --num-envs 4096 --rounds 3 --steady $V/cur_1044.so $V/hf2_1044.so > $OUT/go2w_hf2_ab.txt 2>&1
  timeout 600 python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 3 --steady $V/cur_78.so $V/hf2_78.so > $OUT/g1_hf2_ab.txt 2>&1
  grep -v amdgpu $OUT/a1_hf2_ab.txt $OUT/go2w_hf2_ab.txt $OUT/g1_hf2_ab.txt
  ;;
t)
  # G1: the next joint's link record (and the outward pass's U / D, u / D of all limb joints) read ahead of the chain that consumes them
  # (-DRL_REC_PREFETCH), against the tree; then the spec-vs-interpreter tests with the contact timers counted apart from the state bound
  timeout 600 python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 3 --steady $V/cur2_78.so $V/pref_78.so $V/pref2_78.so > $OUT/g1_prefetch_ab.txt 2>&1
  grep -v amdgpu $OUT/g1_prefetch_ab.txt
  timeout 900 python -m pytest tests/test_gpu_specs.py -m gpu -q > $OUT/pytest_specs.log 2>&1; echo "rc=$?" >> $OUT/pytest_specs.log; tail -4 $OUT/pytest_specs.log
  mv gpurun_out/spec_vs_interpreter.jsonl $OUT/ 2>/dev/null
  ;;
u)
  # G1 read-ahead, second step: + the trunk accumulators and the next joint's axis / origin (pf3 = the tree), + the outward pass's axes / origins
  # in its batch (pf4, -DRL_OUT_AXP_BATCH), against no read-ahead at all (nopf)
  timeout 600 python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 3 --steady $V/nopf_78.so $V/pref2_78.so $V/pf3_78.so $V/pf4_78.so > $OUT/g1_prefetch2_ab.txt 2>&1
  grep -v amdgpu $OUT/g1_prefetch2_ab.txt
  ;;
v)
  # G1: the trunk joints' table reads as one batch in front of the kinematic chain + the link twist kept from the rigid record for the contacts
  timeout 600 python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 3 --steady $V/pf4_78.so $V/fin2_78.so > $OUT/g1_batch3_ab.txt 2>&1
  grep -v amdgpu $OUT/g1_batch3_ab.txt
  ;;
w)
  # A1: the observation rows as streaming (non-temporal) stores, and one word of every line of the wavefront's state tiles requested before the
  # table image is staged (the state's latency under the staging round trip), each and both
  timeout 600 python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 --steady $V/cur3_34.so $V/nt_34.so $V/touch_34.so $V/nttouch_34.so > $OUT/a1_nt_touch_ab.txt 2>&1
  grep -v amdgpu $OUT/a1_nt_touch_ab.txt
  ;;
x)
  # the heightfield's lines as streaming (non-temporal) loads, alone and with the streaming observation stores + the early state touch of call w
  timeout 600 python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 3 --steady $V/cur3_34.so $V/tnt_34.so $V/nttouch_34.so $V/tntall_34.so > $OUT/a1_terrain_nt_ab.txt 2>&1
  timeout 600 python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 3 --steady $V/cur3_78.so $V/tnt_78.so > $OUT/g1_terrain_nt_ab.txt 2>&1
  grep -v amdgpu $OUT/a1_terrain_nt_ab.txt $OUT/g1_terrain_nt_ab.txt
  ;;
y)
  # the collection loop (actor + critic pair, act, env step + record, GAE as one hipGraph) around the env library of the tree and around the
  # variant with streaming observation stores + the early state touch (call w): what the NEXT kernel finds in L2 is part of this loop's time
  for rep in 1 2; do
    RL_ENV_LIB=$GRAFT_REPO_ROOT/$V/cur3_34.so timeout 300 python tools/bench_collect.py $A1 4096 40 2>/dev/null | tail -1 | sed "s/^/cur3    /" >> $OUT/collect_nt_ab.txt
    RL_ENV_LIB=$GRAFT_REPO_ROOT/$V/nttouch_34.so timeout 300 python tools/bench_collect.py $A1 4096 40 2>/dev/null | tail -1 | sed "s/^/nttouch /" >> $OUT/collect_nt_ab.txt
  done
  cut -c1-200 $OUT/collect_nt_ab.txt
  ;;
aa)
  # the act kernel's work in the epilogue of the actor + critic launch (include/rl_act.h): collection tests, then the collection loop with and without
  timeout 900 python -m pytest tests/test_gpu_collect.py tests/test_rollout.py tests/test_policy.py tests/test_gpu_train.py -m gpu -q -x > $OUT/pytest_collect.log 2>&1; echo "rc=$?" >> $OUT/pytest_collect.log; tail -6 $OUT/pytest_collect.log
  for rep in 1 2; do
    RL_FUSED_ACT=0 timeout 300 python tools/bench_collect.py $A1 4096 40 2>/dev/null | tail -1 | sed "s/^/act kernel  /" >> $OUT/collect_fused_ab.txt
    RL_FUSED_ACT=1 timeout 300 python tools/bench_collect.py $A1 4096 40 2>/dev/null | tail -1 | sed "s/^/fused       /" >> $OUT/collect_fused_ab.txt
  done
  RL_FUSED_ACT=0 timeout 300 python tools/bench_collect.py $G1 2048 20 2>/dev/null | tail -1 | sed "s/^/act kernel  /" >> $OUT/collect_fused_ab.txt
  RL_FUSED_ACT=1 timeout 300 python tools/bench_collect.py $G1 2048 20 2>/dev/null | tail -1 | sed "s/^/fused       /" >> $OUT/collect_fused_ab.txt
  cut -c1-200 $OUT/collect_fused_ab.txt
  ;;
ab)
  # kernel traces of the collection loop with the act kernel and with the act epilogue
  for m in 0 1; do
    ( cd /tmp && RL_FUSED_ACT=$m timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_$m -- python $GRAFT_REPO_ROOT/tools/bench_collect.py $A1 4096 20 > $GRAFT_REPO_ROOT/$OUT/collect_$m.txt 2> /dev/null )
    DB=$(find $OUT/prof_$m -name "*.db" | head -1); python tools/rocpd_summary.py $DB > $OUT/collect_trace_$m.txt 2>&1; rm -rf $OUT/prof_$m
    tail -1 $OUT/collect_$m.txt | cut -c1-200; head -9 $OUT/collect_trace_$m.txt | cut -c1-150
  done
  ;;
ac)
  # the epilogue in the 16-row pair kernel too (batches below 4096 rows: G1 2048): collection tests, then G1 and A1 with and without
  timeout 900 python -m pytest tests/test_gpu_collect.py tests/test_rollout.py tests/test_policy.py tests/test_gpu_train.py -m gpu -q -x > $OUT/pytest_collect.log 2>&1; echo "rc=$?" >> $OUT/pytest_collect.log; tail -4 $OUT/pytest_collect.log
  for rep in 1 2; do
    RL_FUSED_ACT=0 timeout 300 python tools/bench_collect.py $G1 2048 20 2>/dev/null | tail -1 | sed "s/^/act kernel  /" >> $OUT/collect_fused_ab.txt
    RL_FUSED_ACT=1 timeout 300 python tools/bench_collect.py $G1 2048 20 2>/dev/null | tail -1 | sed "s/^/fused       /" >> $OUT/collect_fused_ab.txt
  done
  RL_FUSED_ACT=0 timeout 300 python tools/bench_collect.py $A1 4096 40 2>/dev/null | tail -1 | sed "s/^/act kernel  /" >> $OUT/collect_fused_ab.txt
  RL_FUSED_ACT=1 timeout 300 python tools/bench_collect.py $A1 4096 40 2>/dev/null | tail -1 | sed "s/^/fused       /" >> $OUT/collect_fused_ab.txt
  cut -c1-200 $OUT/collect_fused_ab.txt
  ;;
ad)
  # every task id on the last tree, with the library's kernel and specialised at run time (the sweep of call e, after the packed pairs, the
  # flags and the read-ahead)
  timeout 2400 python tools/bench_every_task.py --jit > $OUT/all_tasks_jit.txt 2> $OUT/all_tasks_jit.err; cat $OUT/all_tasks_jit.txt | cut -c1-250
  ;;
zz|zz3|zz4)
  # THE LAST TREE (after call f: reward kinds 31-38 in the specialised evaluation - templates the built-in Specs do not instantiate): the whole GPU tier,
  # smoke(), the default bench line and the driver's flags
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -4 $OUT/pytest_gpu.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log
  grep -E "smoke|rc=|Error" $OUT/smoke.log | tail -8
  timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  timeout 100 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> /dev/null
  python -c "
import json
for n in ('bench_default','bench_driver_flags'):
    d=json.load(open('$OUT/%s.json' % n)); print(n, 'value %.2f M  ms_per_step %.4f  kernel_ms %.4f  roofline.frac %.4f' % (d['value']/1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']), d['config'].get('step_kernel'), d.get('large_batch',{}).get('value'), d.get('mid_batch',{}).get('value'), d.get('cpu_baseline',{}).get('value'))"
  ;;
z|z2|z3|z4|z5)
  # FINAL TREE: the whole GPU tier, smoke(), the bench lines of the BASELINE configs, kernel traces + counter passes, phase clocks, the collection loop
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -4 $OUT/pytest_gpu.log
  mkdir -p $OUT/teacher_forced && mv gpurun_out/teacher_forced_*.json $OUT/teacher_forced/ 2>/dev/null
  mv gpurun_out/spec_vs_interpreter.jsonl gpurun_out/train_distributed_8ranks.json $OUT/ 2>/dev/null
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log
  grep -E "smoke|rc=|Error" $OUT/smoke.log | tail -8
  timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_driver_flags.json 2> /dev/null
  for t in $GO2 $GO2W; do timeout 200 python bench.py --task $t --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_$(echo $t | cut -d- -f6).json 2>/dev/null; done
  timeout 200 python bench.py --task $G1 --num-envs 2048 --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_G1.json 2>/dev/null
  timeout 200 python bench.py --task RobotLab-Isaac-Velocity-Flat-Unitree-A1-v0 --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_A1_Flat.json 2>/dev/null
  python - <<PY | tee $OUT/baseline_configs.txt
import json
for n in ("bench_default", "bench_driver_flags", "bench_Go2", "bench_Go2W", "bench_G1", "bench_A1_Flat"):
    try:
        d = json.load(open("$OUT/%s.json" % n))
    except Exception as ex:
        print(n, "unreadable", ex); continue
    print("%-20s %-62s value %7.2f M env-steps/s  ms_per_step %.4f  kernel_ms %.4f  roofline.frac %.4f  resets in window %s  %s" % (n, d["config"]["workload"].split(",")[0], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("window", {}).get("envs_reset_in_window"), d["config"].get("step_kernel")))
d = json.load(open("$OUT/bench_default.json"))
for leg in ("mid_batch", "large_batch"):
    print(leg, {k: d.get(leg, {}).get(k) for k in ("envs_per_gpu", "value", "ms_per_step", "roofline_frac", "envs_per_wavefront")})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind", "per_core")}, d["cpu_baseline"].get("reference_terms_cpu", {}))
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
  timeout 300 python tools/bench_collect.py $A1 4096 40 2>/dev/null | tail -1 > $OUT/collect.txt
  timeout 300 python tools/bench_collect.py $G1 2048 20 2>/dev/null | tail -1 >> $OUT/collect.txt
  cat $OUT/collect.txt
  timeout 120 python tools/train_demo.py --iterations 300 2>/dev/null | tail -4 > $OUT/train_demo_a1_flat_300.txt; cat $OUT/train_demo_a1_flat_300.txt
  ;;
esac
