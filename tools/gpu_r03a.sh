#!/bin/bash
# Round 3, first GPU call: the widened GPU parity tier (kernel shapes, hard caps, canary, multi-rank), the multi-rank bench lines,
# phase clocks of the A1 / G1 step kernels, FETCH_SIZE / WRITE_SIZE calibration.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r03a.sh'
OUT=gpurun_out/r03a
mkdir -p $OUT
export TMPDIR=/tmp
nproc > $OUT/host_cores.txt
timeout 1000 python -m pytest tests -m gpu -q -rf --durations=15 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
cp gpurun_out/teacher_forced_*.json $OUT/ 2>/dev/null
# the multi-rank body of bench.py on this one GPU
RL_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 200 --warmup 20 > $OUT/bench_share_gpus2.json 2> $OUT/bench_share_gpus2.err
echo "share rc=$?"; tail -c 600 $OUT/bench_share_gpus2.json; tail -5 $OUT/bench_share_gpus2.err
RL_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err
echo "rccl rc=$?"; tail -c 400 $OUT/bench_rccl_world1.json; tail -5 $OUT/bench_rccl_world1.err
# where a wavefront's time goes, by phase (in-kernel shader clock)
RL_ENV_LIB=robot_lab_amd/csrc/variants/clock_34.so timeout 300 python tools/phase_clock.py 2>&1 | grep -v amdgpu.ids | tee $OUT/phase_clock_a1.txt
RL_ENV_LIB=robot_lab_amd/csrc/variants/clock_74.so timeout 300 python tools/phase_clock.py RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 2048 2>&1 | grep -v amdgpu.ids | tee $OUT/phase_clock_g1.txt
# FETCH_SIZE / WRITE_SIZE calibration on the kernel's own access pattern
tools/micro/stream_rows > $OUT/stream_rows.txt 2>&1; cat $OUT/stream_rows.txt
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c -d $GRAFT_REPO_ROOT/$OUT/prof_cal_$c -- $GRAFT_REPO_ROOT/tools/micro/stream_rows > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/cal_$c.err )
  DB=$(find $OUT/prof_cal_$c -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB > $OUT/cal_$c.txt 2>&1 || true
  rm -rf $OUT/prof_cal_$c
  grep -A8 "PMC counters" $OUT/cal_$c.txt
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $OUT/smoke.txt
