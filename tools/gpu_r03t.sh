#!/bin/bash
# Round 3: kernel mix of a PPO training run (rocprofv3 kernel trace of tools/train_demo.py), SQ counters of the 8-lane mapping at 8192 envs
OUT=gpurun_out/r03t
mkdir -p $OUT
export TMPDIR=/tmp
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_train -- python $GRAFT_REPO_ROOT/tools/train_demo.py --iterations 40 --print-every 20 --out $GRAFT_REPO_ROOT/$OUT > $GRAFT_REPO_ROOT/$OUT/train_under_rocprof.txt 2> $GRAFT_REPO_ROOT/$OUT/train.err )
python tools/rocpd_summary.py $(find $OUT/prof_train -name "*.db" | head -1) > $OUT/train_kernel_stats.txt 2>&1; rm -rf $OUT/prof_train
head -30 $OUT/train_kernel_stats.txt | cut -c1-150
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/$OUT/prof_sq -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 30 --no-cpu-baseline --large-batch-envs 0 --num-envs 8192 > $GRAFT_REPO_ROOT/$OUT/bench8192_under_sq.json 2> $GRAFT_REPO_ROOT/$OUT/sq.err )
python tools/rocpd_summary.py $(find $OUT/prof_sq -name "*.db" | head -1) > $OUT/pmc_sq_8192.txt 2>&1; rm -rf $OUT/prof_sq
grep "env_kernel" $OUT/pmc_sq_8192.txt | grep "SQ_" | grep ", 0, 2" | cut -c40-190
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS -d $GRAFT_REPO_ROOT/$OUT/prof_wait -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 30 --no-cpu-baseline --large-batch-envs 0 --num-envs 8192 > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/wait.err )
python tools/rocpd_summary.py $(find $OUT/prof_wait -name "*.db" | head -1) > $OUT/pmc_wait_8192.txt 2>&1; rm -rf $OUT/prof_wait
grep "env_kernel" $OUT/pmc_wait_8192.txt | grep "SQ_" | grep ", 0, 2" | cut -c40-190
