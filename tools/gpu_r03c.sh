#!/bin/bash
# Round 3, third GPU call: the 32-row split-bf16 MLP kernel (correctness + timing by row count), the env kernel with the reward
# descriptors pinned in registers (parity subset + A/B), the distributional parity test, the first round-3 bench line.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r03c.sh'
OUT=gpurun_out/r03c
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
timeout 600 python -m pytest tests/test_policy.py tests/test_gpu_collect.py -m gpu -q -rf > $OUT/pytest_mlp.log 2>&1; echo "pytest mlp rc=$?" >> $OUT/pytest_mlp.log
tail -6 $OUT/pytest_mlp.log
for rt in 1 2; do for rows in 2048 4096 8192; do
  echo "== RL_MLP_SPLIT_RT=$rt rows=$rows" | tee -a $OUT/policy.txt
  RL_MLP_SPLIT_RT=$rt timeout 300 python tools/bench_pair.py $rows 2>&1 | grep -v amdgpu.ids | tee -a $OUT/policy.txt
done; done
for rt in 1 2; do
  echo "== RL_MLP_SPLIT_RT=$rt" | tee -a $OUT/collect.txt
  RL_MLP_SPLIT_RT=$rt timeout 300 python tools/bench_collect.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/collect.txt
done
python tools/bench_collect.py RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 2048 2>&1 | grep -v amdgpu.ids | tee -a $OUT/collect.txt
python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 2 $V/r02_34.so $V/new_34.so $V/pin_34.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_a1.txt
timeout 900 python -m pytest tests/test_gpu_canary.py tests/test_gpu_teacher_forced.py tests/test_gpu_episode_stats.py "tests/test_gpu_parity.py::test_short_horizon_parity" -m gpu -q -rf -s -k "canary or episode or ((A1 or G1 or Go2) and not Flat) or HandStand" > $OUT/pytest_env.log 2>&1; echo "pytest env rc=$?" >> $OUT/pytest_env.log
grep -E "episode-stats|passed|failed|FAILED|rc=" $OUT/pytest_env.log | cut -c1-600 | tail -12
cp gpurun_out/episode_stats_*.json $OUT/ 2>/dev/null
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
