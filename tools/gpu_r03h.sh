#!/bin/bash
# Round 3, eighth GPU call: the cooperative (term-serial, divergence-free) reward evaluation against the lane-per-term one of the
# commit before - one-call A/B on A1 / Go2 / Go2W / G1 and on the one-lane-per-limb mapping at 16384 envs -, then the whole GPU tier,
# smoke, the collection loop and the bench line on the new build.
OUT=gpurun_out/r03h
mkdir -p $OUT
export TMPDIR=/tmp
V=robot_lab_amd/csrc/variants
A1=RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0
GO2=RobotLab-Isaac-Velocity-Rough-Unitree-Go2-v0
GO2W=RobotLab-Isaac-Velocity-Rough-Unitree-Go2W-v0
G1=RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0
T0=$SECONDS
lap() { echo "[lap] $1 at $((SECONDS - T0)) s" | tee -a $OUT/laps.txt; }
python tools/ab_bench.py --task $A1 --num-envs 4096 --rounds 2 $V/prev_34.so $V/coop_34.so 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
python tools/ab_bench.py --task $GO2 --num-envs 4096 --rounds 1 $V/prev_34.so $V/coop_34.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
python tools/ab_bench.py --task $GO2W --num-envs 4096 --rounds 1 $V/prev_1044.so $V/coop_1044.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
python tools/ab_bench.py --task $G1 --num-envs 2048 --rounds 1 $V/prev_74.so $V/coop_74.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
RL_ENV_SUB=1 python tools/ab_bench.py --task $A1 --num-envs 16384 --rounds 1 $V/prev_31.so $V/coop_31.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
lap ab
timeout 1500 python -m pytest tests -m gpu -q -rf -s > $OUT/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|FAILED|rc=" $OUT/pytest_gpu.log | cut -c1-400 | tail -30
cp gpurun_out/episode_stats_*.json gpurun_out/teacher_forced_*.json $OUT/ 2>/dev/null
lap pytest
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -4
timeout 200 python tools/bench_collect.py 2>&1 | grep -v amdgpu.ids | tee $OUT/collect.txt
timeout 200 python tools/sweep_envs.py $A1 4096,8192,16384,65536 2>&1 | grep -v amdgpu.ids | tee $OUT/sweep.txt
lap collect_sweep
