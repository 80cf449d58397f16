#!/bin/bash
# Round 4, fifth call: whole GPU tier on the build with trunk pieces (Booster T1), hosted spine links and the limb-major trunk kernels;
# then the in-kernel phase clocks of the A1 (16 lanes per env) and G1 (32 lanes per env) step kernels - builds without spills since
# the stamp is force-inlined.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_r04e.sh'
TAG=r04e
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
V=robot_lab_amd/csrc/variants
RL_ENV_LIB=$V/clock_34.so python tools/phase_clock.py RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 4096 2>&1 | grep -v amdgpu.ids | tee $OUT/phase_clock_a1.txt
RL_ENV_LIB=$V/clock_78.so python tools/phase_clock.py RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 2048 2>&1 | grep -v amdgpu.ids | tee $OUT/phase_clock_g1.txt
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 --rounds 1 $V/fusedvel_78.so $V/clock_78.so 2>&1 | grep -v amdgpu.ids | tee $OUT/clock_overhead.txt
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 --num-envs 4096 --rounds 1 robot_lab_amd/csrc/librl_env_hip.so $V/clock_34.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/clock_overhead.txt
python tools/bench_every_task.py 2>&1 | grep -v amdgpu.ids > $OUT/all_tasks.txt
tail -12 $OUT/all_tasks.txt
