#!/bin/bash
# Round 4, call s (last GPU minutes): the instance table images without the host-only source arrays and the reward stage sized by the
# task's term count - smoke, canaries, lane mappings, the quadruped parity cases in every mapping, and the mapping choice at 16384 envs.
#   /usr/local/graft/bin/gpurun --timeout 280 -- 'bash tools/gpu_r04s.sh'
TAG=r04s
OUT=gpurun_out/$TAG
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT/smoke.txt
timeout 170 python -m pytest tests/test_gpu_canary.py tests/test_gpu_lane_mapping.py tests/test_gpu_parity.py -m gpu -q -x -k "A1 or Go2 or canary or mapping or lane" > $OUT/pytest_subset.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -4 $OUT/pytest_subset.log
for t in Unitree-A1 Unitree-Go2; do
  RL_ENV_DEBUG=1 timeout 60 python tools/sweep_envs.py RobotLab-Isaac-Velocity-Rough-$t-v0 16384,65536 2>&1 | grep -v amdgpu.ids | grep "rl_env: .*envs:\|^ *[0-9]\|^Robot" | tee -a $OUT/quadruped_mappings.txt
done
