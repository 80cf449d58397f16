#!/bin/bash
# Ablation builds of the A1 instance: the kernel with one stage removed each (results are wrong, timings are what is wanted).
# The time a stage REALLY costs - overlap with its neighbours included - is the difference to the full kernel; the phase clock
# (tools/phase_clock.py) over-counts because every stamp drains the memory pipelines.
#   bash tools/ablate.sh            # builds robot_lab_amd/csrc/variants/abl_*.so;  then on the GPU box:
#   python tools/ab_bench.py --rounds 2 robot_lab_amd/csrc/variants/abl_*.so
cd "$(dirname "$0")/../robot_lab_amd/csrc"
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -mllvm -amdgpu-remove-redundant-endcf=false -std=c++17 -shared -fPIC -DRL_ENV_SINGLE_TU -DRL_ENV_ONLY=${1:-34}"
$B -o variants/abl_full.so rl_env.hip &
$B -DRL_ABL_NO_REWARDS -o variants/abl_norewards.so rl_env.hip &
$B -DRL_ABL_NO_OBS -o variants/abl_noobs.so rl_env.hip &
$B -DRL_ABL_NO_SCAN -o variants/abl_noscan.so rl_env.hip &
wait
$B -DRL_ABL_SUBSTEPS=0 -o variants/abl_sub0.so rl_env.hip &
$B -DRL_ABL_SUBSTEPS=2 -o variants/abl_sub2.so rl_env.hip &
$B -DRL_ABL_NO_CONTACTS -o variants/abl_nocontacts.so rl_env.hip &
$B -DRL_ABL_NO_REWARDS -DRL_ABL_NO_OBS -DRL_ABL_SUBSTEPS=0 -o variants/abl_skeleton.so rl_env.hip &
wait
ls -la variants/abl_*.so
