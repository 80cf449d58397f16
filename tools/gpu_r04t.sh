#!/bin/bash
# Round 4, call t: the trunk + limbs cases of the GPU tier on the final tree (after the table-image trim of call s).
#   /usr/local/graft/bin/gpurun --timeout 230 -- 'bash tools/gpu_r04t.sh'
TAG=r04t
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_self_collision.py tests/test_gpu_teacher_forced.py -m gpu -q -x -k "G1 or GR1T1 or Booster or self" > $OUT/pytest_trunk.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_trunk.log
tail -4 $OUT/pytest_trunk.log
