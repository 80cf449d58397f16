#!/bin/bash
# Round 4, call n: the whole GPU tier + the bench lines on the tree with the packed per-joint constants (the lane-table layout changed).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r04n.sh'
TAG=r04n
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_driver_flags.json 2> /dev/null
python bench.py --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 > $OUT/g1_bench.json 2> /dev/null
python - <<PY | tee $OUT/summary.txt
import json
for n in ("bench_driver_flags", "g1_bench"):
    d = json.load(open("$OUT/%s.json" % n))
    print(n, "value %.2f M" % (d["value"] / 1e6), "ms_per_step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d["window"]["envs_reset_in_window"], d["window"]["mean_bodies_in_contact_at_end"])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -7 | tee $OUT/smoke.txt
