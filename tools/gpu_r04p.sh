#!/bin/bash
# Round 4, call p: the whole GPU tier, smoke and the bench lines on the tree with the granule scratchpad and the sphere vectors (the
# LDS layouts changed), and the final kernels against the tree before the granule scratchpad (base_* = 0b0c8f1+: packed per-joint constants) in one call.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_r04p.sh'
TAG=r04p
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -7 | tee $OUT/smoke.txt
python bench.py > $OUT/bench_default.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_driver_flags.json 2> /dev/null
python bench.py --steps 300 --warmup 50 --no-cpu-baseline --large-batch-envs 0 --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 > $OUT/g1_bench.json 2> /dev/null
python - <<PY | tee $OUT/summary.txt
import json
for n in ("bench_default", "bench_driver_flags", "g1_bench"):
    d = json.load(open("$OUT/%s.json" % n))
    print(n, "value %.2f M" % (d["value"] / 1e6), "ms_per_step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"], d["window"]["envs_reset_in_window"], d["window"]["mean_bodies_in_contact_at_end"])
d = json.load(open("$OUT/bench_default.json"))
print({k: d["cpu_baseline"][k] for k in ("value", "cores", "per_core", "repeats")})
for leg in ("large_batch", "mid_batch"):
    print(leg, {k: d.get(leg, {}).get(k) for k in ("envs_per_gpu", "value", "ms_per_step", "roofline_frac")})
PY
V=robot_lab_amd/csrc/variants
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0 --num-envs 4096 --rounds 2 $V/base_34.so $V/final_34.so 2>&1 | grep -v amdgpu.ids | tee $OUT/final_ab.txt
python tools/ab_bench.py --task RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0 --num-envs 2048 --rounds 2 $V/base_78.so $V/final_78.so 2>&1 | grep -v amdgpu.ids | tee -a $OUT/final_ab.txt
