#!/bin/bash
# Round 4, second call: the 32-lanes-per-env mapping of the trunk + limbs instances (eight sub-lanes per limb) on the GPU - parity tier
# for the trunk robots, then G1 Rough timed in both mappings at 1024 .. 8192 envs in ONE call; the cold-vs-steady-state A/B of the
# A1 headline window; the thread scaling of the CPU baseline.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_r04b.sh'
TAG=r04b
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_canary.py tests/test_gpu_self_collision.py tests/test_gpu_teacher_forced.py tests/test_gpu_all_tasks.py tests/test_gpu_episode_stats.py tests/test_gpu_edge_cases.py -m gpu -q -x -k "G1 or GR1 or ATOM01 or Xbot or Gen1 or Loong or Z1 or self_contact or stays_finite or statistics" > $OUT/pytest_trunk.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_trunk.log
tail -6 $OUT/pytest_trunk.log
G1=RobotLab-Isaac-Velocity-Rough-Unitree-G1-v0
for sub in 4 8; do
  echo "== RL_ENV_SUB=$sub" | tee -a $OUT/g1_sweep.txt
  RL_ENV_SUB=$sub RL_ENV_DEBUG=1 python tools/sweep_envs.py $G1 512,1024,2048,4096,8192 2>&1 | grep -v amdgpu.ids | grep -v "lane program" | tee -a $OUT/g1_sweep.txt
done
echo "== default selection" | tee -a $OUT/g1_sweep.txt
python tools/sweep_envs.py $G1 1024,2048,4096 2>&1 | grep -v amdgpu.ids | tee -a $OUT/g1_sweep.txt
echo "== GR1T1 sub 4 / 8" | tee -a $OUT/g1_sweep.txt
for sub in 4 8; do RL_ENV_SUB=$sub python tools/sweep_envs.py RobotLab-Isaac-Velocity-Rough-FFTAI-GR1T1-v0 2048 2>&1 | grep -v amdgpu.ids | tee -a $OUT/g1_sweep.txt; done
# cold vs steady-state window of the headline (same binary, same call)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch-envs 0 --preroll 0 > $OUT/bench_cold.json 2> /dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_steady.json 2> /dev/null
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --large-batch-envs 0 > $OUT/bench_steady_1000.json 2> /dev/null
python - <<PY | tee $OUT/cold_vs_steady.txt
import json
for n in ("bench_cold", "bench_steady", "bench_steady_1000"):
    d = json.load(open("$OUT/%s.json" % n))
    print(n, "value %.1f M" % (d["value"] / 1e6), "ms_per_step %.4f" % d["ms_per_step"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], d["window"])
PY
# CPU baseline: thread scaling (is the box's quota what limits the 128-thread figure?)
cat /sys/fs/cgroup/cpu.max 2>/dev/null | tee $OUT/cpu_scaling.txt
python - <<'PY' 2>&1 | tee -a $OUT/cpu_scaling.txt
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
import bench
lib = bench.build_host_port()
cpus = bench.physical_cores()
print("physical cores", len(cpus), "affinity", len(os.sched_getaffinity(0)))
code = ("import sys, time, numpy as np\nsys.path.insert(0, %r)\nfrom robot_lab_amd.capi import NativeEnv\nfrom robot_lab_amd.scene import build_world, load_bundle\n"
        "desc, extra = load_bundle('RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0')\nh, to, eo = build_world(desc, extra, 4096, 0)\nnat = NativeEnv(desc, h, to, eo, 4096, 42, 0, %r)\nnat.reset()\n"
        "a = np.random.default_rng(0).uniform(-1, 1, (8, 4096, 12)).astype(np.float32)\nnat.step(a[0].ctypes.data)\nt0 = time.perf_counter()\nfor s in range(STEPS): nat.step(a[s %% 8].ctypes.data)\nprint(4096 * STEPS / (time.perf_counter() - t0))\n" % (os.getcwd(), lib))
for t in (1, 8, 16, 32, 64, 128):
    env = dict(os.environ, RL_EMU_TEAMS=str(t), RL_EMU_CPUS=",".join(map(str, cpus[:t])), RL_EMU_FIBERS="1")
    steps = max(3, min(400, 3 * t))
    p = subprocess.run([sys.executable, "-c", code.replace("STEPS", str(steps))], env=env, capture_output=True, text=True)
    v = float(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else 0.0
    print(f"threads {t:4d}: {v:10.0f} env-steps/s  {v / t:8.0f} per thread")
PY
