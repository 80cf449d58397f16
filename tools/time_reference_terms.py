#!/usr/bin/env python
"""Time the part of the REFERENCE that can run without IsaacLab: its own reward and observation term functions
(`/root/reference/source/robot_lab/.../velocity/mdp/rewards.py:22-687`, `observations.py`), imported unchanged through
robot_lab_amd.shims exactly as tools/gen_golden_terms.py does, evaluated with torch on the CPU (fp32, `set_num_threads(cores)`)
on a recorded simulator state of `--num-envs` environments: the TERM STACK ONLY - no physics, no actuators, no contact sensor, no
ray caster, no manager bookkeeping (all IsaacLab / PhysX, absent here).  So the figure is an upper bound on what the reference's CPU
path could reach on these cores, beside bench.py's `cpu_baseline` (the whole env step, our program on the host).

/root/reference exists in the build container only, so this is measured HERE and committed (profiles/r06_reference_terms_cpu.json);
bench.py quotes the file as an offline figure and never imports the reference.

    python tools/time_reference_terms.py [--task RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0] [--num-envs 4096] [--seconds 10]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..")]
import gen_golden_terms as gg  # noqa: E402  (installs the shims, imports robot_lab.tasks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="RobotLab-Isaac-Velocity-Rough-Unitree-A1-v0")
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--warm-steps", type=int, default=12, help="oracle steps that produce the recorded state")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--out", default=os.path.join(HERE, "..", "profiles", "r06_reference_terms_cpu.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    cfg = gg.parse_env_cfg(a.task, device="cpu")
    desc, spec = gg.compile_cfg(cfg)
    N = a.num_envs
    h, to, eo = gg.build_world(desc, gg.load_bundle(a.task)[1] if "Rough" in a.task else dict(env_spacing=2.5), N, 0)
    ora = gg.OracleEnv(desc, h, to, N, 3, eo)
    ora.reset()
    rng = np.random.default_rng(3)
    for _ in range(a.warm_steps):
        ora.step(rng.uniform(-1, 1, (N, ora.D)))
    env = gg.duck_env(ora, desc)

    def to_f32(obj, seen):  # the reference's tensors are fp32: every float64 tensor of the recorded state -> float32
        if id(obj) in seen:
            return
        seen.add(id(obj))
        items = obj.items() if isinstance(obj, dict) else (vars(obj).items() if hasattr(obj, "__dict__") else ())
        for k, v in list(items):
            if torch.is_tensor(v):
                if v.dtype == torch.float64:
                    (obj.__setitem__ if isinstance(obj, dict) else lambda kk, vv: setattr(obj, kk, vv))(k, v.float())
            elif isinstance(v, (dict,)) or hasattr(v, "__dict__"):
                if not callable(v) or hasattr(v, "data"):
                    to_f32(v, seen)

    to_f32(env, set())
    cmd32, act32, prev32 = torch.tensor(ora.vel_command_b, dtype=torch.float32), torch.tensor(ora.action, dtype=torch.float32), torch.tensor(ora.prev_action, dtype=torch.float32)
    env.command_manager.get_command = lambda name: cmd32
    env.action_manager.action, env.action_manager.prev_action = act32, prev32
    # the reward terms the RewardManager would call every step (weight != 0), instantiated once as the manager does
    rewards = []
    for name, term in vars(cfg.rewards).items():
        if term is None or not hasattr(term, "func") or term.weight == 0:
            continue
        f, params = term.func, gg.resolve(term.params, desc)
        if isinstance(f, type):
            f = f(term, env)
        rewards.append((name, f, params, float(term.weight)))
    # the observation terms of both groups whose function lives in the reference or in the shim's restatement of isaaclab.envs.mdp
    # (height_scan needs the ray caster: left out, as in the golden generator)
    from isaaclab.envs import mdp as up

    obs = []
    for gname in ("policy", "critic"):
        for name, term in vars(getattr(cfg.observations, gname)).items():
            if term is None or not hasattr(term, "func") or term.func is up.height_scan:
                continue
            obs.append((f"{gname}.{name}", term.func, gg.resolve(term.params, desc), term.clip, term.scale))

    def one_step():
        total = torch.zeros(N)
        for _, f, params, w in rewards:
            total += f(env, **params) * (w * ora.step_dt)  # RewardManager.compute: value * weight * dt, summed
        rows = []
        for _, f, params, clip, scale in obs:
            v = f(env, **params)
            if clip is not None:
                v = v.clip(clip[0], clip[1])
            if scale is not None:
                v = v * scale
            rows.append(v)
        return total, torch.cat(rows, -1)

    with torch.inference_mode():
        for _ in range(3):
            one_step()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < a.seconds:
            one_step()
            n += 1
        dt = time.perf_counter() - t0
        per = {}
        for name, f, params, w in rewards:
            t1 = time.perf_counter()
            for _ in range(20):
                f(env, **params)
            per[name] = (time.perf_counter() - t1) / 20 * 1e6
    out = dict(what="the reference's own VEL/mdp reward + observation term functions (imported unchanged through the shims), torch CPU fp32: TERM STACK ONLY "
                    "(no physics, actuators, sensors, ray caster or manager bookkeeping - IsaacLab / PhysX are absent)",
               task=a.task, num_envs=N, threads=a.threads, host="build container (not the GPU box: /root/reference exists only here)",
               torch=torch.__version__, reward_terms=len(rewards), observation_terms=len(obs), calls=n, seconds=dt,
               ms_per_term_stack_call=1e3 * dt / n, value=N * n / dt, unit="env-steps/s (term stack only)",
               reward_term_us={k: round(v, 1) for k, v in per.items()})
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
