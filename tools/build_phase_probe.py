#!/usr/bin/env python
"""Build tools/_probe/librl_env_stamp.so: a copy of the HIP library with s_memtime stamps at the phase
boundaries of the env-step kernel (read back by tools/phase_clock.py).  The product sources are not
touched: the stamps are patched into a scratch copy of robot_lab_amd/csrc.

    python tools/build_phase_probe.py            # phases of step()
    python tools/build_phase_probe.py rewards    # one stamp per reward term (first 13 terms)
"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else "phases"
work = "/tmp/rl_phase_probe"
shutil.rmtree(work, ignore_errors=True)
shutil.copytree(os.path.join(ROOT, "robot_lab_amd", "csrc"), os.path.join(work, "csrc"))
os.makedirs(os.path.join(work, "include"))
shutil.copy(os.path.join(ROOT, "include", "rl_env.h"), os.path.join(work, "include"))


def edit(name, fn):
    p = os.path.join(work, "csrc", name)
    s = fn(open(p).read())
    open(p, "w").write(s)


def must(s, old, new):
    assert old in s, old
    return s.replace(old, new)


STAMP = ('#define RL_STAMP(S, i) do { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0) '
         'reinterpret_cast<long long*>((S).rew_terms + (size_t)24 * (S).Npad)[blockIdx.x * 16 + (i)] = '
         '(long long)__builtin_readcyclecounter(); } while (0)\n')


def hip(s):
    s = s.replace('#include "../../include/rl_env.h"', f'#include "{work}/include/rl_env.h"')
    s = must(s, '#include "env_aos.h"', STAMP + '#include "env_aos.h"')
    if mode == "phases":
        s = must(s, "  const int lane = threadIdx.x;\n  {  // stage", "  const int lane = threadIdx.x;\n  RL_STAMP(S, 0);\n  {  // stage")
        s = must(s, "  __syncthreads();\n  // LDS after the tables", "  __syncthreads();\n  RL_STAMP(S, 1);\n  // LDS after the tables")
    return s


def host(s):
    return s.replace('#include "../../include/rl_env.h"', f'#include "{work}/include/rl_env.h"')


def terms(s):
    s = must(s, '#include "env_step.h"', '#include "env_step.h"\n#ifndef RL_STAMP\n#define RL_STAMP(S, i)\n#endif')
    if mode == "phases":
        s = must(s, "for (int s = 0; s < T.decimation; ++s) this->substep(q_tgt, qd_tgt);",
                 "RL_STAMP(S, 2);\n    for (int s = 0; s < T.decimation; ++s) { this->substep(q_tgt, qd_tgt); RL_STAMP(S, 3 + s); }")
        s = must(s, "    float rew = compute_rewards(terminated);", "    RL_STAMP(S, 7);\n    float rew = compute_rewards(terminated);\n    RL_STAMP(S, 9);")
        s = must(s, "    // per-term outputs + episode sums: staged through LDS", "    RL_STAMP(S, 8);\n    // per-term outputs + episode sums: staged through LDS")
        s = must(s, "    // 9 observations\n    observations();\n    this->store();\n    store_task();",
                 "    RL_STAMP(S, 10);\n    observations();\n    RL_STAMP(S, 13);\n    this->store();\n    store_task();\n    RL_STAMP(S, 14);")
        s = must(s, "      write_obs(sp, oc, T.policy, T.n_policy, T.policy_dim, T.policy_corrupt != 0, 0u, cy, sy, scan_p);\n",
                 "      write_obs(sp, oc, T.policy, T.n_policy, T.policy_dim, T.policy_corrupt != 0, 0u, cy, sy, scan_p);\n      RL_STAMP(S, 11);\n")
        s = must(s, "    ctx.flush_obs(S.obs_policy, T.policy_dim, 0);", "    RL_STAMP(S, 12);\n    ctx.flush_obs(S.obs_policy, T.policy_dim, 0);")
    elif mode == "obs":
        s = must(s, "    derive();\n    float* sp = ctx.obs_stage(0);", "    RL_STAMP(S, 0);\n    derive();\n    float* sp = ctx.obs_stage(0);")
        s = must(s, "    n = ctx.uniform_i(n);\n    bool any_noise = false;", "    n = ctx.uniform_i(n);\n    bool any_noise = false;\n    const int sbase = noise_base == 0u ? 1 : 8;\n    RL_STAMP(S, sbase - 1 + (noise_base == 0u ? 1 : 0));")
        s = must(s, "      obs_term(O, oc, stage, corrupt, cy, sy, scan_p);\n    }\n    if (any_noise)", "      obs_term(O, oc, stage, corrupt, cy, sy, scan_p);\n      if (noise_base == 0u && i < 6) RL_STAMP(S, 2 + i);\n    }\n    if (any_noise)")
        s = must(s, "    if (any_noise) add_noise(stage, terms, n, dim, noise_base);", "    if (any_noise) add_noise(stage, terms, n, dim, noise_base);\n    if (noise_base == 0u) RL_STAMP(S, 8);")
    elif mode == "obsfine":
        s = must(s, "    n = ctx.uniform_i(n);\n    bool any_noise = false;", "    n = ctx.uniform_i(n);\n    bool any_noise = false;\n    if (noise_base == 0u) RL_STAMP(S, 0);")
        s = must(s, "      nxt = fetch_obs(terms, i + 1, n);\n      any_noise", "      nxt = fetch_obs(terms, i + 1, n);\n      if (noise_base == 0u && i < 5) RL_STAMP(S, 1 + 3 * i);\n      any_noise")
        s = must(s, "      obs_term(O, oc, stage, corrupt, cy, sy, scan_p);\n    }\n    if (any_noise)", "      if (noise_base == 0u && i < 5) RL_STAMP(S, 2 + 3 * i);\n      obs_term(O, oc, stage, corrupt, cy, sy, scan_p);\n      if (noise_base == 0u && i < 5) RL_STAMP(S, 3 + 3 * i);\n    }\n    if (any_noise)")
    elif mode == "rewards":
        s = must(s, "    RewCtx rc{", "    RL_STAMP(S, 0);\n    RewCtx rc{")
        s = must(s, "    float total = 0.f;\n    float* rstage = ctx.rew_stage();", "    RL_STAMP(S, 1);\n    float total = 0.f;\n    float* rstage = ctx.rew_stage();")
        s = must(s, "        if (li == 0) rstage[t] = val;\n      }\n    } else {", "        if (li == 0) rstage[t] = val;\n        if (t < 13) RL_STAMP(S, 2 + t);\n      }\n    } else {")
    return s


def step_h(s):
    if mode == "contacts":
        s = must(s, '#include "env_tables.h"\n\nnamespace rl {', '#include "env_tables.h"\n#ifndef RL_STAMP\n#define RL_STAMP(S, i)\n#endif\n\nnamespace rl {')
        s = must(s, "    const float dt = u.dt;\n#pragma unroll 1\n    for (int g = (SUB == 1 ? 0 : sub); g <= CL; g += SUB) {", "    const float dt = u.dt;\n    RL_STAMP(S, 0);\n#pragma unroll 1\n    for (int g = (SUB == 1 ? 0 : sub); g <= CL; g += SUB) {")
        s = must(s, "      float phi[SPL];\n      V3 nw[SPL];\n      bool touching = false;", "      RL_STAMP(S, 1);\n      float phi[SPL];\n      V3 nw[SPL];\n      bool touching = false;")
        s = must(s, "      if (!ctx.any(touching)) continue;  // most link groups", "      RL_STAMP(S, 2);\n      if (!ctx.any(touching)) continue;  // most link groups")
        s = must(s, "        for (int s = 0; s < SPL; ++s) one_slot(s, rad[s], cb[s], phi[s], nw[s]);", "        for (int s = 0; s < SPL; ++s) { one_slot(s, rad[s], cb[s], phi[s], nw[s]); RL_STAMP(S, 3 + s); }")
        s = must(s, "        Contact c = contact_from_phi(C, Rwb, V0, qd, g, s, rad_s, cb_s, phi_s, nw_s);\n        if (c.act) {", "        Contact c = contact_from_phi(C, Rwb, V0, qd, g, s, rad_s, cb_s, phi_s, nw_s);\n        if (s == 0) RL_STAMP(S, 6);\n        if (c.act) {")
        return s
    if mode != "substep":
        return s
    s = must(s, '#include "env_tables.h"\n\nnamespace rl {', '#include "env_tables.h"\n#ifndef RL_STAMP\n#define RL_STAMP(S, i)\n#endif\n\nnamespace rl {')
    s = must(s, "    float tau_e[JX], pd_diag[JX], pd_rhs[JX];\n    actuators(", "    RL_STAMP(S, 0);\n    float tau_e[JX], pd_diag[JX], pd_rhs[JX];\n    actuators(")
    s = must(s, "    // ---- contacts first (they only need the kinematics)", "    RL_STAMP(S, 1);\n    // ---- contacts first (they only need the kinematics)")
    s = must(s, "      contact_pass1(C, Rwb, V0, slot_valid, Ua, ra, active_mask);\n", "      contact_pass1(C, Rwb, V0, slot_valid, Ua, ra, active_mask);\n      RL_STAMP(S, 2);\n      RL_STAMP(S, 3);\n")
    s = must(s, "      contact_pass1(C, Rwb, V0, slot_valid, Uc, rvc, active_mask);\n", "      contact_pass1(C, Rwb, V0, slot_valid, Uc, rvc, active_mask);\n      RL_STAMP(S, 2);\n")
    s = must(s, "      if constexpr (LDSU) {\n        LdsVec<LBS> U{", "      RL_STAMP(S, 3);\n      if constexpr (LDSU) {\n        LdsVec<LBS> U{")
    s = must(s, "    // ---- Schur complement of the limb block", "    RL_STAMP(S, 4);\n    // ---- Schur complement of the limb block")
    s = must(s, "    float nu0[NB];\n    {  // NB x NB Cholesky solve", "    RL_STAMP(S, 5);\n    float nu0[NB];\n    {  // NB x NB Cholesky solve")
    s = must(s, "    // ---- contact sensor: net contact force per body with the NEW velocities", "    RL_STAMP(S, 6);\n    // ---- contact sensor: net contact force per body with the NEW velocities")
    s = must(s, "    // trunk-link bodies can be fed by several lanes", "    RL_STAMP(S, 7);\n    // trunk-link bodies can be fed by several lanes")
    s = must(s, "    // [UPSTREAM B5] ContactSensor: history roll", "    RL_STAMP(S, 8);\n    // [UPSTREAM B5] ContactSensor: history roll")
    s = must(s, "    // ---- integrate (semi-implicit Euler", "    RL_STAMP(S, 9);\n    // ---- integrate (semi-implicit Euler")
    s = must(s, "    pos = pos + dt * vlin;\n  }", "    pos = pos + dt * vlin;\n    RL_STAMP(S, 10);\n  }")
    return s


edit("env_step.h", step_h)
edit("rl_env.hip", hip)
edit("rl_env_host.h", host)
edit("env_terms.h", terms)
out = os.path.join(ROOT, "tools", "_probe")
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "librl_env_stamp.so")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", lib, os.path.join(work, "csrc", "rl_env.hip")]
print(" ".join(cmd))
subprocess.check_call(cmd)
print("built", lib, "mode", mode)
