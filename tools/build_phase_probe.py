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
        s = must(s, "  __syncthreads();\n  constexpr int TAB_F", "  __syncthreads();\n  RL_STAMP(S, 1);\n  constexpr int TAB_F")
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
    else:
        s = must(s, "    RewCtx rc{", "    RL_STAMP(S, 0);\n    RewCtx rc{")
        s = must(s, "    float total = 0.f;\n    float* rstage = ctx.rew_stage();", "    RL_STAMP(S, 1);\n    float total = 0.f;\n    float* rstage = ctx.rew_stage();")
        s = must(s, "        if (li == 0) rstage[t] = val;\n      }\n    } else {", "        if (li == 0) rstage[t] = val;\n        if (t < 13) RL_STAMP(S, 2 + t);\n      }\n    } else {")
    return s


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
