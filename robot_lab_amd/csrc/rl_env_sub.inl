// rl_env_sub.inl - launcher of the env kernels of ONE lane mapping (RL_ENV_TU_SUB = 1: a lane per limb, 2: a lane pair per limb, 8: eight lanes per limb):
// compiled as a translation unit of its own (rl_env_sub1.hip / rl_env_sub2.hip / rl_env_sub8.hip) so that hipcc works on the mappings in parallel,
// or included by rl_env.hip under -DRL_ENV_SINGLE_TU.  Returns a hipError_t, or -2 when the build does not carry the instance.
#ifndef RL_ENV_ONLY
#define RL_ENV_ONLY 0
#endif
#define RL_SUB_CAT2(a, b) a##b
#define RL_SUB_CAT(a, b) RL_SUB_CAT2(a, b)
extern "C" __attribute__((visibility("hidden"))) int RL_SUB_CAT(rl_env_launch_sub, RL_ENV_TU_SUB)(const void* cfgv, const void* Sv, const void* T, int inst, size_t lds1, void* stream) {
  using namespace rl;
  const LaunchCfg& cfg = *static_cast<const LaunchCfg*>(cfgv);
  const KState& S = *static_cast<const KState*>(Sv);
  hipStream_t st = (hipStream_t)stream;
  constexpr int SUB = RL_ENV_TU_SUB;
  switch (inst) {
#if RL_ENV_TU_SUB == 8  // eight sub-lanes per limb: the trunk + limbs instances only (a 7-joint limb has exactly eight link groups)
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 78
    case 7: return (int)launch_cl<TopoG1, SUB>(cfg, S, T, lds1, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 2078
    case 207: return (int)launch_cl<TopoGR, SUB>(cfg, S, T, lds1, st);
#endif
    default: return -2;
  }
}
#else
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 30 + RL_ENV_TU_SUB
    case 3: return (int)launch_cl<TopoQuad3, SUB>(cfg, S, T, lds1, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 40 + RL_ENV_TU_SUB
    case 4: return (int)launch_cl<TopoQuad4, SUB>(cfg, S, T, lds1, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 1040 + RL_ENV_TU_SUB
    case 104: return (int)launch_cl<TopoQuad4M, SUB>(cfg, S, T, lds1, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 4040 + RL_ENV_TU_SUB
    case 404: return (int)launch_cl<TopoQuad4R, SUB>(cfg, S, T, lds1, st);
#endif
    default: return -2;
  }
}
#endif
