// rl_env_spec.inl - the launcher of the step kernels SPECIALISED on one task (env_spec.h; rl_env_kernels.h launch_spec) as a
// translation unit of its own per Spec (spec/rl_env_spec_<id>.hip, written by tools/gen_specs.py: #define RL_ENV_TU_SPEC <struct>,
// RL_ENV_TU_SPEC_ID <id>), so that hipcc works on the specialised tasks in parallel.
#define RL_SPEC_CAT2(a, b) a##b
#define RL_SPEC_CAT(a, b) RL_SPEC_CAT2(a, b)
extern "C" __attribute__((visibility("hidden"))) int RL_SPEC_CAT(rl_env_launch_spec, RL_ENV_TU_SPEC_ID)(const void* cfgv, const void* Sv, const void* T, int sub, size_t lds1, void* stream) {
  return launch_spec<rl::RL_ENV_TU_SPEC>(*static_cast<const LaunchCfg*>(cfgv), *static_cast<const rl::KState*>(Sv), T, sub, lds1, (hipStream_t)stream);
}
