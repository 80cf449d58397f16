// rl_env_capi.inl - the extern "C" entry points of include/rl_env.h, shared by the HIP library and the
// CPU lane emulator.  The including file defines `Backend` first.
using Impl = rl::EnvImpl<Backend>;

extern "C" {

int rl_env_create(const rl_env_desc* desc, const float* terrain_heights, const float* terrain_origins, const float* env_origins,
                  int32_t num_envs, uint64_t seed, int32_t device, rl_env** out) {
  if (!desc || !out) return rl::fail("null argument");
  Impl* impl = new Impl();
  int rc = impl->create(desc, terrain_heights, terrain_origins, env_origins, num_envs, seed, device);
  if (rc) {
    impl->destroy();
    delete impl;
    return rc;
  }
  *out = reinterpret_cast<rl_env*>(impl);
  return 0;
}

int rl_env_reset(rl_env* env, const int32_t* env_ids, int32_t n, void* stream) {
  if (!env) return rl::fail("null env");
  return reinterpret_cast<Impl*>(env)->reset(env_ids, n, stream);
}

int rl_env_step(rl_env* env, const float* action_dev, void* stream) {
  if (!env) return rl::fail("null env");
  return reinterpret_cast<Impl*>(env)->step(action_dev, stream);
}

int rl_env_step_record(rl_env* env, const float* action_dev, const float* values_dev, float* rewards_out_dev, uint8_t* dones_out_dev, float gamma,
                       void* stream) {
  if (!env) return rl::fail("null env");
  if (!rewards_out_dev || !dones_out_dev) return rl::fail("rollout sink needs rewards and dones");  // (values_dev NULL: deferred bootstrap, include/rl_env.h)
  return reinterpret_cast<Impl*>(env)->step(action_dev, stream, values_dev, rewards_out_dev, dones_out_dev, gamma);
}

int rl_env_get_buffer(rl_env* env, int32_t which, void** dev_ptr, int64_t shape[3], int32_t* ndim, int32_t* elem_size) {
  if (!env || !dev_ptr || !shape || !ndim || !elem_size) return rl::fail("null argument");
  Impl& I = *reinterpret_cast<Impl*>(env);
  const int64_t N = I.N, Np = I.Npad, D = I.D, B = I.B;
  constexpr int FS = (int)sizeof(float);  // element size of the real-valued buffers
  auto set = [&](void* p, int nd, int64_t a, int64_t b, int64_t c, int es) {
    *dev_ptr = p; *ndim = nd; shape[0] = a; shape[1] = b; shape[2] = c; *elem_size = es;
    return 0;
  };
  if ((which == RL_BUF_CONTACT_FORCE || which == RL_BUF_JOINT_TORQUE || which == RL_BUF_JOINT_ACC) && I.enable_inspection()) return -1;
  switch (which) {
    case RL_BUF_OBS_POLICY: return set(I.S.obs_policy, 2, N, I.tables.policy_dim, 1, FS);  // the slot the last step()/reset() wrote
    case RL_BUF_OBS_CRITIC: return set(I.S.obs_critic, 2, N, I.tables.critic_dim, 1, FS);
    case RL_BUF_OBS_POLICY_RING: return set(I.obs_ring[0][0], 3, 2, Np, I.tables.policy_dim, FS);
    case RL_BUF_OBS_CRITIC_RING: return set(I.obs_ring[1][0], 3, 2, Np, I.tables.critic_dim, FS);
    case RL_BUF_TASK_STATE: return set(I.task_state, 2, N, RL_TASK_STATE_NF, 1, FS);
    case RL_BUF_GAINS: return set(I.gains, 3, N, 2, D, FS);
    case RL_BUF_CMD_LEVELS: return set(I.S.cmd_levels, 1, rl::CL_WORDS, 1, 1, FS);
    case RL_BUF_REWARD: return set(I.S.reward, 1, N, 1, 1, FS);
    case RL_BUF_TERMINATED: return set(I.S.terminated, 1, N, 1, 1, 1);
    case RL_BUF_TIME_OUT: return set(I.S.time_out, 1, N, 1, 1, 1);
    case RL_BUF_EPISODE_LENGTH: return set(I.S.ep_len, 1, N, 1, 1, 8);
    case RL_BUF_ROOT_STATE: return set(I.root_state, 2, N, 13, 1, FS);
    case RL_BUF_JOINT_POS: return set(I.joint_pos, 2, N, D, 1, FS);
    case RL_BUF_JOINT_VEL: return set(I.joint_vel, 2, N, D, 1, FS);
    case RL_BUF_REWARD_TERMS: return set(I.S.rew_terms, 2, I.tables.n_rewards, Np, 1, FS);
    case RL_BUF_EPISODE_SUMS: return set(I.S.ep_sums, 2, I.tables.n_rewards, Np, 1, FS);
    case RL_BUF_COMMAND: return set(I.S.command_out, 2, N, 3, 1, FS);
    case RL_BUF_CONTACT_FORCE: return set(I.S.dbg_cforce, 3, N, B, 3, FS);
    case RL_BUF_CONTACT_TIMERS: return set(I.ctimers, 3, N, B, 4, FS);
    case RL_BUF_LOG: return set(I.S.log, 3, RL_LOG_RING, RL_LOG_PARTS, RL_LOG_SIZE, FS);
    case RL_BUF_ACTION: return set(I.action_aos, 2, N, D, 1, FS);
    case RL_BUF_JOINT_TORQUE: return set(I.S.dbg_torque, 2, N, D, 1, FS);
    case RL_BUF_JOINT_ACC: return set(I.S.dbg_acc, 2, N, D, 1, FS);
    case RL_BUF_ENV_ORIGIN: return set(I.env_origin_aos, 2, N, 3, 1, FS);
    case RL_BUF_TERRAIN_LEVEL: return set(I.S.level, 1, N, 1, 1, (int)sizeof(int32_t));
    default: return rl::fail("unknown buffer id");
  }
}

int rl_env_export_state(rl_env* env, void* stream) {
  if (!env) return rl::fail("null env");
  return reinterpret_cast<Impl*>(env)->export_state(stream);
}

int rl_env_commit_state(rl_env* env, void* stream) {
  if (!env) return rl::fail("null env");
  return reinterpret_cast<Impl*>(env)->commit_state(stream);
}

int rl_env_import_state(rl_env* env, const float* root_state, const float* joint_pos, const float* joint_vel, void* stream) {
  if (!env) return rl::fail("null env");
  return reinterpret_cast<Impl*>(env)->import_state(root_state, joint_pos, joint_vel, stream);
}

int rl_env_read_log(rl_env* env, float* out_host, void* stream) {
  if (!env || !out_host) return rl::fail("null argument");
  Impl& I = *reinterpret_cast<Impl*>(env);
  if (I.be.activate()) return rl::fail("device activation failed: " + I.be.error());
  // the last step's slot, or - if that step reset nobody - its predecessor, which the kernel has resolved the same way
  // (a slot is RL_LOG_PARTS partial rows: summed here)
  static_assert(RL_LOG_PARTS == rl::LOG_PARTS && RL_LOG_SIZE == rl::LOG_SIZE && RL_LOG_RING == rl::LOG_RING, "include/rl_env.h and csrc/env_tables.h disagree");
  std::vector<float> rows((size_t)RL_LOG_PARTS * RL_LOG_SIZE);
  float two[2][RL_LOG_SIZE];
  for (int i = 0; i < 2; ++i) {
    float* slot = I.S.log + (size_t)((I.step_counter - (uint32_t)i) & (uint32_t)(RL_LOG_RING - 1)) * RL_LOG_PARTS * RL_LOG_SIZE;
    if (I.be.d2h_sync(rows.data(), slot, rows.size() * sizeof(float), stream)) return rl::fail("log read failed: " + I.be.error());
    for (int w = 0; w < RL_LOG_SIZE; ++w) {
      float a = 0.f;
      for (int p = 0; p < RL_LOG_PARTS; ++p) a += rows[(size_t)p * RL_LOG_SIZE + w];
      two[i][w] = a;
    }
  }
  memcpy(out_host, two[0][0] > 0.f ? two[0] : two[1], RL_LOG_SIZE * sizeof(float));
  return 0;
}

int32_t rl_env_log_slot(const rl_env* env) {
  return env ? (int32_t)(reinterpret_cast<const Impl*>(env)->step_counter & (uint32_t)(RL_LOG_RING - 1)) : -1;
}

int32_t rl_env_obs_slot(const rl_env* env) { return env ? reinterpret_cast<const Impl*>(env)->obs_slot : -1; }
int64_t rl_env_step_count(const rl_env* env) { return env ? (int64_t) reinterpret_cast<const Impl*>(env)->step_counter : -1; }
int rl_env_set_step_count(rl_env* env, int64_t count) {
  if (!env) return rl::fail("null env");
  if (count < 0 || count > 0xffffffffll) return rl::fail("step count out of range");
  return reinterpret_cast<Impl*>(env)->set_step_count((uint32_t)count);
}
int rl_env_graph_begin(rl_env* env, void* stream) { return env ? reinterpret_cast<Impl*>(env)->graph_begin(stream) : rl::fail("null env"); }
int rl_env_graph_end(rl_env* env, void* stream) { return env ? reinterpret_cast<Impl*>(env)->graph_end(stream) : rl::fail("null env"); }
int rl_env_graph_launching(rl_env* env, void* stream) { return env ? reinterpret_cast<Impl*>(env)->graph_launching(stream) : rl::fail("null env"); }

int32_t rl_env_num_envs(const rl_env* env) { return reinterpret_cast<const Impl*>(env)->N; }
int32_t rl_env_num_actions(const rl_env* env) { return reinterpret_cast<const Impl*>(env)->D; }
int32_t rl_env_obs_dim(const rl_env* env, int32_t group) {
  const Impl* I = reinterpret_cast<const Impl*>(env);
  return group == 0 ? I->tables.policy_dim : I->tables.critic_dim;
}
int32_t rl_env_max_episode_length(const rl_env* env) { return reinterpret_cast<const Impl*>(env)->tables.max_episode_length; }
int32_t rl_env_envs_per_wavefront(const rl_env* env) { return reinterpret_cast<const Impl*>(env)->ept; }
int32_t rl_env_spec_id(const rl_env* env) { return env ? reinterpret_cast<const Impl*>(env)->spec_id : -1; }

int rl_env_plan(const rl_env_desc* desc, int32_t num_envs, int32_t n_cu, int32_t out[4]) {
  if (!desc || !out) return rl::fail("null argument");
  if (num_envs <= 0) return rl::fail("num_envs must be positive");
  std::vector<int> bl, bs, ll, lp;
  rl::Tables* T = new rl::Tables();
  int rc = rl::compile_tables(*desc, *T, bl, bs, ll, lp);
  if (rc == 0) {
    Backend be;
    const int Npad = (num_envs + rl::ENVS_PER_WAVE - 1) / rl::ENVS_PER_WAVE * rl::ENVS_PER_WAVE;
    rc = be.plan(*T, Npad, n_cu, out);
    if (rc) rl::fail("kernel configuration failed: " + be.error());
  }
  delete T;
  return rc;
}

int rl_env_destroy(rl_env* env) {
  if (!env) return 0;
  Impl* I = reinterpret_cast<Impl*>(env);
  I->destroy();
  delete I;
  return 0;
}

// The C++ source of the Spec of a task (csrc/rl_env_specgen.h): build-time tooling (tools/gen_specs.py) and the first step of specialising a
// task at run time (robot_lab_amd/jit.py).  Returns the length written (0: the task cannot be specialised - rl_env_last_error says why;
// -1: `cap` too small).  Host code: needs no device.
int rl_env_spec_source(const rl_env_desc* desc, const char* struct_name, const char* task, int id, char* out, int cap) {
  if (!desc || !struct_name || !task || !out) return 0;
  const std::string src = rl::spec_source(*desc, struct_name, task, id);
  if (src.empty()) return 0;
  if ((int)src.size() + 1 > cap) return -1;
  memcpy(out, src.c_str(), src.size() + 1);
  return (int)src.size();
}

#ifdef RL_ENV_PLUGINS
const char* rl_env_abi_stamp(void) { return RL_ENV_ABI_STAMP; }
int32_t rl_env_spec_plugin_count(void) { return (int32_t)spec_plugins().size(); }
int rl_env_register_spec_plugin(const char* so_path) {
  if (!so_path) return rl::fail("null path");
  void* h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return rl::fail(std::string("dlopen failed: ") + dlerror());
  SpecPlugin p;
  auto abi = reinterpret_cast<const char* (*)()>(dlsym(h, "rl_spec_plugin_abi"));
  auto idf = reinterpret_cast<int (*)()>(dlsym(h, "rl_spec_plugin_id"));
  p.matches = reinterpret_cast<int (*)(const void*)>(dlsym(h, "rl_spec_plugin_matches"));
  p.launch = reinterpret_cast<int (*)(const void*, const void*, const void*, int, size_t, void*)>(dlsym(h, "rl_spec_plugin_launch"));
  if (!abi || !idf || !p.matches || !p.launch) { dlclose(h); return rl::fail(std::string(so_path) + " does not export the rl_spec_plugin_* entry points"); }
  if (std::string(abi()) != RL_ENV_ABI_STAMP) {  // compiled against other headers than this library: KState / Tables / LaunchCfg may differ
    const std::string theirs = abi();
    dlclose(h);
    return rl::fail(std::string(so_path) + " was compiled against csrc headers " + theirs + ", this library against " RL_ENV_ABI_STAMP);
  }
  p.id = idf();
  if (p.id < 1000) { dlclose(h); return rl::fail("plugin spec ids start at 1000 (below: the Specs built into the library)"); }
  std::lock_guard<std::mutex> lk(spec_plugins_mutex());
  for (const SpecPlugin& q : spec_plugins())
    if (q.id == p.id) return 0;  // already registered (the handle of the second dlopen is the same object: leave it)
  spec_plugins().push_back(p);
  return 0;
}
#else  // (the CPU lane emulator: plugins are HIP code objects)
const char* rl_env_abi_stamp(void) { return "emulator"; }
int32_t rl_env_spec_plugin_count(void) { return 0; }
int rl_env_register_spec_plugin(const char*) { return rl::fail("the CPU lane emulator loads no step-kernel plugins"); }
#endif

const char* rl_env_last_error(void) { return rl::last_error().c_str(); }
uint64_t rl_env_desc_size(void) { return sizeof(rl_env_desc); }

}  // extern "C"
