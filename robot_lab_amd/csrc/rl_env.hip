// rl_env.hip - gfx950 (MI355X, CDNA4) build of the env-step lane program + the C-ABI of include/rl_env.h.
//
// Launch geometry: a 64-lane wavefront = 4 environments x 16 lanes (a DPP row per env, a DPP quad per limb); Npad/4 wavefronts
// (4096 envs -> 1024 wavefronts -> one on every SIMD of the 256 CUs; RL_ENV_SUB=1 selects the older 16 envs x 4 lanes mapping),
// launched four to a workgroup when the launch fills the chip (one staged table image per CU), else one to a workgroup (env_kernel
// below).  The wavefronts of a workgroup share only the staged tables.  The path has no dense contraction -> no MFMA; it is latency
// bound at this size, so a wavefront gets a SIMD's whole register file and every cross-lane reduction is a DPP
// move (quad_perm inside a limb, row mirrors across limbs - no LDS round trip).  Observation rows are staged in LDS and written
// back as one linear, 16-byte-vectorised burst per wavefront.  DESIGN.md section 3 has the LDS budget.
#include <dlfcn.h>

#include <mutex>

#include "rl_env_kernels.h"
#include "rl_env_host.h"
#include "rl_env_specgen.h"

// Step kernels specialised on a task at RUN time (robot_lab_amd/jit.py): a plugin is a shared object compiled from this library's own headers
// + the task's generated Spec, registered through rl_env_register_spec_plugin (include/rl_env.h) and tried by rl_env_create after the Specs
// built into the library.  RL_ENV_ABI_STAMP: a digest of the csrc headers, given to hipcc by __graft_entry__.build() and by the plugin
// build alike - a plugin compiled against other headers is refused.
#define RL_ENV_PLUGINS 1
#ifndef RL_ENV_ABI_STAMP
#define RL_ENV_ABI_STAMP "unstamped"
#endif
namespace {
struct SpecPlugin {
  int id = 0;
  int (*matches)(const void* tables) = nullptr;
  int (*launch)(const void* cfg, const void* S, const void* T, int sub, size_t lds1, void* stream) = nullptr;
};
std::vector<SpecPlugin>& spec_plugins() {
  static std::vector<SpecPlugin> v;
  return v;
}
std::mutex& spec_plugins_mutex() {
  static std::mutex m;
  return m;
}
}  // namespace

// launchers of the other two lane mappings (rl_env_sub.inl: their own translation units unless RL_ENV_SINGLE_TU)
extern "C" __attribute__((visibility("hidden"))) int rl_env_launch_sub1(const void* cfg, const void* S, const void* T, int inst, size_t lds1, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl_env_launch_sub2(const void* cfg, const void* S, const void* T, int inst, size_t lds1, void* stream);
extern "C" __attribute__((visibility("hidden"))) int rl_env_launch_sub8(const void* cfg, const void* S, const void* T, int inst, size_t lds1, void* stream);
// ... and of the step kernels specialised on one task (rl_env_spec.inl; one translation unit per Spec under csrc/spec/)
#ifndef RL_ENV_SINGLE_TU
#define RL_SPEC_DECL(NAME, ID) extern "C" __attribute__((visibility("hidden"))) int rl_env_launch_spec##ID(const void* cfg, const void* S, const void* T, int sub, size_t lds1, void* stream);
#else  // one-instance builds: no specialised kernels, or (-DRL_ENV_SPEC_ONLY=<id>) that one Spec's
#ifndef RL_ENV_SPEC_ONLY
#define RL_ENV_SPEC_ONLY 0
#endif
#define RL_SPEC_DECL(NAME, ID)                                                                                                            \
  static int rl_env_launch_spec##ID(const void* cfg, const void* S, const void* T, int sub, size_t lds1, void* stream) {                      \
    if constexpr (ID == RL_ENV_SPEC_ONLY) return launch_spec<rl::NAME>(*static_cast<const LaunchCfg*>(cfg), *static_cast<const rl::KState*>(S), T, sub, lds1, (hipStream_t)stream); \
    else return -2;                                                                                                                       \
  }
#endif
RL_SPEC_LIST(RL_SPEC_DECL)
#undef RL_SPEC_DECL
#ifdef RL_ENV_SINGLE_TU
#define RL_ENV_TU_SUB 1
#include "rl_env_sub.inl"
#undef RL_ENV_TU_SUB
#define RL_ENV_TU_SUB 2
#include "rl_env_sub.inl"
#undef RL_ENV_TU_SUB
#define RL_ENV_TU_SUB 8
#include "rl_env_sub.inl"
#undef RL_ENV_TU_SUB
#endif

namespace {

using namespace rl;

__global__ void export_kernel(KState S, const Tables* __restrict__ T, AosPtrs A) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < S.Npad) export_env(S, *T, A, e);
}
__global__ void cmd_levels_kernel(float* lv, CmdLevelParams P, const uint32_t* step_base, uint32_t step_offset, uint32_t period) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && (*step_base + step_offset) % period == 0u) apply_cmd_levels(lv, P);
}
__global__ void u32_kernel(uint32_t* p, uint32_t v, int add) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *p = add ? *p + v : v;
}
__global__ void commit_kernel(KState S, const Tables* __restrict__ T, AosPtrs A) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < S.Npad) commit_env(S, *T, A, e);
}

struct Backend {
  std::string err;
  const std::string& error() const { return err; }
  int check(hipError_t e) {
    if (e != hipSuccess) {
      err = hipGetErrorString(e);
      return -1;
    }
    return 0;
  }
  LaunchCfg cfg;
  int& device = cfg.device;
  int& n_cu = cfg.n_cu;
  void read_env() {
    if (const char* v = std::getenv("RL_ENV_WG")) { cfg.wg_waves = atoi(v) == 1 ? 1 : 4; cfg.wg_force = atoi(v) < 0; }  // -4: four-wavefront workgroups whatever the launch size (tests)
  }
  int init(int dev) {
    device = dev;
    if (check(hipSetDevice(dev))) return -1;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    read_env();
    return 0;
  }
  // include/rl_env.h rl_env_plan: the launch geometry create() + launch() arrive at, without a device
  int plan(const Tables& T, int Npad, int n_cu_in, int32_t out[4]) {
    n_cu = n_cu_in > 0 ? n_cu_in : 256;
    read_env();
    const int ept = envs_per_wave(T, Npad);
    if (configure(T)) return -1;
    const size_t tb = staged_bytes(T), lds4 = tb + 4 * (lds_bytes - tb);
    const int tiles = Npad / ept;
    const bool four = (T.NW == 0 || sub == 8) && cfg.wg_waves == 4 && (tiles >= 4 * n_cu || cfg.wg_force) && lds4 <= 160 * 1024;  // launch_cl
    out[0] = sub; out[1] = four ? 4 : 1; out[2] = (int32_t)lds_bytes; out[3] = (int32_t)(four ? lds4 : lds_bytes);
    return 0;
  }
  // every entry point runs on the env's device, whatever the calling thread's current device is
  int activate() { return check(hipSetDevice(device)); }
  void* alloc(size_t n) {
    void* p = nullptr;
    if (check(hipMalloc(&p, n ? n : 16))) return nullptr;
    return p;
  }
  void free(void* p) { (void)hipFree(p); }
  void zero(void* p, size_t n) { check(hipMemset(p, 0, n)); }
  void h2d(void* d, const void* s, size_t n) { check(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); }
  void h2d_stream(void* d, const void* s, size_t n, void* stream) {
    check(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)stream));
    check(hipStreamSynchronize((hipStream_t)stream));  // the host staging vector dies on return
  }
  // Lanes per limb - three mappings of one source, picked from the launch size:
  //   4 (a DPP quad per limb, 16 lanes per env, 4 envs per wavefront): the latency mapping - 4096 envs put one wavefront on every SIMD
  //     and each lane's instruction stream is as short as it gets; but the limb recursion is replicated over the sub-lanes;
  //   2 (a lane pair per limb, 8 envs per wavefront): 1.15x the instructions per wavefront for twice the envs;
  //   1 (a lane per limb, 16 envs per wavefront): the throughput mapping, ~2x the instructions for 4x the envs.
  // A wavefront owns its SIMD (300 - 500 registers), so every mapping runs in ROUNDS of `slots` wavefronts (4 per CU when the LDS
  // allows it - four wavefronts of a workgroup share one staged table image), and a round costs 1 : 1.45 : 2.29 (42.6 / 61.6 / 97.6
  // us on A1 Rough with the chip full, round 4; round 3: 43.5 / 62.5 / 107; Go2W 49.5 / 69.5 / 132: profiles/r03j_sweep_three_mappings.txt).  The choice is the mapping
  // with the cheapest ceil(wavefronts / slots) x cost; on a near-tie the one with more lanes per env (shorter step latency).
  // A1 Rough: <= 4096 envs 16 lanes (43 us), 4097 - 8192 envs 8 lanes (8192: 62.5 us = 131 M env-steps/s against 95 M), ~8.5 k - 16 k
  // envs one lane per limb (16384: 97.6 us = 168 M), and so on by the same rule.  RL_ENV_SUB=4|2|1 forces a mapping; the trunk + limbs
  // instance has 4 only.
  int sub = 4;
  int envs_per_wave(const Tables& T, int Npad) {
    sub = 4;
    if (T.cur_lin || T.cur_ang) {
      sub = 4;  // command-range curricula: the split step's head kernel is built for the 16-lane mapping only (no shipped cfg has them)
    } else if (const char* v = std::getenv("RL_ENV_SUB")) {
      sub = atoi(v) == 1 ? 1 : (atoi(v) == 2 && T.NW == 0 ? 2 : (atoi(v) == 8 && T.NW > 0 && T.sub8_ok ? 8 : 4));
    } else if (T.NW > 0) {
      // trunk + limbs instances: 16 lanes per env (4 envs per wavefront, ~80 KB of LDS: two wavefronts per CU) or 32 (2 envs per
      // wavefront, ~32 KB: four per CU).  Rounds x the cost of a round, as below; RL_ENV_COST8 = cost of a 32-lane round relative to a
      // 16-lane one (measured: profiles/r04*_g1_sub8*.txt)
      sub = 4;
      if (T.sub8_ok) {
        const size_t n4 = T.NW > 3 ? lds_need<TopoGR, 4>(T) : lds_need<TopoG1, 4>(T), n8 = T.NW > 3 ? lds_need<TopoGR, 8>(T) : lds_need<TopoG1, 8>(T);
        const size_t tb = staged_bytes(T);
        const double slots4 = (double)n_cu * (double)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / n4));
        const double slots8 = (double)n_cu * (double)((tb + 4 * (n8 - tb) <= 160 * 1024) ? 4 : std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / n8)));
        double cost8 = 0.93;  // G1 Rough, one call: 143 us per round of 512 wavefronts (2048 envs) against 125 - 135 us per round of 1024 (2048 envs): profiles/r04b_g1_sweep.txt
        if (const char* c = std::getenv("RL_ENV_COST8")) cost8 = atof(c);
        const double t4 = std::ceil((double)(Npad / 4) / slots4), t8 = cost8 * std::ceil((double)(Npad / 2) / slots8);
        if (t8 < t4) sub = 8;
        if (std::getenv("RL_ENV_DEBUG"))
          fprintf(stderr, "rl_env: %d envs: rounds x cost of 16 / 32 lanes per env = %.2f / %.2f (LDS per wavefront %zu / %zu B, tables %zu B) -> %d lanes per limb\n", Npad, t4, t8, n4, n8, tb, sub);
      }
    } else if (T.NW == 0) {
      const size_t tb = staged_bytes(T);
      size_t need[3] = {0, 0, 0};  // LDS of a single-wavefront workgroup, mappings 4 / 2 / 1
      switch (T.CL + (T.merged ? 100 : 0) + (T.rotpad ? 400 : 0)) {
        case 404: need[0] = lds_need<TopoQuad4R, 4>(T); need[1] = lds_need<TopoQuad4R, 2>(T); need[2] = lds_need<TopoQuad4R, 1>(T); break;
        case 3: need[0] = lds_need<TopoQuad3, 4>(T); need[1] = lds_need<TopoQuad3, 2>(T); need[2] = lds_need<TopoQuad3, 1>(T); break;
        case 4: need[0] = lds_need<TopoQuad4, 4>(T); need[1] = lds_need<TopoQuad4, 2>(T); need[2] = lds_need<TopoQuad4, 1>(T); break;
        case 104: need[0] = lds_need<TopoQuad4M, 4>(T); need[1] = lds_need<TopoQuad4M, 2>(T); need[2] = lds_need<TopoQuad4M, 1>(T); break;
        default: break;
      }
      const int subs[3] = {4, 2, 1};
      const double cost[3] = {1.0, 1.45, 2.29};  // (one lane per limb: 97.6 us since its state tiles are addressed as buffers, profiles/r04i_state_buf_ab.txt)
      double best = 0.0, t[3] = {0.0, 0.0, 0.0};
      for (int i = 0; i < 3; ++i) {
        if (need[i] == 0 || need[i] > 160 * 1024) continue;
        // wavefronts a CU holds: single-wavefront workgroups each stage their own table image, a four-wavefront workgroup shares one
        const size_t per_cu = (tb + 4 * (need[i] - tb) <= 160 * 1024) ? 4 : std::min<size_t>(4, (160 * 1024) / need[i]);
        const double slots = (double)n_cu * (double)per_cu, waves = (double)(Npad / (16 / subs[i]));
        t[i] = cost[i] * std::ceil(waves / slots);
        if (best == 0.0 || t[i] < 0.97 * best) { best = t[i]; sub = subs[i]; }
      }
      if (std::getenv("RL_ENV_DEBUG"))
        fprintf(stderr, "rl_env: %d envs: rounds x cost of 4 / 2 / 1 lanes per limb = %.2f / %.2f / %.2f (LDS per wavefront %zu / %zu / %zu B, tables %zu B) -> %d lane(s) per limb\n",
                Npad, t[0], t[1], t[2], need[0], need[1], need[2], tb, sub);
    }
    return 16 / sub;
  }
  size_t lds_bytes = 0;
  // dynamic LDS of a single-wavefront workgroup of an instance: the staged tables + the wavefront's region, by the layout function the
  // kernel itself uses (rl_env_kernels.h lds_plan)
  template <class TP, int SUB>
  static size_t lds_need(const Tables& T) {
    const LdsPlan P = lds_plan<TP, SUB>(T.policy_dim, T.critic_dim, direct_group(T, 0), direct_group(T, 1), T.D, T.n_bodies, T.rew_ext_mask, T.n_rewards);
    return staged_bytes(T) + (size_t)P.words * 4;
  }
  int configure(const Tables& T) {
    const int key = (T.CL + (T.merged ? 100 : 0) + (T.NW > 3 ? 200 : 0) + (T.rotpad ? 400 : 0)) * 10 + sub;
    switch (key) {
      case 31: lds_bytes = lds_need<TopoQuad3, 1>(T); break;
      case 32: lds_bytes = lds_need<TopoQuad3, 2>(T); break;
      case 34: lds_bytes = lds_need<TopoQuad3, 4>(T); break;
      case 41: lds_bytes = lds_need<TopoQuad4, 1>(T); break;
      case 42: lds_bytes = lds_need<TopoQuad4, 2>(T); break;
      case 44: lds_bytes = lds_need<TopoQuad4, 4>(T); break;
      case 1041: lds_bytes = lds_need<TopoQuad4M, 1>(T); break;
      case 1042: lds_bytes = lds_need<TopoQuad4M, 2>(T); break;
      case 1044: lds_bytes = lds_need<TopoQuad4M, 4>(T); break;
      case 4041: lds_bytes = lds_need<TopoQuad4R, 1>(T); break;
      case 4042: lds_bytes = lds_need<TopoQuad4R, 2>(T); break;
      case 4044: lds_bytes = lds_need<TopoQuad4R, 4>(T); break;
      case 71:  // 64 limbs per wavefront: 115 KB of limb-shared words + 30 KB of sensor rows.  The CPU lane emulator runs it (tests/emu); no kernel is built for it
        err = "the one-lane-per-limb mapping (RL_ENV_SUB=1) of the trunk + limbs instance needs more LDS than a CU has";
        return -1;
      case 74: lds_bytes = lds_need<TopoG1, 4>(T); break;
      case 78: lds_bytes = lds_need<TopoG1, 8>(T); break;
      case 2078: lds_bytes = lds_need<TopoGR, 8>(T); break;
      case 2071: err = "the one-lane-per-limb mapping (RL_ENV_SUB=1) of the trunk + limbs instance needs more LDS than a CU has"; return -1;
      case 2074: lds_bytes = lds_need<TopoGR, 4>(T); break;
      default: err = "no lane-program instance for chain length " + std::to_string(T.CL); return -1;
    }
    if (std::getenv("RL_ENV_DEBUG")) fprintf(stderr, "rl_env: %zu B of LDS per single-wavefront workgroup (%zu fit a CU)\n", lds_bytes, (size_t)(160 * 1024) / lds_bytes);
    if (lds_bytes > 160 * 1024) {
      err = "observation rows do not fit the 160 KiB LDS of a CU";
      return -1;
    }
    return 0;
  }
  int spec_id = 0;  // env_spec.h: the Spec whose constants equal this env's tables (rl_env_host.h create), 0: the interpreter
  int (*plugin_launch)(const void*, const void*, const void*, int, size_t, void*) = nullptr;  // spec_id >= 1000: the run-time compiled kernels
  int match_plugin(const void* tables) {  // a registered plugin whose constants equal these tables (rl_env_register_spec_plugin)
    std::lock_guard<std::mutex> lk(spec_plugins_mutex());
    for (const SpecPlugin& p : spec_plugins())
      if (p.matches(tables)) {
        plugin_launch = p.launch;
        return p.id;
      }
    return 0;
  }
  int launch(const KState& S, const void* T, int CL, void* stream) {  // CL: chain length, + 100 for a merged instance, + 400 for the rot / pad quadruped; S.mode: what to run
    hipStream_t st = (hipStream_t)stream;
    if (spec_id >= 1000 && plugin_launch && S.mode == KMODE_STEP) {  // (a plugin built for another lane mapping answers -2: the interpreter below)
      const int rc = plugin_launch(&cfg, &S, T, sub, lds_bytes, stream);
      if (rc != -2) return check((hipError_t)rc);
    }
    if (spec_id != 0 && S.mode == KMODE_STEP) {  // the step kernel specialised on this task, when the build has it for the lane mapping
      int rc = -2;
      switch (spec_id) {
#define RL_SPEC_CASE(NAME, ID) case ID: rc = rl_env_launch_spec##ID(&cfg, &S, T, sub, lds_bytes, stream); break;
        RL_SPEC_LIST(RL_SPEC_CASE)
#undef RL_SPEC_CASE
        default: break;
      }
      if (rc != -2) return check((hipError_t)rc);
    }
    // RL_ENV_ONLY=<CL * 10 + SUB> (e.g. 34; 1044: merged): build that one instance only - kernel experiments compile in 15 s instead of 80
#ifndef RL_ENV_ONLY
#define RL_ENV_ONLY 0
#endif
    const std::string missing = "this build does not carry the lane-program instance for chain length " + std::to_string(CL) + " / " + std::to_string(sub) + " lanes per limb";
    if (sub != 4) {  // the 8- and 4-lane mappings: kernels of their own translation units (rl_env_sub.inl)
      const int rc = sub == 8 ? rl_env_launch_sub8(&cfg, &S, T, CL, lds_bytes, stream)
                              : (sub == 2 ? rl_env_launch_sub2(&cfg, &S, T, CL, lds_bytes, stream) : rl_env_launch_sub1(&cfg, &S, T, CL, lds_bytes, stream));
      if (rc == -2) { err = missing; return -1; }
      return check((hipError_t)rc);
    }
    switch (CL) {
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 34
      case 3: return check(launch_cl<TopoQuad3, 4>(cfg, S, T, lds_bytes, st));
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 44
      case 4: return check(launch_cl<TopoQuad4, 4>(cfg, S, T, lds_bytes, st));
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 1044
      case 104: return check(launch_cl<TopoQuad4M, 4>(cfg, S, T, lds_bytes, st));
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 4044
      case 404: return check(launch_cl<TopoQuad4R, 4>(cfg, S, T, lds_bytes, st));
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 74
      case 7: return check(launch_cl<TopoG1, 4>(cfg, S, T, lds_bytes, st));
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 2074
      case 207: return check(launch_cl<TopoGR, 4>(cfg, S, T, lds_bytes, st));
#endif
      default: err = missing; return -1;
    }
  }
  int launch_export(const KState& S, const Tables* T, const AosPtrs& A, void* stream) {
    hipLaunchKernelGGL(export_kernel, dim3((S.Npad + 63) / 64), dim3(64), 0, (hipStream_t)stream, S, T, A);
    return check(hipGetLastError());
  }
  int launch_cmd_levels(float* lv, const CmdLevelParams& P, const uint32_t* step_base, uint32_t step_offset, uint32_t period, void* stream) {
    hipLaunchKernelGGL(cmd_levels_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, lv, P, step_base, step_offset, period);
    return check(hipGetLastError());
  }
  int launch_u32(uint32_t* p, uint32_t v, int add, void* stream) {  // *p = v / *p += v, stream-ordered (capturable)
    hipLaunchKernelGGL(u32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, v, add);
    return check(hipGetLastError());
  }
  int launch_commit(const KState& S, const Tables* T, const AosPtrs& A, void* stream) {
    hipLaunchKernelGGL(commit_kernel, dim3((S.Npad + 63) / 64), dim3(64), 0, (hipStream_t)stream, S, T, A);
    return check(hipGetLastError());
  }
  int d2h_sync(void* out, const void* src, size_t n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (check(hipMemcpyAsync(out, src, n, hipMemcpyDeviceToHost, st))) return -1;
    return check(hipStreamSynchronize(st));
  }
};

}  // namespace

#include "rl_env_capi.inl"

