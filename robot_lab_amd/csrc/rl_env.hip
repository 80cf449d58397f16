// rl_env.hip - gfx950 (MI355X, CDNA4) build of the env-step lane program + the C-ABI of include/rl_env.h.
//
// Launch geometry: a 64-lane wavefront = 4 environments x 16 lanes (a DPP row per env, a DPP quad per limb); Npad/4 wavefronts
// (4096 envs -> 1024 wavefronts -> one on every SIMD of the 256 CUs; RL_ENV_SUB=1 selects the older 16 envs x 4 lanes mapping),
// launched four to a workgroup when the launch fills the chip (one staged table image per CU), else one to a workgroup (env_kernel
// below).  The wavefronts of a workgroup share only the staged tables.  The path has no dense contraction -> no MFMA; it is latency
// bound at this size, so a wavefront gets a SIMD's whole register file and every cross-lane reduction is a DPP
// move (quad_perm inside a limb, row mirrors across limbs - no LDS round trip).  Observation rows are staged in LDS and written
// back as one linear, 16-byte-vectorised burst per wavefront.  DESIGN.md section 3 has the LDS budget.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>

#define RL_FN __host__ __device__ __forceinline__
#include "env_aos.h"
#include "env_terms.h"
#include "rl_env_host.h"

namespace {

using namespace rl;

template <int CTRL>
__device__ inline float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <-> 15 - i of a 16-lane row
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7 - i of each half row

// Wavefront context.  SUB_ = 1: a lane per leg, 4 lanes per env (a DPP quad), 16 envs per wavefront.
// SUB_ = 4: a DPP quad per leg, 16 lanes per env (a DPP row), 4 envs per wavefront -> 4096 envs fill
// 1024 wavefronts = one per SIMD of the chip, and each lane's instruction stream is ~half as long.
template <int SUB_>
struct WaveCtx {
  static constexpr int LS_STRIDE = 64;
  static constexpr int SUB = SUB_;
  static constexpr int LB_STRIDE = 64 / SUB_;  // limb-shared words: one per limb of the wavefront
  static constexpr bool LIMB_ATOMICS = true;   // limb-shared words are real shared LDS: sub-lanes can ds_add into them
  __device__ static void limb_atomic_add(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  static constexpr int LPE = NLANE * SUB_;
  static constexpr int EPT = 64 / LPE;
  float* lscratch;
  float* lbscratch;
  float* envs;  // this env's words shared by all its lanes (trunk + limbs instance: per-trunk-link accumulators)
  const void* T;  // TablesT<TP> staged in LDS
  float* stage[2];
  float* rstage;
  float* fstage;  // feature vectors of the tile's envs (observations)
  int dim[2], fdim, rtdim;  // rtdim: words of an env's reward tables (they share LDS with the observation rows + features)
  int lane;
  __device__ float* lane_scratch() const { return lscratch + lane; }
  __device__ float* limb_scratch() const { return lbscratch + lane / SUB; }
  __device__ float* env_scratch() const { return envs; }
  __device__ float uniform(float v) const { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
  __device__ int uniform_i(int v) const { return __builtin_amdgcn_readfirstlane(v); }
  __device__ bool any(bool c) const { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
  template <class TT>
  __device__ const TT& tables() const { return *static_cast<const TT*>(T); }
  __device__ int env_in_tile() const { return lane / LPE; }
  __device__ int k() const { return (lane / SUB) & 3; }
  __device__ int sub() const { return lane & (SUB - 1); }
  int wtile;  // this wavefront's tile (= blockIdx.x with one wavefront per workgroup)
  __device__ int tile() const { return wtile; }
  __device__ int env() const { return wtile * EPT + env_in_tile(); }
  // sum over the 4 legs (inputs replicated over a leg's sub-lanes when SUB == 4: the mirrors then pair
  // lanes of different legs, and a + b == b + a bitwise, so all 16 lanes end with identical bits)
  // (SUB == 2: a limb is a lane pair, an env a half row of 8 lanes - the quad xor-2 pairs limbs 0 / 1 and 2 / 3, the half mirror
  // i <-> 7 - i then pairs those sums across the quads)
  __device__ float gsum(float v) const {
    if (SUB == 1) {
      v += dpp<DPP_QUAD_XOR1>(v);
      v += dpp<DPP_QUAD_XOR2>(v);
    } else if (SUB == 2) {
      v += dpp<DPP_QUAD_XOR2>(v);
      v += dpp<DPP_ROW_HALF_MIRROR>(v);
    } else {
      v += dpp<DPP_ROW_HALF_MIRROR>(v);
      v += dpp<DPP_ROW_MIRROR>(v);
    }
    return v;
  }
  __device__ float leg_sum(float v) const {
    if (SUB == 1) return v;
    v += dpp<DPP_QUAD_XOR1>(v);
    if (SUB == 4) v += dpp<DPP_QUAD_XOR2>(v);
    return v;
  }
  __device__ float esum(float v) const { return gsum(leg_sum(v)); }
  // min over the lanes of the env
  __device__ float emin(float v) const {
    v = fminf(v, dpp<DPP_QUAD_XOR1>(v));
    v = fminf(v, dpp<DPP_QUAD_XOR2>(v));
    if (SUB > 1) v = fminf(v, dpp<DPP_ROW_HALF_MIRROR>(v));
    if (SUB > 2) v = fminf(v, dpp<DPP_ROW_MIRROR>(v));
    return v;
  }
  // value held by sub-lane J of this lane's leg (DPP quad_perm broadcast; SUB == 4)
  template <int J>
  __device__ float leg_bcast(float v) const {
    if constexpr (SUB == 1) return v;  // a lane is the whole leg
    else if constexpr (SUB == 2) return dpp<J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6)>(v);  // lane pairs: [J, J, 2 + J, 2 + J]
    else return dpp<J | (J << 2) | (J << 4) | (J << 6)>(v);
  }
  __device__ float gshfl(float v, int leg) const { return __shfl(v, (lane & ~(LPE - 1)) | (leg * SUB) | (lane & (SUB - 1))); }
  __device__ void atomic_add(float* p, float v) const { atomicAdd(p, v); }
  __device__ float* obs_stage(int g) const { return stage[g] + env_in_tile() * dim[g]; }
  __device__ float* rew_stage() const { return rstage + env_in_tile() * MAX_T; }
  __device__ float* feat_stage() const { return fstage + env_in_tile() * fdim; }
  __device__ float* rew_tab() const { return stage[0] + env_in_tile() * rtdim; }
  // Ordering point for LDS traffic between the lanes of the WAVEFRONT (its LDS region is its own, whatever the workgroup width - only
  // the table image is shared, and that is read-only after the staging barrier).  A wavefront's LDS
  // instructions execute in issue order, so a later ds_read of any lane sees an earlier ds_write of any lane without a hardware
  // barrier: all that is needed is that the COMPILER keeps the accesses on their side of this point.  __syncthreads() would add
  // s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier - draining every global load in flight (the terrain and height-scan gathers that
  // are deliberately issued early) ~20 times per step.
  __device__ static void wave_sync() {
#ifdef RL_SYNCTHREADS  // the old form, for A/B runs
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
  }
  __device__ void group_sync() const { wave_sync(); }
  __device__ void flush_obs(float* out, int d, int g) const {
    wave_sync();  // orders the LDS writes above before the reads below
    const int n4 = (EPT * d) >> 2;  // the tile's rows are contiguous in `out`; 16-byte aligned when EPT * d % 4 == 0
    if (((EPT * d) & 3) == 0) {
      const float4* src = reinterpret_cast<const float4*>(stage[g]);
      float4* dst = reinterpret_cast<float4*>(out + (size_t)wtile * EPT * d);
      for (int i = lane; i < n4; i += 64) dst[i] = src[i];
    } else {
      float* dst = out + (size_t)wtile * EPT * d;
      for (int i = lane; i < EPT * d; i += 64) dst[i] = stage[g][i];
    }
    wave_sync();
  }
};

extern __shared__ float4 smem4[];

// WGW wavefronts per workgroup share ONE staged table image in LDS; apart from that staging (and its one s_barrier) the wavefronts
// of a workgroup have nothing to do with each other: each has its own scratch region behind the tables (`wave_words` LDS words) and
// orders its LDS traffic with wave_sync().  4 when the launch has at least 4 wavefronts for every CU (a CU then holds ONE workgroup
// = one wavefront per SIMD, as with single-wavefront workgroups, but stages the tables once instead of four times and the dispatcher
// places a quarter of the workgroups): A1 Rough 4096 52.1 -> 50.2 us.  Smaller launches keep single-wavefront workgroups, which
// spread over more CUs (1024 envs: 49.5 us on 256 CUs, 55.9 us packed four to a CU - profiles/r02_wg_waves.txt).
// __launch_bounds__(256) also for the single-wavefront variant: a leftover of round 2, when declaring 64 threads "caused" a miscompile
// (the push event fired in every env).  The cause is known since round 3 and has nothing to do with the declaration (DESIGN.md
// section 3, profiles/r03d_pin_desc_miscompile.txt): LLVM drops the EXEC restore of an inner divergent region that ends where the
// enclosing one ends, the register allocator then places a live-range-split reload (v_accvgpr_read vX, aY) into the empty flow block,
// and it executes under the inner region's - or an empty - EXEC mask while the outer region's lanes hold a temporary in vX.  Which
// value is hit is a matter of register pressure.  This library is therefore built with -mllvm -amdgpu-remove-redundant-endcf=false,
// and __graft_entry__.build() refuses a build in whose assembly tools/isa_exec_hazard.py finds a vector write under a stale EXEC.
#ifndef RL_LB
#define RL_LB(w) 256
#endif
template <class TP, int RESET, int SUB, int WGW>
__global__ __launch_bounds__(RL_LB(WGW)) void env_kernel(KState S, const void* __restrict__ Tgv, uint32_t wave_words) {
  using Ctx = WaveCtx<SUB>;
  using Tables = TablesT<TP>;
  const Tables* __restrict__ Tg = static_cast<const Tables*>(Tgv);
  float* smem = reinterpret_cast<float*>(smem4);
  Tables* Tl = reinterpret_cast<Tables*>(smem);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  S.step_counter += *S.step_base;  // scalar load: the launch carries the offset from the device-side anchor (rl_env_graph_*)
  {  // stage the used part of the table image into LDS (16-byte vectors): all loads in flight before the first LDS write
    const float4* src = reinterpret_cast<const float4*>(Tg);
    float4* dst = reinterpret_cast<float4*>(Tl);
    constexpr int TPB = 64 * WGW;
    constexpr int NIT = ((int)(sizeof(Tables) / 16) + TPB - 1) / TPB;
    const int n4 = (int)(S.table_bytes >> 4);
    float4 tmp[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = (int)threadIdx.x + TPB * it;
      tmp[it] = src[i < n4 ? i : n4 - 1];  // unconditional (clamped) loads keep tmp[] in registers
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = (int)threadIdx.x + TPB * it;
      if (i < n4) dst[i] = tmp[it];
    }
  }
  if (WGW > 1) __syncthreads();
  else Ctx::wave_sync();
  // LDS after the tables (only the staged bytes take room: the unused tail of the reward table is never touched):
  //   lane scratchpad | limb-shared words | observation staging rows | reward stage
  // On the instances with a contact stash (quadrupeds, 16 lanes per env) the staging rows and the reward stage live ON the
  // stash words of the scratchpad when they fit: the stash is dead once the substeps are over and nothing before them
  // touches the rows.  26 KB -> 20 KB per workgroup = 8 instead of 6 workgroups per CU.
  const int TAB_F = (int)(S.table_bytes >> 2);
  using LS = typename LsFor<TP, SUB>::type;
  constexpr int STASH_WORDS = LsFor<TP, SUB>::STASH * CONTACT_WORDS * 64;
  Ctx ctx;
  ctx.T = Tl;
  ctx.dim[0] = Tl->policy_dim;
  ctx.dim[1] = Tl->critic_dim;
  // (one lane per limb: a group without noise has no staging row - env_terms.h write_group<DIRECT>)
  const int s0w = (SUB == 1 && direct_group(*Tl, 0)) ? 0 : (Ctx::EPT * ctx.dim[0] + 3) & ~3;
  int s1w = (SUB == 1 && direct_group(*Tl, 1)) ? 0 : (Ctx::EPT * ctx.dim[1] + 3) & ~3;
  {  // the staging rows double as limb-shared scratch inside the substeps (streaming CRBA): at least that big
    const int need = LbLayout<TP>::AUX_WORDS * Ctx::LB_STRIDE;
    if (s0w + s1w < need) s1w = need - s0w;
  }
  ctx.lscratch = smem + TAB_F + (WGW > 1 ? wv * wave_words : 0u);  // wave_words: LDS words of one wavefront behind the shared tables
  ctx.wtile = (int)blockIdx.x * WGW + wv;
  if (WGW > 1 && ctx.wtile >= S.Npad / Ctx::EPT) return;
  ctx.lbscratch = ctx.lscratch + LS::WORDS * 64;
  ctx.envs = ctx.lbscratch + LbLayout<TP>::WORDS * Ctx::LB_STRIDE + (lane / Ctx::LPE) * LbLayout<TP>::ENV_WORDS;
  float* tail = ctx.lbscratch + LbLayout<TP>::WORDS * Ctx::LB_STRIDE + Ctx::EPT * LbLayout<TP>::ENV_WORDS;
  // (not on the trunk + limbs instance: its staging rows double as limb-shared scratch during the substeps, when the stash is live)
  ctx.fdim = feat_count(Tl->D);
  ctx.rtdim = rew_tab_words(Tl->D, Tl->n_bodies, Tl->rew_ext_mask);
  // reward stage | [ observation rows + feature vectors ] = [ reward tables ] (the tables die before the rows are written)
  int region = s0w + s1w + Ctx::EPT * ctx.fdim;
  if (region < Ctx::EPT * ctx.rtdim) s1w += Ctx::EPT * ctx.rtdim - region, region = Ctx::EPT * ctx.rtdim;
  // Where they live (same rule as Backend::configure): quadrupeds - on the contact stash of the lane scratchpad, dead once the
  // substeps are over; trunk + limbs instance - on the link-record / elimination words of the limb-shared area, dead likewise
  // (its kinematics words stay: rewards and the scanner pose recompute the chain into them); else behind everything.
  constexpr int LB_FREE = TP::NW > 0 ? (LbLayout<TP>::WORDS - LbLayout<TP>::REC) * Ctx::LB_STRIDE : 0;
  const bool alias = TP::NW == 0 && STASH_WORDS > 0 && Ctx::EPT * MAX_T + region <= STASH_WORDS;
  const bool alias_lb = TP::NW > 0 && Ctx::EPT * MAX_T + region <= LB_FREE;
  float* base = alias ? ctx.lscratch + LS::CT * 64 : (alias_lb ? ctx.lbscratch + LbLayout<TP>::REC * Ctx::LB_STRIDE : tail);
  ctx.rstage = base;
  ctx.stage[0] = base + Ctx::EPT * MAX_T;
  ctx.stage[1] = ctx.stage[0] + s0w;
  ctx.fstage = ctx.stage[1] + s1w;
  ctx.lane = lane;
  EnvProgram<Ctx, TP> prog(ctx, S);
  if (RESET == 1)
    prog.reset_entry();  // (KMODE_RESET, and KMODE_STEP_TAIL: the second launch of a step split around the command-range decision)
  else if (RESET == 2)
    prog.step_head();    // KMODE_STEP_HEAD: the first launch of such a step
  else
    prog.step();
}

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
  int device = 0, n_cu = 256;
  int init(int dev) {
    device = dev;
    if (check(hipSetDevice(dev))) return -1;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    if (const char* v = std::getenv("RL_ENV_WG")) { wg_waves = atoi(v) == 1 ? 1 : 4; wg_force = atoi(v) < 0; }  // -4: four-wavefront workgroups whatever the launch size (tests)
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
  // allows it - four wavefronts of a workgroup share one staged table image), and a round costs 1 : 1.45 : 2.48 (43.5 / 62.5 / 107
  // us on A1 Rough with the chip full; Go2W 49.5 / 69.5 / 132: profiles/r03j_sweep_three_mappings.txt).  The choice is the mapping
  // with the cheapest ceil(wavefronts / slots) x cost; on a near-tie the one with more lanes per env (shorter step latency).
  // A1 Rough: <= 4096 envs 16 lanes (43 us), 4097 - 8192 envs 8 lanes (8192: 62.5 us = 131 M env-steps/s against 95 M), ~8.5 k - 16 k
  // envs one lane per limb (16384: 107 us = 153 M), and so on by the same rule.  RL_ENV_SUB=4|2|1 forces a mapping; the trunk + limbs
  // instance has 4 only.
  int sub = 4;
  int envs_per_wave(const Tables& T, int Npad) {
    sub = 4;
    if (const char* v = std::getenv("RL_ENV_SUB")) {
      sub = atoi(v) == 1 ? 1 : (atoi(v) == 2 && T.NW == 0 ? 2 : 4);
    } else if (T.NW == 0) {
      const size_t tb = staged_bytes(T);
      size_t need[3] = {0, 0, 0};  // LDS of a single-wavefront workgroup, mappings 4 / 2 / 1
      switch (T.CL + (T.merged ? 100 : 0)) {
        case 3: need[0] = lds_need<TopoQuad3, 4>(T); need[1] = lds_need<TopoQuad3, 2>(T); need[2] = lds_need<TopoQuad3, 1>(T); break;
        case 4: need[0] = lds_need<TopoQuad4, 4>(T); need[1] = lds_need<TopoQuad4, 2>(T); need[2] = lds_need<TopoQuad4, 1>(T); break;
        case 104: need[0] = lds_need<TopoQuad4M, 4>(T); need[1] = lds_need<TopoQuad4M, 2>(T); need[2] = lds_need<TopoQuad4M, 1>(T); break;
        default: break;
      }
      const int subs[3] = {4, 2, 1};
      const double cost[3] = {1.0, 1.45, 2.48};
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
  template <class TP, int SUB, int WGW>
  int launch_w(const KState& S, const void* T, size_t lds1, hipStream_t st) {  // lds1: LDS bytes with one wavefront per workgroup; S.mode picks the kernel
    const int tiles = S.Npad / (16 / SUB);
    dim3 grid((tiles + WGW - 1) / WGW), block(64 * WGW);
    const uint32_t wave_words = (uint32_t)((lds1 - S.table_bytes) >> 2);
    const size_t lds = S.table_bytes + (size_t)WGW * (lds1 - S.table_bytes);
    if (lds > 64 * 1024) {
      // opt in to the large LDS carve-out (160 KB per CU on gfx950).  The attribute belongs to the (kernel, device) pair and
      // must cover the LARGEST request: remember per device what was configured and raise it when an env needs more.
      static size_t configured[64] = {};
      static std::mutex mu;
      std::lock_guard<std::mutex> lock(mu);
      size_t& have = configured[device & 63];
      if (lds > have) {
        if (check(hipFuncSetAttribute(reinterpret_cast<const void*>(&env_kernel<TP, 1, SUB, WGW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))) return -1;
        if (check(hipFuncSetAttribute(reinterpret_cast<const void*>(&env_kernel<TP, 0, SUB, WGW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))) return -1;
        if constexpr (WGW == 1)
          if (check(hipFuncSetAttribute(reinterpret_cast<const void*>(&env_kernel<TP, 2, SUB, WGW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))) return -1;
        have = lds;
      }
    }
    if (S.mode == KMODE_RESET || S.mode == KMODE_STEP_TAIL)
      hipLaunchKernelGGL((env_kernel<TP, 1, SUB, WGW>), grid, block, lds, st, S, T, wave_words);
    else if (S.mode == KMODE_STEP_HEAD) {
      if constexpr (WGW == 1) hipLaunchKernelGGL((env_kernel<TP, 2, SUB, WGW>), grid, block, lds, st, S, T, wave_words);
    } else
      hipLaunchKernelGGL((env_kernel<TP, 0, SUB, WGW>), grid, block, lds, st, S, T, wave_words);
    return check(hipGetLastError());
  }
  int wg_waves = 4;  // RL_ENV_WG=1: single-wavefront workgroups always
  bool wg_force = false;
  template <class TP, int SUB>
  int launch_cl(const KState& S, const void* T, size_t lds1, hipStream_t st) {
    if constexpr (TP::NW == 0) {  // (the trunk + limbs instance gains nothing: 174.4 vs 173.8 us with two wavefronts per workgroup)
      const int tiles = S.Npad / (16 / SUB);
      const size_t lds4 = S.table_bytes + 4 * (lds1 - S.table_bytes);
      // (the two launches of a split step - command-range curricula, no shipped cfg - use the single-wavefront workgroups: half the kernels to build)
      if (wg_waves == 4 && (tiles >= 4 * n_cu || wg_force) && lds4 <= 160 * 1024 && S.mode != KMODE_STEP_HEAD && S.mode != KMODE_STEP_TAIL)
        return launch_w<TP, SUB, 4>(S, T, lds1, st);
    }
    return launch_w<TP, SUB, 1>(S, T, lds1, st);
  }
  size_t lds_bytes = 0;
  // dynamic LDS of an instance: the SAME layout arithmetic as env_kernel (tables | lane scratchpad | limb-shared words | per-env
  // words | staging rows + reward stage unless they alias the contact stash / the record words)
  template <class TP, int SUB>
  static size_t lds_need(const Tables& T) {
    using Ctx = WaveCtx<SUB>;
    using LS = typename LsFor<TP, SUB>::type;
    const int s0w = (SUB == 1 && direct_group(T, 0)) ? 0 : (Ctx::EPT * T.policy_dim + 3) & ~3;
    int s1w = (SUB == 1 && direct_group(T, 1)) ? 0 : (Ctx::EPT * T.critic_dim + 3) & ~3;
    const int need = LbLayout<TP>::AUX_WORDS * Ctx::LB_STRIDE;
    if (s0w + s1w < need) s1w = need - s0w;
    int region = s0w + s1w + Ctx::EPT * feat_count(T.D);
    region = std::max(region, Ctx::EPT * rew_tab_words(T.D, T.n_bodies, T.rew_ext_mask));
    constexpr int STASH_WORDS = LsFor<TP, SUB>::STASH * CONTACT_WORDS * 64;
    constexpr int LB_FREE = TP::NW > 0 ? (LbLayout<TP>::WORDS - LbLayout<TP>::REC) * Ctx::LB_STRIDE : 0;
    const bool alias = TP::NW == 0 && STASH_WORDS > 0 && Ctx::EPT * MAX_T + region <= STASH_WORDS;
    const bool alias_lb = TP::NW > 0 && Ctx::EPT * MAX_T + region <= LB_FREE;
    size_t words = (size_t)LS::WORDS * 64 + (size_t)LbLayout<TP>::WORDS * Ctx::LB_STRIDE + (size_t)Ctx::EPT * LbLayout<TP>::ENV_WORDS;
    if (!alias && !alias_lb) words += (size_t)Ctx::EPT * MAX_T + region;
    return staged_bytes(T) + words * 4;
  }
  int configure(const Tables& T) {
    const int key = (T.CL + (T.merged ? 100 : 0)) * 10 + sub;
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
      case 71:  // 64 limbs per wavefront: 115 KB of limb-shared words + 30 KB of sensor rows.  The CPU lane emulator runs it (tests/emu); no kernel is built for it
        err = "the one-lane-per-limb mapping (RL_ENV_SUB=1) of the trunk + limbs instance needs more LDS than a CU has";
        return -1;
      case 74: lds_bytes = lds_need<TopoG1, 4>(T); break;
      default: err = "no lane-program instance for chain length " + std::to_string(T.CL); return -1;
    }
    if (lds_bytes > 160 * 1024) {
      err = "observation rows do not fit the 160 KiB LDS of a CU";
      return -1;
    }
    return 0;
  }
  int launch(const KState& S, const void* T, int CL, void* stream) {  // CL: chain length, + 100 for a merged instance; S.mode: what to run
    hipStream_t st = (hipStream_t)stream;
    // RL_ENV_ONLY=<CL * 10 + SUB> (e.g. 34; 1044: merged): build that one instance only - kernel experiments compile in 15 s instead of 80
#ifndef RL_ENV_ONLY
#define RL_ENV_ONLY 0
#endif
    switch (CL * 10 + sub) {
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 31
      case 31: return launch_cl<TopoQuad3, 1>(S, T, lds_bytes, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 41
      case 41: return launch_cl<TopoQuad4, 1>(S, T, lds_bytes, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 34
      case 34: return launch_cl<TopoQuad3, 4>(S, T, lds_bytes, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 32
      case 32: return launch_cl<TopoQuad3, 2>(S, T, lds_bytes, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 42
      case 42: return launch_cl<TopoQuad4, 2>(S, T, lds_bytes, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 1042
      case 1042: return launch_cl<TopoQuad4M, 2>(S, T, lds_bytes, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 44
      case 44: return launch_cl<TopoQuad4, 4>(S, T, lds_bytes, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 1041
      case 1041: return launch_cl<TopoQuad4M, 1>(S, T, lds_bytes, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 1044
      case 1044: return launch_cl<TopoQuad4M, 4>(S, T, lds_bytes, st);
#endif
#if RL_ENV_ONLY == 0 || RL_ENV_ONLY == 74
      case 74: return launch_cl<TopoG1, 4>(S, T, lds_bytes, st);
#endif
      default: err = "this build does not carry the lane-program instance for chain length " + std::to_string(CL) + " / " + std::to_string(sub) + " lanes per limb"; return -1;
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

