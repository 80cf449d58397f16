// rl_env_kernels.h - the gfx950 wavefront context, the env kernel and its launch helpers: what every translation unit of the env
// library shares.  The library is built from one translation unit per lane mapping (rl_env.hip: 16 lanes per env + the C-ABI; rl_env_sub8 /
// _sub2 / _sub1.hip: 32, 8 and 4 lanes per env - 33 interpreter kernels of ~10 s each) plus one per task-specialised step kernel set
// (spec/rl_env_spec_<id>.hip), so that hipcc compiles them side by side; -DRL_ENV_SINGLE_TU puts ONE instance into rl_env.hip
// (tools/build_variant.sh, tools/kbuild.sh).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>

#define RL_FN __host__ __device__ __forceinline__
#include "env_aos.h"
#include "env_terms.h"

namespace {

using namespace rl;

template <int CTRL>
__device__ inline float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
// a pure lane move (every lane has a valid source): bound_ctrl, so that no `old` value has to be materialised in front of the v_mov_dpp
template <int CTRL>
__device__ inline float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <-> 15 - i of a 16-lane row
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7 - i of each half row

// Wavefront context.  SUB_ = 1: a lane per leg, 4 lanes per env (a DPP quad), 16 envs per wavefront.
// SUB_ = 8 (trunk + limbs instances): two DPP quads per limb, 32 lanes per env, 2 envs per wavefront -> 2048 envs fill 1024 wavefronts.
// SUB_ = 4: a DPP quad per leg, 16 lanes per env (a DPP row), 4 envs per wavefront -> 4096 envs fill
// 1024 wavefronts = one per SIMD of the chip, and each lane's instruction stream is ~half as long.
template <int SUB_>
struct WaveCtx {
  static constexpr int LS_STRIDE = 64;
  static constexpr int SUB = SUB_;
  static constexpr int LIMBS = 64 / SUB_;      // limbs of the wavefront's envs: limb-shared LDS blocks (trunk + limbs instance)
  static constexpr bool LIMB_ATOMICS = true;   // limb-shared words are real shared LDS: sub-lanes can ds_add into them
  __device__ static void limb_atomic_add(float* p, float v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  static constexpr int LPE = NLANE * SUB_;
  static constexpr int EPT = 64 / LPE;
  float* lscratch;
  float* lbchain;  // this lane's limb: its kinematics block (LbLayout::CHAINW words; the blocks of the wavefront's limbs follow each other)
  float* lbrec;    // ... and its block of link records / per-joint words (LbLayout::RECW words), behind all the kinematics blocks
  int lbrec_stride;
  float* envs;  // this env's words shared by all its lanes (trunk + limbs instance: per-trunk-link accumulators)
  const void* T;  // TablesT<TP> staged in LDS
  float* stage[2];
  float* rstage;
  float* fstage;  // feature vectors of the tile's envs (observations)
  int dim[2], fdim, rtdim, rsdim;  // rtdim: words of an env's reward tables (they share LDS with the observation rows + features); rsdim: of its reward-stage row
  int lane;
  // this lane's place in the scratchpad: GRAN - its 16-byte piece of granule 0 (granule g: + g * 4 * LS_STRIDE words); else its word of row 0 (word f: + f * LS_STRIDE)
  template <bool GRAN>
  __device__ float* lane_scratch() const { return lscratch + (GRAN ? lane * 4 : lane); }
  __device__ float* limb_chain() const { return lbchain; }
  __device__ float* limb_rec() const { return lbrec; }
  __device__ float* limb_rec_of(int k2) const { return lbrec + (k2 - ((lane / SUB) & 3)) * lbrec_stride; }  // limb k2 of this lane's env
  __device__ float* env_scratch() const { return envs; }
  __device__ float uniform(float v) const { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
  __device__ int uniform_i(int v) const { return __builtin_amdgcn_readfirstlane(v); }
  // a tile of the HBM state as a raw buffer (stride 0, `bytes` long: reads beyond it return 0, writes are dropped)
  struct StateBuf {
    __amdgpu_buffer_rsrc_t rs;
  };
  __device__ static StateBuf state_buf(float* uniform_base, uint32_t bytes) { return StateBuf{__builtin_amdgcn_make_buffer_rsrc(uniform_base, 0, (int)bytes, 0x00020000)}; }
  __device__ static float buf_ld(const StateBuf& b, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.rs, (int)voff, (int)soff, 0));
  }
  __device__ static void buf_st(const StateBuf& b, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), b.rs, (int)voff, (int)soff, 0);
  }
  template <class P>
  __device__ static P* uniform_ptr(P* p) {  // a pointer every lane holds the same value of: into scalar registers
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return reinterpret_cast<P*>(((uint64_t)hi << 32) | lo);
  }
  __device__ bool any(bool c) const { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
  template <class TT>
  __device__ const TT& tables() const { return *static_cast<const TT*>(T); }
  const void* Tg;  // the whole table image in HBM (the LDS copy `T` ends with the used reward descriptors)
  template <class TT>
  __device__ const TT& tables_global() const { return *static_cast<const TT*>(Tg); }
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
  // (SUB == 8: a limb is two DPP quads, an env two DPP rows - 32 lanes, 2 envs per wavefront: the row mirror pairs limbs 0 / 1 and
  // 2 / 3, then the two rows of the env are exchanged with a ds_swizzle in 32-lane bit mode, xor 16: no LDS memory is touched)
  __device__ static float swap_rows(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); }  // and 0x1F, or 0, xor 0x10
  __device__ float gsum(float v) const {
    if (SUB == 8) {
      v += dpp<DPP_ROW_MIRROR>(v);
      v += swap_rows(v);
    } else if (SUB == 1) {
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
    if (SUB >= 4) v += dpp<DPP_QUAD_XOR2>(v);
    if (SUB == 8) v += dpp<DPP_ROW_HALF_MIRROR>(v);  // the limb's other quad (lane i <-> 7 - i of the limb's eight)
    return v;
  }
  __device__ float esum(float v) const { return gsum(leg_sum(v)); }
  // the value held by the same sub-lane position of limb k ^ X of this env (X = 1, 2, 3): one DPP move, or two where the exchange is
  // not a single pattern of the mapping (the specialised joint_mirror term, env_terms.h compute_rewards_spec).  Inputs are replicated
  // over a limb's sub-lanes, so WHICH sub-lane of the partner limb a lane reads does not matter.
  template <int X>
  __device__ float limb_xor(float v) const {
    static_assert(X >= 1 && X <= 3, "limb index xor");
    constexpr int DPP_QUAD_REV = 0x1B;  // quad_perm:[3,2,1,0]
    constexpr int DPP_ROW_ROR8 = 0x128; // row_ror:8
    if constexpr (SUB == 1) return dpp_move<X == 1 ? DPP_QUAD_XOR1 : (X == 2 ? DPP_QUAD_XOR2 : DPP_QUAD_REV)>(v);  // a limb is a lane of the quad
    else if constexpr (SUB == 2) {  // a limb is a lane pair, the env a half row
      if constexpr (X == 1) return dpp_move<DPP_QUAD_XOR2>(v);
      else if constexpr (X == 3) return dpp_move<DPP_ROW_HALF_MIRROR>(v);
      else return dpp_move<DPP_ROW_HALF_MIRROR>(dpp_move<DPP_QUAD_XOR2>(v));
    } else if constexpr (SUB == 4) {  // a limb is a quad, the env a row
      return dpp_move<X == 1 ? DPP_ROW_HALF_MIRROR : (X == 3 ? DPP_ROW_MIRROR : DPP_ROW_ROR8)>(v);
    } else {  // SUB == 8: a limb is half a row, the env two rows
      if constexpr (X == 1) return dpp_move<DPP_ROW_MIRROR>(v);
      else if constexpr (X == 2) return swap_rows(v);
      else return swap_rows(dpp_move<DPP_ROW_MIRROR>(v));
    }
  }
  // min over the lanes of the env
  __device__ float emin(float v) const {
    v = fminf(v, dpp<DPP_QUAD_XOR1>(v));
    v = fminf(v, dpp<DPP_QUAD_XOR2>(v));
    if (SUB > 1) v = fminf(v, dpp<DPP_ROW_HALF_MIRROR>(v));
    if (SUB > 2) v = fminf(v, dpp<DPP_ROW_MIRROR>(v));
    if (SUB > 4) v = fminf(v, swap_rows(v));
    return v;
  }
  // value held by sub-lane J of this lane's leg (DPP quad_perm broadcast; SUB == 4)
  template <int J>
  __device__ float leg_bcast(float v) const {
    if constexpr (SUB == 1) return v;  // a lane is the whole leg
    else if constexpr (SUB == 2) return dpp_move<J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6)>(v);  // lane pairs: [J, J, 2 + J, 2 + J]
    else if constexpr (SUB == 8) {  // sub-lane J & 3 of both quads, then the other quad's copy where J sits there
      constexpr int J4 = J & 3;
      const float t = dpp_move<J4 | (J4 << 2) | (J4 << 4) | (J4 << 6)>(v);
      const float o = dpp_move<DPP_ROW_HALF_MIRROR>(t);
      return ((lane >> 2) & 1) == (J >> 2) ? t : o;
    } else return dpp_move<J | (J << 2) | (J << 4) | (J << 6)>(v);
  }
  // the lane's DPP quad (SUB >= 4: four sub-lanes of one limb): sum over it (identical bits in its four lanes: (x0 + x1) + (x2 + x3) either
  // way round) and the value of its lane J
  __device__ static float quad_sum(float v) {
    v += dpp<DPP_QUAD_XOR1>(v);
    v += dpp<DPP_QUAD_XOR2>(v);
    return v;
  }
  template <int J>
  __device__ static float quad_bcast(float v) { return dpp_move<J | (J << 2) | (J << 4) | (J << 6)>(v); }
  // the dealing of per-joint work (chain_kinematics_dealt) is among the four lanes of a DPP quad in every mapping with SUB >= 4
  template <int J>
  __device__ __forceinline__ M3 deal_bcast_m3(const M3& m) const {
    if constexpr (SUB == 8) {
      auto b = [](float v) { return dpp_move<J | (J << 2) | (J << 4) | (J << 6)>(v); };
      return M3{{b(m.r0.x), b(m.r0.y), b(m.r0.z)}, {b(m.r1.x), b(m.r1.y), b(m.r1.z)}, {b(m.r2.x), b(m.r2.y), b(m.r2.z)}};
    } else return leg_bcast_m3<J>(m);
  }
  // a 3 x 3 matrix held by sub-lane J of this lane's leg (nine quad broadcasts)
  template <int J>
  __device__ __forceinline__ M3 leg_bcast_m3(const M3& m) const {
    return M3{{leg_bcast<J>(m.r0.x), leg_bcast<J>(m.r0.y), leg_bcast<J>(m.r0.z)},
              {leg_bcast<J>(m.r1.x), leg_bcast<J>(m.r1.y), leg_bcast<J>(m.r1.z)},
              {leg_bcast<J>(m.r2.x), leg_bcast<J>(m.r2.y), leg_bcast<J>(m.r2.z)}};
  }
  // w[] of sub-lane (sub - D) of this lane's limb, for the sub-lanes that have one (sub >= D; the others get something: their callers select).
  // Eight sub-lanes per limb: a limb is half a DPP row, row_shr:D stays inside it for those lanes.  (kinematics_scan, env_step.h)
  template <int D, int N>
  __device__ __forceinline__ void sub_shr(float (&w)[N]) const {
    static_assert(SUB == 8 && D >= 1 && D < 8, "eight sub-lanes per limb");
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = dpp_move<0x110 + D>(w[i]);  // row_shr:D
  }
  __device__ float gshfl(float v, int leg) const { return __shfl(v, (lane & ~(LPE - 1)) | (leg * SUB) | (lane & (SUB - 1))); }
  __device__ void atomic_add(float* p, float v) const { atomicAdd(p, v); }
  __device__ float* obs_stage(int g) const { return stage[g] + env_in_tile() * dim[g]; }
  __device__ float* rew_stage() const { return rstage + env_in_tile() * rsdim; }
  __device__ float* feat_stage() const { return fstage + env_in_tile() * fdim; }
  __device__ float* rew_tab() const { return stage[0] + env_in_tile() * rtdim; }
  __device__ float* rand_tab() const { return stage[0] + env_in_tile() * RESET_RAND_WORDS; }  // reset uniforms: between the reward tables' death and the observation rows' birth
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

// LDS of ONE wavefront behind the staged table image, in words from the wavefront's base - the one place that decides it: env_kernel lays
// its pointers out with it and the host sizes the launch with it (rl_env.hip lds_need), so the two cannot drift apart.
//   lane scratchpad | limb-shared words (kinematics blocks, record blocks) | per-env words | [ reward stage | observation rows + feature vectors ]
// The bracketed part shares its words with the reward tables and the reset uniforms (both dead before the rows are written) and lives ON
// words that are dead after the substeps where it fits: the contact stash of the lane scratchpad (quadrupeds), the record blocks of the
// limb-shared area (trunk + limbs instances; their kinematics words stay: rewards and the scanner pose recompute the chain into them).
struct LdsPlan {
  int s0w, s1w;                   // staging-row words of the two observation groups (0: one lane per limb, a group without noise goes straight to HBM)
  int fdim, rtdim, rsdim;         // words of an env's feature vector, of its reward tables, of its reward-stage row
  int lb0, lbrec0, envs, tail;    // limb kinematics blocks, limb record blocks, env words, first word behind them
  int rstage, stage0, stage1, fstage;
  int words;                      // the wavefront's region (a multiple of 4: the next wavefront's granules start 16-byte aligned)
};
template <class TP, int SUB>
RL_FN LdsPlan lds_plan(int policy_dim, int critic_dim, bool direct0, bool direct1, int D, int n_bodies, uint64_t rew_ext_mask, int n_rewards) {
  constexpr int EPT = 64 / (NLANE * SUB), LIMBS = 64 / SUB;
  using LS = typename LsFor<TP, SUB>::type;
  constexpr int STASH_WORDS = LsFor<TP, SUB>::STASH * LsFor<TP, SUB>::type::SSW * 64;
  constexpr int LB_FREE = TP::NW > 0 ? LbLayout<TP>::RECW * LIMBS : 0;  // the record blocks of all limbs: one contiguous region
  LdsPlan P;
  P.s0w = (SUB == 1 && direct0) ? 0 : (EPT * policy_dim + 3) & ~3;
  P.s1w = (SUB == 1 && direct1) ? 0 : (EPT * critic_dim + 3) & ~3;
  P.fdim = feat_count(D);
  P.rtdim = rew_tab_words(D, n_bodies, rew_ext_mask);
  P.rsdim = rew_stage_words(n_rewards);  // the task's terms (not MAX_T: 16 envs x 20 unused words were 5 KB of a one-lane-per-limb workgroup)
  int region = P.s0w + P.s1w + EPT * P.fdim;
  if (region < EPT * P.rtdim) P.s1w += EPT * P.rtdim - region, region = EPT * P.rtdim;
  if (SUB > 1 && region < EPT * RESET_RAND_WORDS) P.s1w += EPT * RESET_RAND_WORDS - region, region = EPT * RESET_RAND_WORDS;  // ... or the reset uniforms (env_terms.h reset_uniforms)
  P.lb0 = LS::WORDS * 64;
  P.lbrec0 = P.lb0 + LbLayout<TP>::CHAINW * LIMBS;
  P.envs = P.lb0 + LbLayout<TP>::WORDS * LIMBS;
  P.tail = P.envs + EPT * LbLayout<TP>::ENV_WORDS;
  const bool alias = TP::NW == 0 && STASH_WORDS > 0 && EPT * P.rsdim + region <= STASH_WORDS;
  const bool alias_lb = TP::NW > 0 && EPT * P.rsdim + region <= LB_FREE;
  P.rstage = alias ? LS::CT * 64 : (alias_lb ? P.lbrec0 : P.tail);
  P.stage0 = P.rstage + EPT * P.rsdim;
  P.stage1 = P.stage0 + P.s0w;
  P.fstage = P.stage1 + P.s1w;
  P.words = (alias || alias_lb) ? P.tail : P.tail + EPT * P.rsdim + region;
  P.words = (P.words + 3) & ~3;
  return P;
}

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
// SP: NoSpec - the term stack interpreted from the table image; a Spec of spec/env_specs_gen.h - the step kernel of ONE task, its reward
// terms (and, RESET == 0 only) constant expressions (env_spec.h)
template <class TP, int RESET, int SUB, int WGW, class SP = NoSpec>
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
  // LDS after the tables (only the staged bytes take room: the unused tail of the reward table is never touched): lds_plan above
  const int TAB_F = (int)(S.table_bytes >> 2);
  const LdsPlan P = lds_plan<TP, SUB>(Tl->policy_dim, Tl->critic_dim, direct_group(*Tl, 0), direct_group(*Tl, 1), Tl->D, Tl->n_bodies, Tl->rew_ext_mask, Tl->n_rewards);
  Ctx ctx;
  ctx.T = Tl;
  ctx.Tg = Tgv;
  ctx.dim[0] = Tl->policy_dim;
  ctx.dim[1] = Tl->critic_dim;
  ctx.lscratch = smem + TAB_F + (WGW > 1 ? wv * wave_words : 0u);  // wave_words: LDS words of one wavefront behind the shared tables
  ctx.wtile = (int)blockIdx.x * WGW + wv;
  if (WGW > 1 && ctx.wtile >= S.Npad / Ctx::EPT) return;
  // limb-shared words (trunk + limbs instance): [kinematics block of every limb] [record block of every limb] [env words of every env]
  ctx.lbchain = ctx.lscratch + P.lb0 + (lane / SUB) * LbLayout<TP>::CHAINW;
  ctx.lbrec = ctx.lscratch + P.lbrec0 + (lane / SUB) * LbLayout<TP>::RECW;
  ctx.lbrec_stride = LbLayout<TP>::RECW;
  ctx.envs = ctx.lscratch + P.envs + (lane / Ctx::LPE) * LbLayout<TP>::ENV_WORDS;
  ctx.fdim = P.fdim;
  ctx.rtdim = P.rtdim;
  ctx.rsdim = P.rsdim;
  ctx.rstage = ctx.lscratch + P.rstage;
  ctx.stage[0] = ctx.lscratch + P.stage0;
  ctx.stage[1] = ctx.lscratch + P.stage1;
  ctx.fstage = ctx.lscratch + P.fstage;
  ctx.lane = lane;
  EnvProgram<Ctx, TP, SP> prog(ctx, S);
  if (RESET == 1)
    prog.reset_entry();  // (KMODE_RESET, and KMODE_STEP_TAIL: the second launch of a step split around the command-range decision)
  else if (RESET == 2)
    prog.step_head();    // KMODE_STEP_HEAD: the first launch of such a step
  else
    prog.step();
}


// what the launch helpers need from the env's backend
struct LaunchCfg {
  int device = 0, n_cu = 256;
  int wg_waves = 4;       // RL_ENV_WG=1: single-wavefront workgroups always
  bool wg_force = false;  // RL_ENV_WG=-4: four wavefronts per workgroup whatever the launch size (tests)
};

// lds1: LDS bytes with one wavefront per workgroup; S.mode picks the kernel.  Returns a hipError_t (hipSuccess = launched).
// Kernels per (instance, mapping): step with one / four wavefronts per workgroup; the reset entry (+ the tail of a split step) with
// one; the head of a split step - command-range curricula, which no shipped cfg has - with one and for the 16-lane mapping only
// (rl_env_create keeps such tasks on it).
template <class TP, int SUB, int WGW, class SP = NoSpec>
hipError_t launch_w(const LaunchCfg& cfg, const KState& S, const void* T, size_t lds1, hipStream_t st) {
  const int tiles = S.Npad / (16 / SUB);
  dim3 grid((tiles + WGW - 1) / WGW), block(64 * WGW);
  const uint32_t wave_words = (uint32_t)((lds1 - S.table_bytes) >> 2);
  const size_t lds = S.table_bytes + (size_t)WGW * (lds1 - S.table_bytes);
  // (a specialised instance is the step kernel only: resets and the halves of a split step run the interpreter's kernels)
  constexpr bool HEAD = SUB == 4 && WGW == 1 && !SP::ON, RESET = WGW == 1 && !SP::ON;
  if (lds > 64 * 1024) {
    // opt in to the large LDS carve-out (160 KB per CU on gfx950).  The attribute belongs to the (kernel, device) pair and
    // must cover the LARGEST request: remember per device what was configured and raise it when an env needs more.
    static size_t configured[64] = {};
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = configured[cfg.device & 63];
    if (lds > have) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&env_kernel<TP, 0, SUB, WGW, SP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      if constexpr (RESET) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&env_kernel<TP, 1, SUB, WGW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
      }
      if constexpr (HEAD) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&env_kernel<TP, 2, SUB, WGW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
      }
      have = lds;
    }
  }
  if (S.mode == KMODE_RESET || S.mode == KMODE_STEP_TAIL) {
    if constexpr (RESET) hipLaunchKernelGGL((env_kernel<TP, 1, SUB, WGW>), grid, block, lds, st, S, T, wave_words);
    else return hipErrorInvalidValue;
  } else if (S.mode == KMODE_STEP_HEAD) {
    if constexpr (HEAD) hipLaunchKernelGGL((env_kernel<TP, 2, SUB, WGW>), grid, block, lds, st, S, T, wave_words);
    else return hipErrorInvalidValue;
  } else {
    hipLaunchKernelGGL((env_kernel<TP, 0, SUB, WGW, SP>), grid, block, lds, st, S, T, wave_words);
  }
  return hipGetLastError();
}
template <class TP, int SUB, class SP = NoSpec>
hipError_t launch_cl(const LaunchCfg& cfg, const KState& S, const void* T, size_t lds1, hipStream_t st) {
  // (the trunk + limbs instance with 16 lanes per env gains nothing: 174.4 vs 173.8 us with two wavefronts per workgroup - and its 80 KB
  // per wavefront leave no room; with 32 lanes per env four ~32 KB wavefronts and ONE table image are what lets a CU hold four)
  if constexpr (TP::NW == 0 || SUB == 8) {
    const int tiles = S.Npad / (16 / SUB);
    const size_t lds4 = S.table_bytes + 4 * (lds1 - S.table_bytes);
    // (only the step kernel exists in the four-wavefront shape: resets and the halves of a split step are off the hot path)
    if (cfg.wg_waves == 4 && (tiles >= 4 * cfg.n_cu || cfg.wg_force) && lds4 <= 160 * 1024 && S.mode == KMODE_STEP)
      return launch_w<TP, SUB, 4, SP>(cfg, S, T, lds1, st);
  }
  return launch_w<TP, SUB, 1, SP>(cfg, S, T, lds1, st);
}

// Launch of the step kernel SPECIALISED on the task of Spec SP (env_spec.h) in lane mapping `sub`; a hipError_t, or -2 when there is no
// specialised kernel for the mapping / the launch mode (the caller then launches the interpreter's kernel).  Instantiated once per Spec:
// in that Spec's own translation unit (csrc/spec/rl_env_spec_<id>.hip, so that hipcc compiles the specialised tasks side by side), or -
// single-translation-unit builds of ONE instance (tools/build_variant.sh, tools/kbuild.sh) - in rl_env.hip for -DRL_ENV_SPEC_ONLY=<id>.
template <class SP>
int launch_spec(const LaunchCfg& cfg, const KState& S, const void* T, int sub, size_t lds1, hipStream_t st) {
  using TP = typename SP::TP;
  if (S.mode != KMODE_STEP) return -2;
  if constexpr (TP::NW > 0) {  // trunk + limbs instances: the 32-lanes-per-env mapping (what every launch size picks when the model fits it)
    if (sub == 8) return (int)launch_cl<TP, 8, SP>(cfg, S, T, lds1, st);
    return -2;
  } else {
#ifdef RL_ENV_SPEC_SUB  // (variant builds: one lane mapping only)
    if (sub == RL_ENV_SPEC_SUB) return (int)launch_cl<TP, RL_ENV_SPEC_SUB, SP>(cfg, S, T, lds1, st);
    return -2;
#else
    switch (sub) {
      case 4: return (int)launch_cl<TP, 4, SP>(cfg, S, T, lds1, st);
      case 2: return (int)launch_cl<TP, 2, SP>(cfg, S, T, lds1, st);
      case 1: return (int)launch_cl<TP, 1, SP>(cfg, S, T, lds1, st);
      default: return -2;
    }
#endif
  }
}

}  // namespace
