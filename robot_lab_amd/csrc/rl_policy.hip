// rl_policy.hip - fused MLP inference for gfx950 (MI355X): the actor / critic networks of the reference's PPO
// configs (.../unitree_a1/agents/rsl_rl_ppo_cfg.py:15-22: 45 -> 512 -> 256 -> 128 -> 12, ELU) evaluated in ONE
// kernel launch.  C-ABI: include/rl_policy.h.
//
// Mapping: a workgroup (4 wavefronts) owns a tile of 16 rows (environments): 4096 envs -> 256 workgroups -> one
// per CU.  Activations never leave the CU: the 16 x K tile of a layer's input sits in LDS, the layer's output is
// written to the other LDS buffer.  The contraction runs on the matrix cores in exact fp32
// (v_mfma_f32_16x16x4_f32: D[16x16] += A[16x4] B[4x16]; result = a k-ordered fmaf chain, so parity with an fp32
// reference is round-off only).  Wavefront w accumulates the 16-column output tiles w, w + 4, ... (up to 8
// independent accumulators -> the 40-cycle dependent MFMA latency is hidden).  Both operands are kept in
// FRAGMENT-MAJOR order, so a 16-deep k block costs a lane ONE ds_read_b128 (A: its four k-steps, conflict-free) and
// one global_load_dwordx4 per output tile (B: the host pre-arranges the weights as [k block][tile][lane][4 k-steps],
// 1 KB contiguous per wavefront load, L2-resident: the whole actor is 0.75 MB) for 4 MFMAs per tile.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/rl_policy.h"
#define RL_FN __host__ __device__ __forceinline__
#include "rl_math.h"
#include "rollout/rl_sample.h"

namespace {

constexpr int MT = 16;                              // rows per workgroup
constexpr int KMAX = RL_MLP_MAX_WIDTH;              // widest layer

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct MlpParams {
  int n_layers, act;
  int in_dim, out_dim;
  int KB[RL_MLP_MAX_LAYERS];   // 16-deep k blocks of the layer's input (input width padded to 16)
  int N[RL_MLP_MAX_LAYERS];    // true output width
  int NT8[RL_MLP_MAX_LAYERS];  // groups of 8 output tiles (output width padded to 128 columns: 4 wavefronts x 2 or 8 x 1 tiles per group)
  const float* W[RL_MLP_MAX_LAYERS];  // fragment-major weight image [KB][8 * NT8 tiles][64 lanes][4 k-steps]
  const float* b[RL_MLP_MAX_LAYERS];  // [128 * NT8]
  // split-precision path (layer_s): 32-deep k blocks of the layer's input (width padded to 32), 16-column output tiles the layer
  // computes (non-last layers: padded to the next layer's 32-deep blocks) and the weight image as three bf16 planes
  int KB32[RL_MLP_MAX_LAYERS], NTS[RL_MLP_MAX_LAYERS];
  const uint16_t* Ws[RL_MLP_MAX_LAYERS];  // [KB32][NTS tiles][3 splits][64 lanes][8 k-steps] bf16
};

__device__ inline float activate(float v, int act) {
  switch (act) {
    case RL_ACT_ELU: return v > 0.f ? v : __expf(v) - 1.0f;  // |abs error| ~1e-7: inside the fp32 round-off of the layer
    case RL_ACT_RELU: return fmaxf(v, 0.f);
    default: return tanhf(v);
  }
}

// LDS activation tile, "fragment-major": element (row, k) of the 16 x K tile sits where the A operand of
// v_mfma_f32_16x16x4_f32 wants it - lane (row = lane & 15, ak = lane >> 4) reads the four k-steps of a 16-deep k
// block (k = 16 kb + 4 s + ak, s = 0..3) as ONE aligned ds_read_b128, and consecutive lanes read consecutive
// 16-byte slots (conflict-free).
__device__ inline int lds_index(int row, int k) { return ((((k >> 4) * 4 + (k & 3)) * 16 + row) << 2) + ((k >> 2) & 3); }  // [kb][ak][row][s]

// TPW output tiles per wavefront, RT 16-row tiles per workgroup, WAVES wavefronts per workgroup (tile index of wavefront
// w: w, w + WAVES, ...).  <RT 1, WAVES 4> is the 16-row kernel.  <RT 2, WAVES 8> reuses every weight fragment for two row
// tiles with the same two wavefronts per SIMD as two resident 16-row workgroups: half the L2 -> CU weight stream per flop
// (at 16 rows per workgroup every CU pulls ~32 B / cycle of weights at full MFMA rate, ~20 TB/s of L2 reads chip-wide).
template <int TPW, int RT, int WAVES>
__device__ inline void layer(const MlpParams& P, int l, const float* __restrict__ xin, float* __restrict__ xout, float* __restrict__ y, int row0,
                             int n_rows, int lane, int wave) {  // wave: index of this wavefront's first tile (tiles wave, wave + WAVES, ...)
  // Everything that is the same for the 64 lanes is moved into SGPRs by hand: the parameter block may be reached through a
  // pointer the compiler cannot prove uniform, and then the k-block count, the tile stride and the weight pointer live in VGPRs
  // and every weight load costs 64-bit VALU address arithmetic (two v_mul_lo_u32 + a v_mad_u64_u32 per k block).  That is not
  // free here: the f32 MFMA runs at the vector rate and VALU instructions do NOT issue in its shadow (tools/micro/mfma_rate.hip:
  // 2 VALU per MFMA cost 25 %).  With scalar bases a weight load is `global_load_dwordx4 v, v_lane16, s[base]`.
  const int KB = __builtin_amdgcn_readfirstlane(P.KB[l]);
  const int NT = 8 * __builtin_amdgcn_readfirstlane(P.NT8[l]);  // tiles per k block in the weight image (>= WAVES * TPW: only the
                                                                // tiles that carry real or next-layer-padding columns are computed)
  const int Nl = __builtin_amdgcn_readfirstlane(P.N[l]);
  wave = __builtin_amdgcn_readfirstlane(wave);
  auto uniform_ptr = [](const float* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  };
  const uint64_t Wu = uniform_ptr(P.W[l]), bu = uniform_ptr(P.b[l]);
  constexpr int TILE = MT * KMAX;  // floats of one 16-row LDS tile
  const bool last = l == __builtin_amdgcn_readfirstlane(P.n_layers) - 1;
  // TWO accumulators per output tile (even / odd k-steps, summed in the epilogue): a dependent v_mfma_f32_16x16x4_f32 can start
  // 40 cycles after its predecessor, an independent one after 32, so a wavefront that owns ONE tile (the 16-wavefront kernel's
  // 256- and 128-wide layers) would issue a dependent chain.  Every kernel of this file uses the same split, so they all produce
  // the same bits.  (Measured effect: none - the wavefronts of a SIMD cover each other's dependent gaps; kept because it costs
  // one add per output and removes the question.)
  f32x4 acc[RT][TPW][2];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[r][t][0] = acc[r][t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4* xa = reinterpret_cast<const f32x4*>(xin) + lane;                                    // + kb * 64 (+ r * TILE / 4)
  // the weight pointer is loaded from the parameter block, so the compiler would fall back to FLAT loads (they count on
  // vmcnt AND lgkmcnt: every wait for an LDS operand then drains the whole weight pipeline) - say that it is global memory
  typedef const f32x4 __attribute__((address_space(1))) * GlobalV4;
  const uint32_t lane16 = (uint32_t)lane * 16u;                                                     // the only per-lane part of a weight address
  // software pipeline, two k blocks deep: the operands of blocks kb + 1 and kb + 2 are in flight while block kb's
  // 4 x TPW x RT MFMAs issue (an L2 hit is ~600-800 cycles, a block's MFMAs are 128 x TPW x RT cycles)
  auto load_b = [&](int kb, f32x4 (&b)[TPW]) {
#ifdef MLP_ABL_NOLOAD  // kernel analysis: the weight fragments of the first blocks are reused for the whole layer (wrong results)
    if (kb > 2) return;
#endif
    const int kc = kb < KB ? kb : KB - 1;  // clamped: the tail re-reads the last block instead of branching
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const uint64_t tile = Wu + ((uint64_t)(uint32_t)(kc * NT + wave + WAVES * t) << 10);  // scalar: 1 KB per (k block, tile)
      b[t] = *(GlobalV4)(uintptr_t)(tile + lane16);
    }
  };
  f32x4 b0[TPW], b1[TPW], b2[TPW];
  load_b(0, b0);
  load_b(1, b1);
  // the epilogue's biases travel with the first weight blocks (loaded at the end they are one more exposed L2 round trip)
  float bias_r[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) bias_r[t] = *(const float __attribute__((address_space(1)))*)(uintptr_t)(bu + (uint64_t)(uint32_t)((wave + WAVES * t) * 64) + (uint32_t)((lane & 15) * 4));
  // the A fragment (LDS) of block kb + 1 is read while block kb's MFMAs issue: a wavefront that has the SIMD to itself (the
  // wavefronts of a SIMD do not finish a layer together) would otherwise sit out an LDS round trip per k block
  f32x4 an[RT];
  auto load_a = [&](int kb, f32x4 (&a)[RT]) {
    const int kc = kb < KB ? kb : KB - 1;
#pragma unroll
#ifdef MLP_ABL_NOLDS  // kernel analysis: one activation block for the whole layer (wrong results)
    for (int r = 0; r < RT; ++r) a[r] = xa[r * (TILE / 4) + 0 * kc];
#else
    for (int r = 0; r < RT; ++r) a[r] = xa[kc * 64 + r * (TILE / 4)];
#endif
  };
  load_a(0, an);
  auto mma = [&](int kb, const f32x4 (&b)[TPW]) {
    f32x4 a[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) a[r] = an[r];
    load_a(kb + 1, an);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          acc[r][t][s & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][s], b[t][s], acc[r][t][s & 1], 0, 0, 0);
#ifdef MLP_PIN  // kernel analysis: pin the issue order (hipcc sorts the MFMAs into runs on one accumulator); measured +3 %, i.e. worse
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
  };
  int kb = 0;
  for (; kb + 3 <= KB; kb += 3) {  // rotate the three buffers by unrolling three blocks
    load_b(kb + 2, b2);
    mma(kb, b0);
    load_b(kb + 3, b0);
    mma(kb + 1, b1);
    load_b(kb + 4, b1);
    mma(kb + 2, b2);
  }
  if (kb < KB) {
    mma(kb, b0);
    if (kb + 1 < KB) mma(kb + 1, b1);
  }
  // epilogue: D[row = (lane >> 4) * 4 + reg][col = lane & 15] -> bias, activation -> next LDS tile / global
  const int col = lane & 15, rbase = (lane >> 4) * 4;
  const int act = __builtin_amdgcn_readfirstlane(P.act);  // wave-uniform: the selects below become scalar branches around straight-line code
  // lds_index(rbase + r, (wave + WAVES t) * 16 + col) = lane-constant + 256 WAVES t + 4 r
  const int obase = wave * 256 + (col & 3) * 64 + rbase * 4 + ((col >> 2) & 3);
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int n = (wave + WAVES * t) * 16 + col;
    const float bias = bias_r[t];
    const bool valid = n < Nl;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (acc[rt][t][0][r] + acc[rt][t][1][r]) + bias;
#ifdef MLP_ABL_NOEPI  // kernel analysis: no activation, one store per accumulator register group (wrong results)
        if (r == 0 && v == 1.2345e-30f) xout[0] = v;
        continue;
#endif
        if (!last) {
          const float av = act == RL_ACT_ELU ? (v > 0.f ? v : __expf(v) - 1.0f) : act == RL_ACT_RELU ? fmaxf(v, 0.f) : tanhf(v);
          xout[rt * TILE + obase + 256 * WAVES * t + 4 * r] = valid ? av : 0.f;  // padded columns feed zeros into the next layer
        } else if (valid && row0 + rt * MT + rbase + r < n_rows) {
          y[(size_t)(row0 + rt * MT + rbase + r) * P.out_dim + n] = v;
        }
      }
  }
}

template <int RT, int WAVES>
__device__ __forceinline__ void mlp_tile(const MlpParams& P, const float* __restrict__ x, float* __restrict__ y, int n_rows, int tile) {
  extern __shared__ float4 smem4[];  // two buffers of RT [16 x KMAX] fragment-major activation tiles: RT x 64 KB
  float* buf0 = reinterpret_cast<float*>(smem4);
  float* buf1 = buf0 + RT * MT * KMAX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = tile * (RT * MT);
  const int K0 = P.KB[0] * 16;
  // input tile -> LDS (zero fill of the k padding and of rows past the end): a wavefront per row, lanes over the columns
  for (int r = wave; r < RT * MT; r += WAVES) {
    const bool live = row0 + r < n_rows;
    const float* __restrict__ xr = x + (size_t)(row0 + r) * P.in_dim;
    float* __restrict__ dst = buf0 + (r >> 4) * (MT * KMAX);
    for (int c = lane; c < K0; c += 64) dst[lds_index(r & 15, c)] = (live && c < P.in_dim) ? xr[c] : 0.f;
  }
  __syncthreads();
  float *cur = buf0, *nxt = buf1;
  for (int l = 0; l < P.n_layers; ++l) {
    const int tpw = ((P.N[l] + 15) / 16 + WAVES - 1) / WAVES;  // 16-column tiles per wavefront that carry output columns
    // at most 4 tiles (8 accumulators) per wavefront and pass: wider layers take two passes over the input tile (LDS re-read)
    for (int done = 0; done < tpw; done += 4) {
      const int first = wave + WAVES * done;
      switch (tpw - done >= 4 ? 4 : tpw - done) {
        case 1: layer<1, RT, WAVES>(P, l, cur, nxt, y, row0, n_rows, lane, first); break;
        case 2: layer<2, RT, WAVES>(P, l, cur, nxt, y, row0, n_rows, lane, first); break;
        case 3: layer<3, RT, WAVES>(P, l, cur, nxt, y, row0, n_rows, lane, first); break;
        default: layer<4, RT, WAVES>(P, l, cur, nxt, y, row0, n_rows, lane, first); break;
      }
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
}

__global__ __launch_bounds__(256) void mlp_forward_kernel(MlpParams P, const float* __restrict__ x, float* __restrict__ y, int n_rows) {
  mlp_tile<1, 4>(P, x, y, n_rows, blockIdx.x);
}

// Two networks over the same rows in ONE launch (actor + critic of a rollout step): even workgroups run network A, odd
// ones network B on the same row tile.  A 4096-row call of one network is 256 16-row workgroups = one per CU = one
// wavefront per SIMD, which cannot hide the L2 latency of the weight stream behind its own MFMAs (the 65536-row call
// runs 1.55x faster per row with two workgroups resident per CU); the pair puts two wavefronts on every SIMD.
struct MlpPair {
  const MlpParams *a, *b;  // device copies
  const float *xa, *xb;
  float *ya, *yb;
  unsigned long long* clk;  // -DRL_MLP_CLOCK: [workgroup][16] s_memtime stamps of wavefront 0 (kernel analysis builds only)
};
#ifdef RL_MLP_CLOCK
#define MLP_STAMP(i) do { if ((threadIdx.x & 63) == 0) q.clk[(size_t)blockIdx.x * 256 + (threadIdx.x >> 6) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define MLP_STAMP(i) do { } while (0)
#endif
template <int RT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void mlp_forward_pair_kernel(MlpPair q, int n_rows) {
  const bool second = blockIdx.x & 1;
  const MlpParams& P = *(second ? q.b : q.a);  // workgroup-uniform address: the fields arrive by scalar loads
  mlp_tile<RT, WAVES>(P, second ? q.xb : q.xa, second ? q.yb : q.ya, n_rows, blockIdx.x >> 1);
}

// ---- actor + critic, FUSED per row tile ------------------------------------------------------------------------------------
// One workgroup (8 wavefronts) owns 16 rows and runs BOTH networks on them, layer by layer: in layer l the 16-column output
// tiles of network A and of network B form one list that is dealt round-robin to the eight wavefronts, so every wavefront has
// the same work in every layer whatever the two networks' widths are (the critic of the A1 task is 1.5x the actor: with one
// network per workgroup the actor's CUs idle a third of the call), the 1-tile last layers land on two different SIMDs instead
// of being padded to eight tiles, and each SIMD carries two wavefronts for the whole call.  4096 rows = 256 workgroups = one
// per CU.  LDS: two activation buffers per network, 4 x 32 KB.  Both networks must have the same number of layers (one
// barrier per layer).  16 wavefronts per workgroup (4 per SIMD, 1 - 2 tiles each) by default: the kernel enters a rollout loop
// with cold caches (the env step ran in between) and four wavefronts per SIMD ride out the misses that two cannot
// (RL_MLP_FUSED_WAVES=8 selects the 8-wavefront build: same speed with warm caches, 12 us slower cold).
// Measured and dropped: an explicit L2 warm-up sweep of the weight image at kernel start (+5 us in the loop), a deeper
// asm-pinned weight pipeline (the L2-hit latency of ~200 cycles is already covered; +3 us).
template <int WAVES>
__device__ __forceinline__ void stage_rows(const MlpParams& P, const float* __restrict__ x, float* __restrict__ dst, int row0, int n_rows, int lane, int wave) {
  // 16 rows over WAVES wavefronts (rows wave, wave + WAVES, ...); every load of the wavefront is in flight before the first LDS
  // write (the rows were just written by another kernel: each is an L2 miss, and a load -> wait -> write loop pays that per load)
  constexpr int JMAX = KMAX / 64, H = MT / WAVES;
  const int K0 = P.KB[0] * 16;
  float v[H][JMAX];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const int r = wave + WAVES * h;
    const bool live = row0 + r < n_rows;
    const float* __restrict__ xr = x + (size_t)(row0 + r) * P.in_dim;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int c = lane + 64 * j;
      v[h][j] = (64 * j < K0 && live && c < P.in_dim) ? xr[c] : 0.f;
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int c = lane + 64 * j;
      if (64 * j < K0 && c < K0) dst[lds_index(wave + WAVES * h, c)] = v[h][j];
    }
}

// the tiles first, first + WAVES, ... < nt of layer l (none: returns)
template <int WAVES>
__device__ __forceinline__ void fused_segment(const MlpParams& P, int l, const float* xin, float* xout, float* y, int row0, int n_rows, int lane, int first) {
  const int nt = (P.N[l] + 15) >> 4;
  if (first >= nt) return;
  const int cnt = (nt - first + WAVES - 1) / WAVES;
  if constexpr (WAVES == 8) {
    switch (cnt) {
      case 1: layer<1, 1, WAVES>(P, l, xin, xout, y, row0, n_rows, lane, first); break;
      case 2: layer<2, 1, WAVES>(P, l, xin, xout, y, row0, n_rows, lane, first); break;
      case 3: layer<3, 1, WAVES>(P, l, xin, xout, y, row0, n_rows, lane, first); break;
      default: layer<4, 1, WAVES>(P, l, xin, xout, y, row0, n_rows, lane, first); break;  // 512 columns = 32 tiles = 4 per wavefront
    }
  } else {
    if (cnt == 1) layer<1, 1, WAVES>(P, l, xin, xout, y, row0, n_rows, lane, first);
    else layer<2, 1, WAVES>(P, l, xin, xout, y, row0, n_rows, lane, first);  // 512 columns = 32 tiles = 2 per wavefront
  }
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void mlp_fused_pair_kernel(MlpPair q, int n_rows) {
  extern __shared__ float4 smem4[];
  constexpr int TILE = MT * KMAX;
  float* bufA = reinterpret_cast<float*>(smem4);
  float* bufB = bufA + 2 * TILE;
  const MlpParams& A = *q.a;
  const MlpParams& B = *q.b;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * MT;
  MLP_STAMP(0);
  stage_rows<WAVES>(A, q.xa, bufA, row0, n_rows, lane, wave);
  stage_rows<WAVES>(B, q.xb, bufB, row0, n_rows, lane, wave);
  MLP_STAMP(1);
  __syncthreads();
  MLP_STAMP(2);
  int cur = 0;
  const bool a_first = ((wave >> 2) & 1) == 0;
  for (int l = 0; l < A.n_layers; ++l) {
    const int ntA = (A.N[l] + 15) >> 4;
    const int firstB = (wave - ntA) & (WAVES - 1);  // B's tiles continue A's round-robin
    // Wavefronts that share a SIMD (w, w + 4, ...) take their two segments in alternating order: the pipeline fill / bias + ELU +
    // LDS-store drain of one (several thousand cycles per segment, whatever its length) falls into another one's MFMA stream.
    // ONE call site (the network is picked by a wavefront-uniform pointer): the layer code is inlined once, not four times -
    // the kernel must fit the instruction cache, which it enters cold in a rollout loop (the env step ran in between).
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const bool isA = (h == 0) == a_first;
      const MlpParams& P = isA ? A : B;
      float* buf = isA ? bufA : bufB;
      fused_segment<WAVES>(P, l, buf + cur * TILE, buf + (cur ^ 1) * TILE, isA ? q.ya : q.yb, row0, n_rows, lane, isA ? wave : firstB);
      MLP_STAMP(3 + 3 * l + h);
    }
    __syncthreads();
    MLP_STAMP(5 + 3 * l);
    cur ^= 1;
  }
}


// ---- split-precision path: fp32 results from the bf16 matrix cores ---------------------------------------------------------------
// An fp32 number is the EXACT sum of three bf16 numbers when each split truncates (hi = top 16 bits of x, mid = top 16 bits of
// x - hi, lo = x - hi - mid: every step peels at least 8 of the 24 significand bits).  With both operands split that way,
//     a b = (ah + am + al)(bh + bm + bl) = ah bh + [ah bm + am bh + am bm + ah bl + al bh] + O(2^-24 |a b|)
// and every bf16 x bf16 product is exact in the fp32 accumulator of v_mfma_f32_16x16x32_bf16.  Six MFMAs of the bf16 rate
// (16x the f32 MFMA rate per flop) replace 8 k-steps of v_mfma_f32_16x16x4_f32: 16 / 6 = 2.7x the contraction rate of the exact
// f32 path, and the result is NOT lower precision - the three dropped products are below fp32 round-off of the term, and the sum
// sees one fp32 rounding per 32-deep block instead of one per k (measured on the A1 networks against the fp64 oracle: max |err|
// 8e-7 vs 2e-6 for an fp32 fmaf chain; tests/test_policy.py keeps rtol = atol = 2e-5).  The small products accumulate in their
// own registers so that they are not rounded away against the running main sum.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Split3 {
  uint16_t h, m, l;
};
__host__ __device__ inline Split3 split3(float v) {
  union { float f; uint32_t u; } a, b, c, t;
  a.f = v;
  t.u = a.u & 0xffff0000u;
  b.f = v - t.f;  // exact
  t.u = b.u & 0xffff0000u;
  c.f = b.f - t.f;  // exact, <= 8 significant bits left
  Split3 r{(uint16_t)(a.u >> 16), (uint16_t)(b.u >> 16), (uint16_t)(c.u >> 16)};
  if ((a.u & 0x7f800000u) == 0x7f800000u) r.m = r.l = 0;  // inf / nan: keep it in the leading part only (inf - inf would poison the rest)
  return r;
}

// LDS activation tile of the split path, in bf16 units: element (row, k) of plane s sits where the A operand of
// v_mfma_f32_16x16x32_bf16 wants it - lane (row = lane & 15, kg = lane >> 4) reads its eight k-steps k = 32 kb + 8 kg + j of a
// 32-deep block as ONE aligned ds_read_b128 at 16-byte slot (kb * 3 + s) * 64 + lane (conflict-free).
__device__ inline int lds_index_s(int row, int k, int s) { return (((((k >> 5) * 3 + s) * 4 + ((k >> 3) & 3)) * 16 + row) << 3) + (k & 7); }

// RT 16-row tiles per workgroup (row tile r of an activation buffer starts `tstride` bf16 units behind row tile r - 1): every weight
// fragment is used for RT row tiles, i.e. 1 / RT of the L2 -> CU weight stream per flop.  DEPTH weight buffers (k blocks in flight).
template <int TPW, int WAVES, int RT = 1, int DEPTH = 2>
__device__ inline void layer_s(const MlpParams& P, int l, const uint16_t* __restrict__ xin, uint16_t* __restrict__ xout, float* __restrict__ y, int row0,
                               int n_rows, int lane, int wave, int tstride_in = 0, int tstride_out = 0) {  // wave: this wavefront's first tile (tiles wave, wave + WAVES, ...)
  static_assert(DEPTH == 2 || DEPTH == 3, "two or three weight buffers");
  const int KB = __builtin_amdgcn_readfirstlane(P.KB32[l]);
  const int NT = __builtin_amdgcn_readfirstlane(P.NTS[l]);
  const int Nl = __builtin_amdgcn_readfirstlane(P.N[l]);
  wave = __builtin_amdgcn_readfirstlane(wave);
  auto uniform_ptr = [](const void* p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  };
  const uint64_t Wu = uniform_ptr(P.Ws[l]), bu = uniform_ptr(P.b[l]);
  const bool last = l == __builtin_amdgcn_readfirstlane(P.n_layers) - 1;
  // three accumulators per output tile: the leading product, and the five small ones on two chains
  f32x4 acc[RT][TPW][3];
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[r][t][0] = acc[r][t][1] = acc[r][t][2] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef const f32x4 __attribute__((address_space(1))) * GlobalV4;
  const uint32_t lane16 = (uint32_t)lane * 16u;
  struct BFrag {
    f32x4 p[3];  // hi, mid, lo plane of the tile's 32 x 16 weight block: 8 bf16 per lane and plane
  };
  auto load_b = [&](int kb, BFrag (&b)[TPW]) {
    const int kc = kb < KB ? kb : KB - 1;  // clamped: the tail re-reads the last block instead of branching
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      // a wavefront's slot t may lie behind the layer's last tile (odd tile counts): it re-reads the last tile and its result is dropped
      const int tile_i = wave + WAVES * t < NT ? wave + WAVES * t : NT - 1;
      const uint64_t tile = Wu + (uint64_t)(uint32_t)(kc * NT + tile_i) * 3072u;  // scalar: 3 KB per (k block, tile)
#pragma unroll
      for (int s = 0; s < 3; ++s) b[t].p[s] = *(GlobalV4)(uintptr_t)(tile + (uint32_t)(s * 1024) + lane16);
    }
  };
  BFrag bq[DEPTH][TPW];
  load_b(0, bq[0]);
  if (DEPTH == 3) load_b(1, bq[1]);
  float bias_r[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) bias_r[t] = *(const float __attribute__((address_space(1)))*)(uintptr_t)(bu + (uint64_t)(uint32_t)((wave + WAVES * t) * 64) + (uint32_t)((lane & 15) * 4));
  const f32x4* xa = reinterpret_cast<const f32x4*>(xin) + lane;  // plane s of block kb of row tile r: + (kb * 3 + s) * 64 + r * tstride_in / 8
  const int ts4 = tstride_in >> 3;
  f32x4 an[RT][3];
  auto load_a = [&](int kb, f32x4 (&a)[RT][3]) {
    const int kc = kb < KB ? kb : KB - 1;
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) a[r][s] = xa[(kc * 3 + s) * 64 + r * ts4];
  };
  load_a(0, an);
  auto mma = [&](int kb, const BFrag (&b)[TPW]) {
    bf16x8 a[RT][3];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) a[r][s] = __builtin_bit_cast(bf16x8, an[r][s]);
    load_a(kb + 1, an);
#define RL_MMA(ai, bi, ci) acc[r][t][ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[r][ai], __builtin_bit_cast(bf16x8, b[t].p[bi]), acc[r][t][ci], 0, 0, 0)
#define RL_MMA_ALL(ai, bi, ci)                                  \
  _Pragma("unroll") for (int t = 0; t < TPW; ++t)               \
      _Pragma("unroll") for (int r = 0; r < RT; ++r) RL_MMA(ai, bi, ci);
    RL_MMA_ALL(0, 0, 0)  // hi hi
    RL_MMA_ALL(0, 1, 1)  // hi mid
    RL_MMA_ALL(1, 0, 2)  // mid hi
    RL_MMA_ALL(1, 1, 1)  // mid mid
    RL_MMA_ALL(0, 2, 2)  // hi lo
    RL_MMA_ALL(2, 0, 1)  // lo hi
#undef RL_MMA_ALL
#undef RL_MMA
  };
  int kb = 0;
  if (DEPTH == 2) {
    for (; kb + 2 <= KB; kb += 2) {  // two weight buffers, rotated by unrolling two blocks
      load_b(kb + 1, bq[1]);
      mma(kb, bq[0]);
      load_b(kb + 2, bq[0]);
      mma(kb + 1, bq[1]);
    }
    if (kb < KB) mma(kb, bq[0]);
  } else {
    constexpr int D2 = DEPTH - 1;  // (= 2: written so that the DEPTH == 2 instantiation indexes inside its array)
    for (; kb + 3 <= KB; kb += 3) {  // three weight buffers: blocks kb + 1 and kb + 2 in flight while block kb's MFMAs issue
      load_b(kb + 2, bq[D2]);
      mma(kb, bq[0]);
      load_b(kb + 3, bq[0]);
      mma(kb + 1, bq[1]);
      load_b(kb + 4, bq[1]);
      mma(kb + 2, bq[D2]);
    }
    if (kb < KB) {
      mma(kb, bq[0]);
      if (kb + 1 < KB) mma(kb + 1, bq[1]);
    }
  }
  // epilogue: D[row = (lane >> 4) * 4 + reg][col = lane & 15] -> bias, activation -> the next layer's three planes / global
  const int col = lane & 15, rbase = (lane >> 4) * 4;
  const int act = __builtin_amdgcn_readfirstlane(P.act);
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int n = (wave + WAVES * t) * 16 + col;
    const float bias = bias_r[t];
    const bool valid = n < Nl, in_layer = wave + WAVES * t < NT;
    const int o = lds_index_s(rbase, n, 0);  // + 8 r per row, + 512 s per plane
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = ((acc[rt][t][1][r] + acc[rt][t][2][r]) + acc[rt][t][0][r]) + bias;
        if (!last) {
          const float av = act == RL_ACT_ELU ? (v > 0.f ? v : __expf(v) - 1.0f) : act == RL_ACT_RELU ? fmaxf(v, 0.f) : tanhf(v);
          const Split3 sp = split3(valid ? av : 0.f);  // padded columns feed zeros into the next layer
          if (in_layer) {
            uint16_t* xo = xout + rt * tstride_out + o + 8 * r;
            xo[0] = sp.h;
            xo[512] = sp.m;
            xo[1024] = sp.l;
          }
        } else if (valid && in_layer && row0 + rt * MT + rbase + r < n_rows) {
          y[(size_t)(row0 + rt * MT + rbase + r) * P.out_dim + n] = v;
        }
      }
  }
}

template <int WAVES>
__device__ __forceinline__ void stage_rows_s(const MlpParams& P, const float* __restrict__ x, uint16_t* __restrict__ dst, int row0, int n_rows, int lane, int wave,
                                             float* __restrict__ copy_dst = nullptr) {
  // 16 rows over WAVES wavefronts; every load of the wavefront is in flight before the first LDS write (see stage_rows)
  constexpr int JMAX = KMAX / 64, H = MT / WAVES;
  const int K0 = P.KB32[0] * 32;
  float v[H][JMAX];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const int r = wave + WAVES * h;
    const bool live = row0 + r < n_rows;
    const float* __restrict__ xr = x + (size_t)(row0 + r) * P.in_dim;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int c = lane + 64 * j;
      v[h][j] = (64 * j < K0 && live && c < P.in_dim) ? xr[c] : 0.f;
    }
  }
  if (copy_dst != nullptr) {  // (act epilogue, include/rl_act.h) the rows go into the rollout slot while they are in registers
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int r = wave + WAVES * h;
      if (row0 + r < n_rows) {
        float* __restrict__ dr = copy_dst + (size_t)(row0 + r) * P.in_dim;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
          const int c = lane + 64 * j;
          if (64 * j < K0 && c < P.in_dim) dr[c] = v[h][j];
        }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int c = lane + 64 * j;
      if (64 * j < K0 && c < K0) {
        const Split3 sp = split3(v[h][j]);
        const int o = lds_index_s(wave + WAVES * h, c, 0);
        dst[o] = sp.h;
        dst[o + 512] = sp.m;
        dst[o + 1024] = sp.l;
      }
    }
}

template <int WAVES>
__device__ __forceinline__ void split_segment(const MlpParams& P, int l, const uint16_t* xin, uint16_t* xout, float* y, int row0, int n_rows, int lane, int first) {
  const int nt = P.NTS[l];
  if (first >= nt) return;
  const int cnt = (nt - first + WAVES - 1) / WAVES;
  if (cnt == 1) layer_s<1, WAVES>(P, l, xin, xout, y, row0, n_rows, lane, first);
  else layer_s<2, WAVES>(P, l, xin, xout, y, row0, n_rows, lane, first);  // 512 columns = 32 tiles = 2 per wavefront
}

// One workgroup (16 wavefronts) owns 16 rows and runs network A and - when q.b is given - network B on them, layer by layer, as
// mlp_fused_pair_kernel does (tiles of the two networks dealt round-robin, SIMD-mates alternate the order of their two segments).
// LDS: per network two activation buffers of three bf16 planes; layer l reads buffer l & 1 (widths: host, rl_mlp_split_lds).
struct SplitLds {
  int a1, b0, b1;  // offsets (in bf16 units) of A's second buffer and of B's two buffers; A's first buffer starts at 0
};
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void mlp_split_kernel(MlpPair q, int n_rows, SplitLds o, rl_act_epilogue ep) {  // (ep.num_envs > 0: as mlp_split2_kernel)
  extern __shared__ float4 smem4[];
  uint16_t* base = reinterpret_cast<uint16_t*>(smem4);
  const MlpParams& A = *q.a;
  const bool two = q.b != nullptr;
  const MlpParams& B = two ? *q.b : *q.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * MT;
  const bool act = ep.num_envs > 0;
  stage_rows_s<WAVES>(A, q.xa, base, row0, n_rows, lane, wave, act ? ep.s_obs : nullptr);
  if (two) stage_rows_s<WAVES>(B, q.xb, base + o.b0, row0, n_rows, lane, wave, act ? ep.s_critic_obs : nullptr);
  __syncthreads();
  const bool a_first = ((wave >> 2) & 1) == 0;
  for (int l = 0; l < A.n_layers; ++l) {
    const int ntA = A.NTS[l];
    const int firstB = (wave - ntA) & (WAVES - 1);  // B's tiles continue A's round-robin
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const bool isA = (h == 0) == a_first;
      if (!isA && !two) continue;
      const MlpParams& P = isA ? A : B;
      const int even = isA ? 0 : o.b0, odd = isA ? o.a1 : o.b1;  // (offsets, not an array of pointers indexed by l: that would live in scratch)
      split_segment<WAVES>(P, l, base + ((l & 1) ? odd : even), base + ((l & 1) ? even : odd), isA ? q.ya : q.yb, row0, n_rows, lane, isA ? wave : firstB);
    }
    __syncthreads();
  }
  if (act) {  // the 16 rows' stochastic head (see mlp_split2_kernel): one thread per (row, Philox block of 4 actions)
    float* part = reinterpret_cast<float*>(base);
    const int rows = min(MT, n_rows - row0);
    const int nA = ep.act_dim, nblk = (nA + 3) >> 2;
    const int el = tid / nblk, blk = tid - el * nblk, e = row0 + el;
    const bool live = el < rows;
    float logp = 0.f;
    if (live) logp = rl::act_block(ep, q.ya, e, blk);
    part[tid] = logp;
    __syncthreads();
    if (live && blk == 0) {
      float sacc = 0.f;
      for (int i = 0; i < nblk; ++i) sacc += part[tid + i];
      ep.s_logp[e] = sacc;
    }
  }
}

// ---- 32 rows of ONE network per workgroup (the rollout size) --------------------------------------------------------------------
// mlp_split_kernel runs both networks on 16 rows: every CU streams all of both networks' weights (2.8 MB of bf16 planes for the A1
// pair, 726 MB out of the L2s per call) and pays a pipeline fill / drain / barrier for each of its eight (network, layer) segments -
// measured, that overhead and the weight stream, not the MFMAs, are what the call costs (44.7 us against 10 us of MFMA issue,
// profiles/r03b_policy.txt).  Here a workgroup owns 32 rows of one network (even workgroups network A, odd ones network B on the
// same rows): each weight fragment feeds two row tiles (half the L2 -> CU bytes per flop), a workgroup has four segments instead of
// eight, each twice as long.  8 wavefronts (2 per SIMD, 256 registers each), two output tiles per wavefront and pass, three
// weight buffers.  LDS: (widest even-layer input + widest odd-layer input) x 32 rows x three bf16 planes - 147 KB for the A1 critic.
// The actor's workgroups finish early (0.47x the critic's flops); the call is as long as a critic workgroup.
template <int WAVES>
__device__ __forceinline__ void stage_rows_s2(const MlpParams& P, const float* __restrict__ x, uint16_t* __restrict__ dst, int tstride, int row0, int n_rows, int lane, int wave,
                                              float* __restrict__ copy_dst = nullptr) {
  // 32 rows over WAVES wavefronts (rows wave, wave + WAVES, ...), all loads of the wavefront in flight before the first LDS write
  constexpr int JMAX = KMAX / 64, H = 2 * MT / WAVES;
  const int K0 = P.KB32[0] * 32;
  float v[H][JMAX];
#pragma unroll
  for (int h = 0; h < H; ++h) {
    const int r = wave + WAVES * h;
    const bool live = row0 + r < n_rows;
    const float* __restrict__ xr = x + (size_t)(row0 + r) * P.in_dim;
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int c = lane + 64 * j;
      v[h][j] = (64 * j < K0 && live && c < P.in_dim) ? xr[c] : 0.f;
    }
  }
  if (copy_dst != nullptr) {  // (act epilogue, include/rl_act.h) the rows go into the rollout slot while they are in registers: stores that ride under the layers
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int r = wave + WAVES * h;
      if (row0 + r < n_rows) {
        float* __restrict__ dr = copy_dst + (size_t)(row0 + r) * P.in_dim;
#pragma unroll
        for (int j = 0; j < JMAX; ++j) {
          const int c = lane + 64 * j;
          if (64 * j < K0 && c < P.in_dim) dr[c] = v[h][j];
        }
      }
    }
  }
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      const int c = lane + 64 * j, r = wave + WAVES * h;
      if (64 * j < K0 && c < K0) {
        const Split3 sp = split3(v[h][j]);
        uint16_t* d = dst + (r >> 4) * tstride + lds_index_s(r & 15, c, 0);
        d[0] = sp.h;
        d[512] = sp.m;
        d[1024] = sp.l;
      }
    }
}

// ep.num_envs > 0: the rollout step's stochastic head behind the actor's last layer (include/rl_act.h), see the end of the kernel.  (A run-time
// flag of the ONE kernel, not a second instantiation: with two kernels calling them the layer functions stop being inlined - 248 registers and
// 84 B of scratch instead of 162 and none, 36.6 -> 47.8 us for the pair.)
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void mlp_split2_kernel(MlpPair q, int n_rows, int cols_a0, int cols_a1, int cols_b0, int cols_b1, rl_act_epilogue ep) {
  extern __shared__ float4 smem4[];
  uint16_t* base = reinterpret_cast<uint16_t*>(smem4);
  const bool two = q.b != nullptr;
  const bool second = two && (blockIdx.x & 1);
  const MlpParams& P = *(second ? q.b : q.a);  // workgroup-uniform: scalar loads
  const float* x = second ? q.xb : q.xa;
  float* y = second ? q.yb : q.ya;
  const int c0 = second ? cols_b0 : cols_a0, c1 = second ? cols_b1 : cols_a1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = (two ? (int)(blockIdx.x >> 1) : (int)blockIdx.x) * (2 * MT);
  // buffer of the even layers' inputs: 2 row tiles of c0 columns; behind it the odd layers': 2 row tiles of c1 columns
  const int ts0 = c0 * MT * 3, ts1 = c1 * MT * 3;
  uint16_t* buf0 = base;
  uint16_t* buf1 = base + 2 * ts0;
  stage_rows_s2<WAVES>(P, x, buf0, ts0, row0, n_rows, lane, wave, ep.num_envs > 0 ? (second ? ep.s_critic_obs : ep.s_obs) : nullptr);
  __syncthreads();
  for (int l = 0; l < P.n_layers; ++l) {
    const int nt = P.NTS[l];
    const uint16_t* xin = (l & 1) ? buf1 : buf0;
    uint16_t* xout = (l & 1) ? buf0 : buf1;
    const int tsi = (l & 1) ? ts1 : ts0, tso = (l & 1) ? ts0 : ts1;
    // two tiles per wavefront and pass: tiles first, first + WAVES with first = wave + 2 WAVES pass
    for (int first = wave; first < nt; first += 2 * WAVES) {
      if (first + WAVES < nt) layer_s<2, WAVES, 2, 3>(P, l, xin, xout, y, row0, n_rows, lane, first, tsi, tso);
      else layer_s<1, WAVES, 2, 3>(P, l, xin, xout, y, row0, n_rows, lane, first, tsi, tso);
    }
    __syncthreads();
  }
  if (ep.num_envs > 0) {
    // The workgroup's 32 rows of the step's transition, first half (what act_kernel of rl_rollout.hip does in a launch of its own):
    // the actor's workgroup samples from the means it has just written (visible to the whole workgroup behind the barrier above) -
    // one thread per (row, Philox block of 4 actions), an env's log-prob partials meet in LDS in ascending block order.  (Both networks'
    // workgroups copied their observation rows into the slot when they staged them; the critic's V went to the slot as `y`.  The actor is
    // the shorter network: its workgroups sample while the critic's are still in their layers.)
    float* part = reinterpret_cast<float*>(base);  // (the activation buffers are dead)
    const int rows = min(2 * MT, n_rows - row0);
    if (!second) {
      const int A = ep.act_dim, nblk = (A + 3) >> 2;
      const int el = tid / nblk, blk = tid - el * nblk, e = row0 + el;
      const bool live = el < rows;
      float logp = 0.f;
      if (live) logp = rl::act_block(ep, y, e, blk);
      part[tid] = logp;
      __syncthreads();
      if (live && blk == 0) {
        float sacc = 0.f;
        for (int i = 0; i < nblk; ++i) sacc += part[tid + i];  // fixed order: blocks 0, 1, ...
        ep.s_logp[e] = sacc;
      }
    }
  }
}

std::string& err() {
  static thread_local std::string e;
  return e;
}
int fail(const std::string& m) {
  err() = m;
  return -1;
}

}  // namespace

struct rl_mlp {
  MlpParams P;
  MlpParams* dP = nullptr;  // device copy (rl_mlp_forward_pair)
  int device = 0;
  int cols[2] = {0, 0};  // split path: widest (32-padded) layer input held by the even / the odd activation buffer
  int dims[RL_MLP_MAX_LAYERS + 1] = {};  // true layer widths (rl_mlp_set_weights re-reads nn.Linear images of these shapes)
  std::vector<void*> allocs;
};

namespace {
// RL_MLP_PRECISION=f32 selects the exact-f32 MFMA kernels (v_mfma_f32_16x16x4_f32); default: the split-bf16 path (layer_s), which
// is at least as accurate (see its header) and 2.7x the contraction rate.  Networks whose activation planes do not fit the 160 KB
// of LDS fall back to the f32 kernels by themselves.
bool split_wanted() {  // (read at every call: the tests switch it inside one process)
  const char* e = getenv("RL_MLP_PRECISION");
  return !(e && (e[0] == 'f' || e[0] == 'F'));
}
constexpr size_t LDS_MAX = 160 * 1024;
size_t split_lds_bytes(const rl_mlp* m) { return (size_t)(m->cols[0] + m->cols[1]) * MT * 3 * sizeof(uint16_t); }
// 32 rows x one network per workgroup (mlp_split2_kernel) once that fills the chip's CUs; RL_MLP_SPLIT_RT=1|2 forces either kernel
int launch_split2(rl_mlp* a, const float* xa, float* ya, rl_mlp* b, const float* xb, float* yb, int n_rows, void* stream, const rl_act_epilogue* ep = nullptr) {
  const size_t la = 2 * split_lds_bytes(a), lb = b ? 2 * split_lds_bytes(b) : 0, lds = std::max({la, lb, (size_t)(ep ? 512 * sizeof(float) : 0)});
  static size_t configured[64] = {};
  if (lds > configured[a->device & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_split2_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail("cannot reserve the LDS of the 32-row split-precision kernel");
    configured[a->device & 63] = lds;
  }
  MlpPair q{a->dP, b ? b->dP : nullptr, xa, xb, ya, yb, nullptr};
  const int tiles = (n_rows + 2 * MT - 1) / (2 * MT);
  rl_act_epilogue off{};  // (num_envs = 0: no epilogue)
  hipLaunchKernelGGL(mlp_split2_kernel<8>, dim3(b ? 2 * tiles : tiles), dim3(512), lds, (hipStream_t)stream, q, n_rows, a->cols[0], a->cols[1],
                     b ? b->cols[0] : 0, b ? b->cols[1] : 0, ep ? *ep : off);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(hipGetErrorString(e));
}
int launch_split(rl_mlp* a, const float* xa, float* ya, rl_mlp* b, const float* xb, float* yb, int n_rows, void* stream, const rl_act_epilogue* ep = nullptr) {
  {
    const char* e = getenv("RL_MLP_SPLIT_RT");
    const int forced = e ? atoi(e) : 0;
    const size_t need2 = 2 * std::max(split_lds_bytes(a), b ? split_lds_bytes(b) : (size_t)0);
    // from 4096 rows on (128 row tiles x 2 networks = one workgroup per CU; a single network: 8192 rows)
    const bool big = (size_t)n_rows * (b ? 2 : 1) >= 8192;
    if (need2 <= LDS_MAX && (forced == 2 || (forced != 1 && big))) return launch_split2(a, xa, ya, b, xb, yb, n_rows, stream, ep);
  }
  const size_t la = split_lds_bytes(a), lb = b ? split_lds_bytes(b) : 0;
  static size_t configured[64] = {};  // the LDS opt-in belongs to the (kernel, device) pair
  if (la + lb > configured[a->device & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_split_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(la + lb)) != hipSuccess)
      return fail("cannot reserve the LDS of the split-precision kernel");
    configured[a->device & 63] = la + lb;
  }
  MlpPair q{a->dP, b ? b->dP : nullptr, xa, xb, ya, yb, nullptr};
  SplitLds o{a->cols[0] * MT * 3, (int)(la / 2), (int)(la / 2) + (b ? b->cols[0] * MT * 3 : 0)};
  rl_act_epilogue off{};  // (num_envs = 0: no epilogue)
  hipLaunchKernelGGL(mlp_split_kernel<16>, dim3((n_rows + MT - 1) / MT), dim3(1024), la + lb, (hipStream_t)stream, q, n_rows, o, ep ? *ep : off);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(hipGetErrorString(e));
}
// host re-layout of one layer's parameters (nn.Linear [out][in]) into the two device images the kernels read, and the copy
int upload_layer(rl_mlp* m, int l, const float* weights, const float* biases) {
  if (!weights || !biases) return fail("null layer pointer");
  const int K = m->dims[l], N = m->dims[l + 1];
  const int KB = m->P.KB[l], NT = 8 * m->P.NT8[l], KB32 = m->P.KB32[l], NTS = m->P.NTS[l];
  std::vector<float> Wf((size_t)KB * NT * 64 * 4, 0.f), bp((size_t)NT * 16, 0.f);
  for (int n = 0; n < N; ++n) bp[n] = biases[n];
  for (int kb = 0; kb < KB; ++kb)
    for (int t = 0; t < NT; ++t)
      for (int ln = 0; ln < 64; ++ln)
        for (int sI = 0; sI < 4; ++sI) {
          const int k = kb * 16 + 4 * sI + (ln >> 4), n = t * 16 + (ln & 15);  // B[k][n] = W[n][k] (nn.Linear: [out][in])
          if (k < K && n < N) Wf[(((size_t)kb * NT + t) * 64 + ln) * 4 + sI] = weights[(size_t)n * K + k];
        }
  std::vector<uint16_t> Wsp((size_t)KB32 * NTS * 3 * 64 * 8, 0);
  for (int kb = 0; kb < KB32; ++kb)
    for (int t = 0; t < NTS; ++t)
      for (int ln = 0; ln < 64; ++ln)
        for (int j = 0; j < 8; ++j) {
          const int k = kb * 32 + 8 * (ln >> 4) + j, n = t * 16 + (ln & 15);
          if (k < K && n < N) {
            const Split3 sp = split3(weights[(size_t)n * K + k]);
            const size_t o = ((((size_t)kb * NTS + t) * 3) * 64 + ln) * 8 + j;
            Wsp[o] = sp.h; Wsp[o + 512] = sp.m; Wsp[o + 1024] = sp.l;
          }
        }
  if (hipMemcpy(const_cast<float*>(m->P.W[l]), Wf.data(), Wf.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(const_cast<float*>(m->P.b[l]), bp.data(), bp.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(const_cast<uint16_t*>(m->P.Ws[l]), Wsp.data(), Wsp.size() * 2, hipMemcpyHostToDevice) != hipSuccess)
    return fail("weight upload failed");
  return 0;
}
}  // namespace

extern "C" {

int rl_mlp_set_weights(rl_mlp* m, const float* const* weights, const float* const* biases, void* stream) {
  if (!m || !weights || !biases) return fail("null argument");
  if (hipSetDevice(m->device) != hipSuccess) return fail("hipSetDevice failed");
  // launches in flight read the old images - on the caller's stream or on any other (a captured graph replaying on a second stream):
  // let ALL of them finish (synchronising `stream` alone left a forward on another non-blocking stream reading
  // half-updated weights); the copies below are synchronous, and a forward launched after this call returns sees the new f32 AND
  // split-bf16 images, never a mix
  // A device-wide wait is illegal while a stream capture is open (it would fail AND invalidate the capture - ADVICE r4): refuse up front when
  // the caller's stream is capturing; parameters are pushed between replays of a collection graph, never inside its capture.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return fail("rl_mlp_set_weights inside a stream capture: the upload is a device-wide wait + synchronous copies; push parameters before hipStreamBeginCapture or between graph replays");
  if (hipDeviceSynchronize() != hipSuccess) return fail("device synchronisation failed");
  for (int l = 0; l < m->P.n_layers; ++l)
    if (upload_layer(m, l, weights[l], biases[l])) return -1;
  return 0;
}

int rl_mlp_create(const int32_t* dims, int32_t n_layers, int32_t activation, const float* const* weights, const float* const* biases,
                  int32_t device, rl_mlp** out) {
  if (!dims || !weights || !biases || !out) return fail("null argument");
  if (n_layers < 1 || n_layers > RL_MLP_MAX_LAYERS) return fail("unsupported layer count");
  for (int l = 0; l <= n_layers; ++l)
    if (dims[l] < 1 || dims[l] > RL_MLP_MAX_WIDTH) return fail("layer width out of range (1.." + std::to_string(RL_MLP_MAX_WIDTH) + ")");
  if (activation < RL_ACT_ELU || activation > RL_ACT_TANH) return fail("unknown activation");
  if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
  rl_mlp* m = new rl_mlp();
  m->device = device;
  memset(&m->P, 0, sizeof(m->P));
  m->P.n_layers = n_layers; m->P.act = activation; m->P.in_dim = dims[0]; m->P.out_dim = dims[n_layers];
  for (int l = 0; l < n_layers; ++l) {
    const int K = dims[l], N = dims[l + 1];
    m->dims[l] = K; m->dims[l + 1] = N;
    // layer l contracts over the 16-padded width of its input; its output is padded to 128 columns (groups of 8 tiles).
    // For l > 0 the input's padding columns were written as zeros by layer l-1 (128-padded >= 16-padded).
    const int KB = (K + 15) / 16, NT8 = (N + 127) / 128, NT = 8 * NT8;
    m->P.KB[l] = KB; m->P.N[l] = N; m->P.NT8[l] = NT8;
    // split-precision image: 32-deep k blocks; a non-last layer also computes the zero columns that pad its output to the next
    // layer's 32-deep blocks (never more tiles than the 128-column padding of the bias vector holds)
    const int KB32 = (K + 31) / 32, NTS = l + 1 < n_layers ? 2 * ((N + 31) / 32) : (N + 15) / 16;
    m->P.KB32[l] = KB32; m->P.NTS[l] = NTS;
    m->cols[l & 1] = std::max(m->cols[l & 1], KB32 * 32);
    void *dW = nullptr, *db = nullptr, *dWs = nullptr;
    if (hipMalloc(&dW, (size_t)KB * NT * 64 * 4 * 4) != hipSuccess || hipMalloc(&db, (size_t)NT * 16 * 4) != hipSuccess ||
        hipMalloc(&dWs, (size_t)KB32 * NTS * 3 * 64 * 8 * 2) != hipSuccess) {
      rl_mlp_destroy(m);
      return fail("device allocation failed");
    }
    m->allocs.push_back(dW); m->allocs.push_back(db); m->allocs.push_back(dWs);
    m->P.W[l] = (const float*)dW; m->P.b[l] = (const float*)db; m->P.Ws[l] = (const uint16_t*)dWs;
    if (upload_layer(m, l, weights[l], biases[l])) {
      rl_mlp_destroy(m);
      return -1;
    }
  }
  void* dP = nullptr;
  if (hipMalloc(&dP, sizeof(MlpParams)) != hipSuccess) {
    rl_mlp_destroy(m);
    return fail("device allocation failed");
  }
  m->allocs.push_back(dP);
  (void)hipMemcpy(dP, &m->P, sizeof(MlpParams), hipMemcpyHostToDevice);
  m->dP = (MlpParams*)dP;
  *out = m;
  return 0;
}

int rl_mlp_forward_pair(rl_mlp* a, const float* xa_dev, float* ya_dev, rl_mlp* b, const float* xb_dev, float* yb_dev, int32_t n_rows, void* stream) {
  if (!a || !b || !xa_dev || !ya_dev || !xb_dev || !yb_dev) return fail("null argument");
  if (n_rows <= 0) return 0;
  // fused (one workgroup = 16 rows of BOTH networks, balanced) once its n_rows / 16 workgroups fill the chip; below that one
  // network per workgroup (twice the workgroups).  RL_MLP_PAIR_MODE=fused|split overrides; RL_MLP_PAIR_RT=1|2 picks the split tile.
  static const int forced = [] { const char* e = getenv("RL_MLP_PAIR_RT"); return e ? atoi(e) : 0; }();
  static const int mode = [] { const char* e = getenv("RL_MLP_PAIR_MODE"); return !e ? 0 : (e[0] == 'f' ? 1 : 2); }();
  const bool fused = a->P.n_layers == b->P.n_layers && (mode == 1 || (mode == 0 && n_rows >= 3072));
  const int RT = forced == 1 || forced == 2 ? forced : (n_rows >= 4096 ? 2 : 1);
  const size_t lds = fused ? sizeof(float) * 4 * MT * KMAX : sizeof(float) * 2 * MT * KMAX * RT;
  if (a->device != b->device) return fail("the two networks live on different devices");
  if (hipSetDevice(a->device) != hipSuccess) return fail("hipSetDevice failed");
  if (split_wanted() && a->P.n_layers == b->P.n_layers && split_lds_bytes(a) + split_lds_bytes(b) <= LDS_MAX)
    return launch_split(a, xa_dev, ya_dev, b, xb_dev, yb_dev, n_rows, stream);
  static bool attr_done[64] = {};  // the LDS opt-in belongs to the (kernel, device) pair
  if (!attr_done[a->device & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_forward_pair_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * 2 * MT * KMAX)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_forward_pair_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * 4 * MT * KMAX)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_pair_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * 4 * MT * KMAX)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_pair_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(float) * 4 * MT * KMAX)) != hipSuccess)
      return fail("cannot reserve the LDS of the paired kernel");
    attr_done[a->device & 63] = true;
  }
  MlpPair q{a->dP, b->dP, xa_dev, xb_dev, ya_dev, yb_dev, nullptr};
#ifdef RL_MLP_CLOCK
  static unsigned long long* clk = nullptr;
  const int n_wg = (n_rows + MT - 1) / MT;
  if (!clk) (void)hipMalloc(&clk, sizeof(unsigned long long) * 256 * 8192);
  q.clk = clk;
#endif
  const int tiles = (n_rows + RT * MT - 1) / (RT * MT);
  static const int fw = [] { const char* e = getenv("RL_MLP_FUSED_WAVES"); return e ? atoi(e) : 16; }();
  if (fused && fw == 16)
    hipLaunchKernelGGL(mlp_fused_pair_kernel<16>, dim3((n_rows + MT - 1) / MT), dim3(1024), lds, (hipStream_t)stream, q, n_rows);
  else if (fused)
    hipLaunchKernelGGL(mlp_fused_pair_kernel<8>, dim3((n_rows + MT - 1) / MT), dim3(512), lds, (hipStream_t)stream, q, n_rows);
  else if (RT == 2)
    hipLaunchKernelGGL((mlp_forward_pair_kernel<2, 8>), dim3(2 * tiles), dim3(512), lds, (hipStream_t)stream, q, n_rows);
  else
    hipLaunchKernelGGL((mlp_forward_pair_kernel<1, 4>), dim3(2 * tiles), dim3(256), lds, (hipStream_t)stream, q, n_rows);
#ifdef RL_MLP_CLOCK
  static int calls = 0;
  if (fused && ++calls == 100) {  // one report: per-phase cycles of every wavefront, averaged over the workgroups
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)n_wg * 256);
    (void)hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    const int ns = 3 + 3 * a->P.n_layers, nw = fw == 16 ? 16 : 8;
    // per workgroup: span from the earliest first stamp to the latest last stamp of its wavefronts; then the phases per wavefront
    double span = 0;
    for (int g = 0; g < n_wg; ++g) {
      unsigned long long t0 = ~0ull, t1 = 0;
      for (int w = 0; w < nw; ++w) { t0 = std::min(t0, h[(size_t)g * 256 + w * 16]); t1 = std::max(t1, h[(size_t)g * 256 + w * 16 + ns - 1]); }
      span += (double)(t1 - t0);
    }
    printf("[mlp clock] %d workgroups x %d wavefronts, mean first stamp -> last stamp per workgroup: %.0f ticks\n", n_wg, nw, span / n_wg);
    for (int w = 0; w < nw; ++w) {
      printf("[mlp clock] wave %2d:", w);
      for (int i = 1; i < ns; ++i) {
        double s = 0;
        for (int g = 0; g < n_wg; ++g) s += (double)(h[(size_t)g * 256 + w * 16 + i] - h[(size_t)g * 256 + w * 16 + i - 1]);
        printf(" %6.0f", s / n_wg);
      }
      printf("\n");
    }
  }
#endif
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(hipGetErrorString(e));
}

// include/rl_policy.h: the pair launch with the rollout step's stochastic head in its epilogue - on the kernel the collection loop runs at
// rollout sizes (32 rows x one network per workgroup, split precision); any other size / precision reports 1 and launches nothing
int rl_mlp_forward_pair_act(rl_mlp* a, const float* xa_dev, float* ya_dev, rl_mlp* b, const float* xb_dev, float* yb_dev, int32_t n_rows,
                            const rl_act_epilogue* ep, void* stream) {
  if (!a || !b || !xa_dev || !ya_dev || !xb_dev || !yb_dev || !ep) return fail("null argument");
  if (n_rows <= 0) return 0;
  if (a->device != b->device) return fail("the two networks live on different devices");
  if (ep->num_envs != n_rows || ep->obs_dim != a->P.in_dim || ep->critic_dim != b->P.in_dim || ep->act_dim != a->P.out_dim || b->P.out_dim != 1)
    return fail("rl_mlp_forward_pair_act: the epilogue's sizes are not the networks' (rows, obs_dim, critic_dim, act_dim; the critic has one output)");
  // the split-precision pair kernels carry the epilogue (16 rows of both networks, or 32 rows of one, per workgroup: launch_split picks);
  // (act_dim + 3) / 4 Philox blocks per row must fit the workgroup's threads in both
  const bool split = split_wanted() && a->P.n_layers == b->P.n_layers && split_lds_bytes(a) + split_lds_bytes(b) <= LDS_MAX && (ep->act_dim + 3) / 4 * 32 <= 512;
  if (!split) return 1;
  if (hipSetDevice(a->device) != hipSuccess) return fail("hipSetDevice failed");
  return launch_split(a, xa_dev, ya_dev, b, xb_dev, yb_dev, n_rows, stream, ep);
}

int rl_mlp_forward(rl_mlp* m, const float* x_dev, float* y_dev, int32_t n_rows, void* stream) {
  if (!m || !x_dev || !y_dev) return fail("null argument");
  if (n_rows <= 0) return 0;
  constexpr size_t lds = sizeof(float) * 2 * MT * KMAX;
  if (hipSetDevice(m->device) != hipSuccess) return fail("hipSetDevice failed");
  if (split_wanted() && split_lds_bytes(m) <= LDS_MAX) return launch_split(m, x_dev, y_dev, nullptr, nullptr, nullptr, n_rows, stream);
  static bool attr_done[64] = {};
  if (!attr_done[m->device & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail("cannot reserve 64 KB of LDS");
    attr_done[m->device & 63] = true;
  }
  hipLaunchKernelGGL(mlp_forward_kernel, dim3((n_rows + MT - 1) / MT), dim3(256), lds, (hipStream_t)stream, m->P, x_dev, y_dev, n_rows);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(hipGetErrorString(e));
}

// include/rl_policy.h: the small-footprint launch (16 rows per workgroup of four wavefronts, 140 registers per lane, 64 KB of LDS)
int rl_mlp_forward_small(rl_mlp* m, const float* x_dev, float* y_dev, int32_t n_rows, void* stream) {
  if (!m || !x_dev || !y_dev) return fail("null argument");
  if (n_rows <= 0) return 0;
  constexpr size_t lds = sizeof(float) * 2 * MT * KMAX;
  if (hipSetDevice(m->device) != hipSuccess) return fail("hipSetDevice failed");
  static bool attr_done[64] = {};
  if (!attr_done[m->device & 63]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_forward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail("cannot reserve 64 KB of LDS");
    attr_done[m->device & 63] = true;
  }
  hipLaunchKernelGGL(mlp_forward_kernel, dim3((n_rows + MT - 1) / MT), dim3(256), lds, (hipStream_t)stream, m->P, x_dev, y_dev, n_rows);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(hipGetErrorString(e));
}

int32_t rl_mlp_in_dim(const rl_mlp* m) { return m ? m->P.in_dim : 0; }
int32_t rl_mlp_out_dim(const rl_mlp* m) { return m ? m->P.out_dim : 0; }

int rl_mlp_destroy(rl_mlp* m) {
  if (!m) return 0;
  (void)hipSetDevice(m->device);
  for (void* p : m->allocs) (void)hipFree(p);
  delete m;
  return 0;
}

const char* rl_mlp_last_error(void) { return err().c_str(); }

}  // extern "C"
