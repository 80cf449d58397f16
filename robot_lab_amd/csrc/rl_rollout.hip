// rl_rollout.hip - on-policy rollout storage for gfx950 (MI355X): stochastic policy head, transition record and
// GAE.  C-ABI: include/rl_rollout.h (which restates the rsl-rl-lib 3.0.1 arithmetic this follows).
//
// Everything here is HBM-bound byte / element work (per rollout step ~1.4 KB per env: 45 + 235 observation columns,
// 3 x 12 action-sized rows, 5 scalars), so the kernels are plain coalesced streams:
//   act      one launch: the first blocks copy the two observation batches into slot t as flat 16-byte streams, the
//            rest draw the actions (one thread per env x 4-action Philox block: 4 uniforms -> 4 normals by Box-Muller)
//            and reduce the log-probability of an env through LDS
//   record   one thread per env
//   returns  one thread per env walks its T values backwards ([T][N] arrays: every step is a coalesced row), block
//            partial sums feed the advantage normalisation; the partials are summed in a fixed order by every block
//            of the next kernel (no atomics: bit-reproducible)
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <string>

#include "../../include/rl_rollout.h"
#define RL_FN __host__ __device__ __forceinline__
#include "rl_math.h"
#include "rollout/rl_sample.h"

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_PARTIALS = 4096;

thread_local std::string g_err;
int fail(const std::string& m) {
  g_err = m;
  return -1;
}
#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) return fail(std::string(#x ": ") + hipGetErrorString(e_)); \
  } while (0)

__global__ void u32_kernel(uint32_t* p, uint32_t v, int add) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *p = add ? *p + v : v;
}

struct ActArgs {
  const float *obs, *critic_obs, *mean, *std, *values;
  float *actions_out, *s_obs, *s_critic, *s_actions, *s_mu, *s_sigma, *s_logp, *s_values;
  int N, obs_dim, critic_dim, act_dim;
  int copy_blocks_obs, copy_blocks_critic, sample_blocks;
  rl_act_epilogue ep;  // the sampling half (slot pointers, std, seed, counter: Philox counter of the launch = *counter_base + counter)
};

__device__ inline void stream_copy(const float* __restrict__ src, float* __restrict__ dst, size_t n, int block, int nblocks) {
  // n floats: float4 body + scalar tail when both ends are 16-byte aligned.  The destination is slot t of a [T][N][dim]
  // array, i.e. base + t * N * dim floats - only 4-byte aligned when N * dim is not a multiple of 4 (N = 37, dim = 45): scalar copy
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) != 0) {
    for (size_t i = (size_t)block * BLOCK + threadIdx.x; i < n; i += (size_t)nblocks * BLOCK) dst[i] = src[i];
    return;
  }
  const size_t n4 = n >> 2;
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(dst);
  for (size_t i = (size_t)block * BLOCK + threadIdx.x; i < n4; i += (size_t)nblocks * BLOCK) d4[i] = s4[i];
  if (block == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = src[(n4 << 2) + threadIdx.x];
}

__global__ __launch_bounds__(BLOCK) void act_kernel(ActArgs a) {
  int b = blockIdx.x;
#ifdef RL_ACT_NOCOPY  // analysis: what the two observation copies cost
  if (b < a.copy_blocks_obs + a.copy_blocks_critic) return;
#endif
  if (b < a.copy_blocks_obs) {
    stream_copy(a.obs, a.s_obs, (size_t)a.N * a.obs_dim, b, a.copy_blocks_obs);
    return;
  }
  b -= a.copy_blocks_obs;
  if (b < a.copy_blocks_critic) {  // (0 blocks when critic_obs is NULL: rl_rollout_store_critic_obs copies the row on the critic's stream)
    stream_copy(a.critic_obs, a.s_critic, (size_t)a.N * a.critic_dim, b, a.copy_blocks_critic);
    return;
  }
  b -= a.copy_blocks_critic;
  // one thread per (env, Philox block of 4 actions): a block of 256 threads covers 256 / nblk envs; the log-probability
  // partials of an env meet in LDS (one thread per env alone left 16 workgroups walking 3 Philox blocks each: 9.7 us)
  __shared__ float part[BLOCK];
  const int A = a.act_dim, nblk = (A + 3) >> 2, epb = BLOCK / nblk;
  const int el = threadIdx.x / nblk, blk = threadIdx.x - el * nblk;
  const int e = b * epb + el;
  const bool live = el < epb && e < a.N;
  float logp = 0.f;
  if (live) logp = rl::act_block(a.ep, a.mean, e, blk);  // (csrc/rl_sample.h: shared with the actor launch's epilogue, include/rl_act.h)
  part[threadIdx.x] = logp;
  __syncthreads();
  if (live && blk == 0) {
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += part[threadIdx.x + i];  // fixed order: blocks 0, 1, ...
    a.s_logp[e] = s;
    if (a.values != nullptr) a.s_values[e] = a.values[e];  // (NULL: the critic writes the slot itself, rl_rollout_values_slot)
  }
}

__global__ __launch_bounds__(BLOCK) void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  stream_copy(src, dst, n, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(BLOCK) void record_kernel(const float* __restrict__ rewards, const uint8_t* __restrict__ terminated,
                                                       const uint8_t* __restrict__ time_outs, const float* __restrict__ s_values,
                                                       float* __restrict__ s_rewards, uint8_t* __restrict__ s_dones, float gamma, int N) {
  const int e = blockIdx.x * BLOCK + threadIdx.x;
  if (e >= N) return;
  const bool to = time_outs[e] != 0;
  s_rewards[e] = to ? fmaf(gamma, s_values[e], rewards[e]) : rewards[e];  // bootstrapping on time outs (one explicit fma in all three places that do it)
  s_dones[e] = (terminated[e] != 0 || to) ? 1 : 0;
}

__device__ inline double block_sum(double v, double* sh) {
  // fixed-order tree over the 256 threads
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int s = BLOCK / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(BLOCK) void gae_kernel(float* __restrict__ rewards, const float* __restrict__ values,
                                                    uint8_t* __restrict__ dones, const float* __restrict__ last_values,
                                                    float* __restrict__ returns, float* __restrict__ adv, double* __restrict__ partial, float gamma,
                                                    float lam, int T, int N) {
  __shared__ double sh[BLOCK];
  const int e = blockIdx.x * BLOCK + threadIdx.x;
  double sum = 0.0;
  if (e < N) {
    float a = 0.f, next_v = last_values[e];
    for (int t = T - 1; t >= 0; --t) {
      const size_t i = (size_t)t * N + e;
      const float v = values[i];
      const uint8_t d = dones[i];
      float r = rewards[i];
      if (d & 2) {  // deferred bootstrap on a time out (rl_env_step_record with values NULL): the same expression record_kernel / the env kernel evaluate
        r = fmaf(gamma, v, r);
        rewards[i] = r;
        dones[i] = 1;
      }
      const float nt = d ? 0.f : 1.f;
      const float delta = r + nt * gamma * next_v - v;
      a = delta + nt * gamma * lam * a;
      const float ret = a + v;
      returns[i] = ret;
      const float ad = ret - v;  // "advantages = returns - values", not the running a (they differ by one rounding)
      adv[i] = ad;
      sum += (double)ad;
      next_v = v;
    }
  }
  const double s = block_sum(sum, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__device__ inline double sum_partials(const double* __restrict__ p, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += p[i];  // same order in every thread of every block
  return s;
}

__global__ __launch_bounds__(BLOCK) void var_kernel(const float* __restrict__ adv, const double* __restrict__ partial_sum, double* __restrict__ partial_sq,
                                                    int nb_gae, size_t count) {
  __shared__ double sh[BLOCK];
  const double mean = sum_partials(partial_sum, nb_gae) / (double)count;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < count; i += (size_t)gridDim.x * BLOCK) {
    const double d = (double)adv[i] - mean;
    acc += d * d;
  }
  const double s = block_sum(acc, sh);
  if (threadIdx.x == 0) partial_sq[blockIdx.x] = s;
}

__global__ __launch_bounds__(BLOCK) void norm_kernel(float* __restrict__ adv, const double* __restrict__ partial_sum, const double* __restrict__ partial_sq,
                                                     int nb_gae, int nb_var, size_t count) {
  const double mean = sum_partials(partial_sum, nb_gae) / (double)count;
  const double var = sum_partials(partial_sq, nb_var) / (double)(count > 1 ? count - 1 : 1);  // torch.std: unbiased
  const float m = (float)mean, inv = 1.0f / ((float)sqrt(var) + 1e-8f);
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < count; i += (size_t)gridDim.x * BLOCK) adv[i] = (adv[i] - m) * inv;
}

// out[s * n_rows + r][c] = sign[s][c] * in[r][perm[s][c]]: blockIdx.y = copy s, one thread per element of a copy (32-bit
// index math); the rows of `in` are re-read by the n_sym copies out of L2, writes are fully coalesced
__global__ __launch_bounds__(BLOCK) void symmetry_kernel(const float* __restrict__ in, float* __restrict__ out, const int32_t* __restrict__ perm,
                                                         const float* __restrict__ sign, uint32_t per, uint32_t dim) {
  const uint32_t s = blockIdx.y;
  const int32_t* __restrict__ pm = perm + s * dim;
  const float* __restrict__ sg = sign + s * dim;
  float* __restrict__ o = out + (size_t)s * per;
  for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < per; i += gridDim.x * BLOCK) {
    const uint32_t r = i / dim, c = i - r * dim;
    o[i] = sg[c] * in[r * dim + (uint32_t)pm[c]];
  }
}

}  // namespace

struct rl_symmetry {
  int n_sym, dim, device;
  int32_t* perm = nullptr;
  float* sign = nullptr;
};

struct rl_rollout {
  int N, T, obs_dim, critic_dim, act_dim, device;
  uint64_t seed;
  uint32_t counter = 0;  // transitions drawn so far (Philox counter)
  int step = 0;
  bool acted = false;    // rl_rollout_act done for the current step, rl_rollout_record pending
  void* buf[RL_RO_NUM_BUFFERS] = {};
  int64_t count[RL_RO_NUM_BUFFERS] = {};
  double *partial_sum = nullptr, *partial_sq = nullptr;
  // hipGraph capture (rl_rollout_graph_*): the act kernel takes the counter as *counter_base + literal; `anchor` mirrors the word
  uint32_t* counter_base = nullptr;
  uint32_t anchor = 0;
  bool capturing = false;
  uint32_t snap_counter = 0, graph_n = 0;
  int snap_step = 0, end_step = 0;
  bool snap_acted = false;
};

namespace {
int dim_of(const rl_rollout* r, int which) {
  switch (which) {
    case RL_RO_OBS: return r->obs_dim;
    case RL_RO_CRITIC_OBS: return r->critic_dim;
    case RL_RO_ACTIONS: case RL_RO_MU: case RL_RO_SIGMA: return r->act_dim;
    default: return 1;
  }
}
template <class Tp>
Tp* slot(const rl_rollout* r, int which, int t) {
  return static_cast<Tp*>(r->buf[which]) + (size_t)t * r->N * dim_of(r, which);
}
int blocks_for(size_t n) { return (int)((n + BLOCK - 1) / BLOCK); }
}  // namespace

extern "C" {

const char* rl_rollout_last_error(void) { return g_err.c_str(); }

int rl_rollout_create(int32_t num_envs, int32_t num_steps, int32_t obs_dim, int32_t critic_dim, int32_t act_dim, uint64_t seed, int32_t device,
                      rl_rollout** out) {
  if (!out) return fail("out is NULL");
  if (num_envs < 1 || num_steps < 1 || obs_dim < 1 || critic_dim < 1 || act_dim < 1) return fail("sizes must be positive");
  if (act_dim > 4 * BLOCK) return fail("act_dim above 1024");
  if (blocks_for((size_t)num_envs) > MAX_PARTIALS) return fail("num_envs too large for the reduction scratch");
  HIP_OK(hipSetDevice(device));
  rl_rollout* r = new rl_rollout();
  r->N = num_envs; r->T = num_steps; r->obs_dim = obs_dim; r->critic_dim = critic_dim; r->act_dim = act_dim; r->device = device; r->seed = seed;
  for (int w = 0; w < RL_RO_NUM_BUFFERS; ++w) {
    r->count[w] = (int64_t)num_steps * num_envs * dim_of(r, w);
    const size_t bytes = (size_t)r->count[w] * (w == RL_RO_DONES ? 1 : 4);
    if (hipMalloc(&r->buf[w], bytes) != hipSuccess || hipMemset(r->buf[w], 0, bytes) != hipSuccess) {
      rl_rollout_destroy(r);
      return fail("hipMalloc of the rollout storage failed");
    }
  }
  if (hipMalloc(&r->partial_sum, MAX_PARTIALS * sizeof(double)) != hipSuccess || hipMalloc(&r->partial_sq, MAX_PARTIALS * sizeof(double)) != hipSuccess) {
    rl_rollout_destroy(r);
    return fail("hipMalloc of the reduction scratch failed");
  }
  if (hipMalloc(&r->counter_base, 4 * sizeof(uint32_t)) != hipSuccess || hipMemset(r->counter_base, 0, 4 * sizeof(uint32_t)) != hipSuccess) {
    rl_rollout_destroy(r);
    return fail("hipMalloc of the counter word failed");
  }
  HIP_OK(hipDeviceSynchronize());  // the memsets ran on the null stream; callers launch on non-blocking streams
  *out = r;
  return 0;
}

static void fill_epilogue(rl_rollout* r, const float* std, float* actions_out, float clip, rl_act_epilogue* e) {
  const int t = r->step;
  e->actions_out = actions_out;
  e->s_obs = slot<float>(r, RL_RO_OBS, t); e->s_critic_obs = slot<float>(r, RL_RO_CRITIC_OBS, t); e->s_actions = slot<float>(r, RL_RO_ACTIONS, t);
  e->s_mu = slot<float>(r, RL_RO_MU, t); e->s_sigma = slot<float>(r, RL_RO_SIGMA, t); e->s_logp = slot<float>(r, RL_RO_LOG_PROB, t);
  e->s_values = slot<float>(r, RL_RO_VALUES, t);
  e->std = std; e->counter_base = r->counter_base; e->seed = r->seed; e->counter = r->counter - r->anchor;
  e->num_envs = r->N; e->obs_dim = r->obs_dim; e->critic_dim = r->critic_dim; e->act_dim = r->act_dim; e->clip = clip;
}

// include/rl_rollout.h: the current step's act as a descriptor for the actor / critic launch (include/rl_act.h) - no launch, no state change;
// rl_rollout_act_done marks the step as acted once that launch is enqueued
int rl_rollout_act_epilogue(rl_rollout* r, const float* std, float* actions_out, float clip, rl_act_epilogue* out) {
  if (!r || !std || !actions_out || !out) return fail("NULL argument");
  if (r->step >= r->T) return fail("rollout storage overflow: call rl_rollout_clear after num_steps transitions");
  if (r->acted) return fail("rl_rollout_act_epilogue: the step is already acted (rl_rollout_record closes it)");
  fill_epilogue(r, std, actions_out, clip, out);
  return 0;
}
int rl_rollout_act_done(rl_rollout* r) {
  if (!r) return fail("NULL argument");
  if (r->acted) return fail("rl_rollout_act_done called twice without rl_rollout_record");
  r->acted = true;
  return 0;
}

int rl_rollout_act(rl_rollout* r, const float* obs, const float* critic_obs, const float* mean, const float* std, const float* values,
                   float* actions_out, void* stream) {
  if (!r || !obs || !mean || !std || !actions_out) return fail("NULL argument");  // (critic_obs / values may be NULL: include/rl_rollout.h)
  if (r->step >= r->T) return fail("rollout storage overflow: call rl_rollout_clear after num_steps transitions");
  if (r->acted) return fail("rl_rollout_act called twice without rl_rollout_record");
  HIP_OK(hipSetDevice(r->device));
  ActArgs a;
  const int t = r->step;
  a.obs = obs; a.critic_obs = critic_obs; a.mean = mean; a.std = std; a.values = values; a.actions_out = actions_out;
  a.s_obs = slot<float>(r, RL_RO_OBS, t); a.s_critic = slot<float>(r, RL_RO_CRITIC_OBS, t); a.s_actions = slot<float>(r, RL_RO_ACTIONS, t);
  a.s_mu = slot<float>(r, RL_RO_MU, t); a.s_sigma = slot<float>(r, RL_RO_SIGMA, t); a.s_logp = slot<float>(r, RL_RO_LOG_PROB, t);
  a.s_values = slot<float>(r, RL_RO_VALUES, t);
  a.N = r->N; a.obs_dim = r->obs_dim; a.critic_dim = r->critic_dim; a.act_dim = r->act_dim;
  fill_epilogue(r, std, actions_out, -1.f, &a.ep);
  // 4 float4 per thread of the copy blocks
  a.copy_blocks_obs = std::max(1, blocks_for(((size_t)r->N * r->obs_dim) >> 4));
  a.copy_blocks_critic = critic_obs ? std::max(1, blocks_for(((size_t)r->N * r->critic_dim) >> 4)) : 0;
  const int envs_per_block = BLOCK / ((r->act_dim + 3) / 4);
  a.sample_blocks = (r->N + envs_per_block - 1) / envs_per_block;
  hipLaunchKernelGGL(act_kernel, dim3(a.copy_blocks_obs + a.copy_blocks_critic + a.sample_blocks), dim3(BLOCK), 0, (hipStream_t)stream, a);
  HIP_OK(hipGetLastError());
  r->acted = true;
  return 0;
}

int rl_rollout_record(rl_rollout* r, const float* rewards, const uint8_t* terminated, const uint8_t* time_outs, float gamma, void* stream) {
  if (!r || !rewards || !terminated || !time_outs) return fail("NULL argument");
  if (!r->acted) return fail("rl_rollout_record needs rl_rollout_act for this step first");
  HIP_OK(hipSetDevice(r->device));
  const int t = r->step;
  hipLaunchKernelGGL(record_kernel, dim3(blocks_for((size_t)r->N)), dim3(BLOCK), 0, (hipStream_t)stream, rewards, terminated, time_outs,
                     slot<float>(r, RL_RO_VALUES, t), slot<float>(r, RL_RO_REWARDS, t), slot<uint8_t>(r, RL_RO_DONES, t), gamma, r->N);
  HIP_OK(hipGetLastError());
  r->acted = false;
  r->step += 1;
  r->counter += 1;
  return 0;
}

int rl_rollout_values_slot(rl_rollout* r, float** values) {
  if (!r || !values) return fail("NULL argument");
  if (r->step >= r->T) return fail("rollout storage overflow: call rl_rollout_clear after num_steps transitions");
  *values = slot<float>(r, RL_RO_VALUES, r->step);
  return 0;
}

int rl_rollout_store_critic_obs(rl_rollout* r, const float* critic_obs, void* stream) {
  if (!r || !critic_obs) return fail("NULL argument");
  if (r->step >= r->T) return fail("rollout storage overflow: call rl_rollout_clear after num_steps transitions");
  HIP_OK(hipSetDevice(r->device));
  const size_t n = (size_t)r->N * r->critic_dim;
  hipLaunchKernelGGL(copy_kernel, dim3(std::max(1, blocks_for(n >> 4))), dim3(BLOCK), 0, (hipStream_t)stream, critic_obs, slot<float>(r, RL_RO_CRITIC_OBS, r->step), n);
  HIP_OK(hipGetLastError());
  return 0;
}

int rl_rollout_record_slots(rl_rollout* r, const float** values, float** rewards, uint8_t** dones) {
  if (!r || !values || !rewards || !dones) return fail("NULL argument");
  if (!r->acted) return fail("rl_rollout_record_slots needs rl_rollout_act for this step first");
  const int t = r->step;
  *values = slot<float>(r, RL_RO_VALUES, t);
  *rewards = slot<float>(r, RL_RO_REWARDS, t);
  *dones = slot<uint8_t>(r, RL_RO_DONES, t);
  r->acted = false;
  r->step += 1;
  r->counter += 1;
  return 0;
}

int rl_rollout_compute_returns(rl_rollout* r, const float* last_values, float gamma, float lam, int32_t normalize_advantage, void* stream) {
  if (!r || !last_values) return fail("NULL argument");
  if (r->step != r->T || r->acted) return fail("compute_returns needs a full storage (num_steps recorded transitions)");
  HIP_OK(hipSetDevice(r->device));
  hipStream_t s = (hipStream_t)stream;
  const int nb = blocks_for((size_t)r->N);
  const size_t count = (size_t)r->T * r->N;
  hipLaunchKernelGGL(gae_kernel, dim3(nb), dim3(BLOCK), 0, s, (float*)r->buf[RL_RO_REWARDS], (const float*)r->buf[RL_RO_VALUES],
                     (uint8_t*)r->buf[RL_RO_DONES], last_values, (float*)r->buf[RL_RO_RETURNS], (float*)r->buf[RL_RO_ADVANTAGES], r->partial_sum,
                     gamma, lam, r->T, r->N);
  if (normalize_advantage) {
    const int nbv = std::min(256, std::max(1, blocks_for(count >> 2)));  // every thread re-sums the partials: keep them few
    hipLaunchKernelGGL(var_kernel, dim3(nbv), dim3(BLOCK), 0, s, (const float*)r->buf[RL_RO_ADVANTAGES], r->partial_sum, r->partial_sq, nb, count);
    hipLaunchKernelGGL(norm_kernel, dim3(nbv), dim3(BLOCK), 0, s, (float*)r->buf[RL_RO_ADVANTAGES], r->partial_sum, r->partial_sq, nb, nbv, count);
  }
  HIP_OK(hipGetLastError());
  return 0;
}

int rl_rollout_clear(rl_rollout* r) {
  if (!r) return fail("NULL handle");
  r->step = 0;
  r->acted = false;
  return 0;
}

int rl_rollout_get_buffer(rl_rollout* r, int32_t which, void** dev_ptr, int64_t* count) {
  if (!r || which < 0 || which >= RL_RO_NUM_BUFFERS) return fail("unknown buffer");
  if (dev_ptr) *dev_ptr = r->buf[which];
  if (count) *count = r->count[which];
  return 0;
}

int32_t rl_rollout_step(const rl_rollout* r) { return r ? r->step : -1; }

// include/rl_rollout.h "hipGraph capture"
int rl_rollout_graph_begin(rl_rollout* r, void* stream) {
  if (!r) return fail("NULL handle");
  if (r->capturing) return fail("rl_rollout_graph_begin: a capture is already open");
  HIP_OK(hipSetDevice(r->device));
  hipLaunchKernelGGL(u32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, r->counter_base, r->counter, 0);
  HIP_OK(hipGetLastError());
  r->anchor = r->counter;
  r->snap_counter = r->counter; r->snap_step = r->step; r->snap_acted = r->acted;
  r->capturing = true;
  return 0;
}

int rl_rollout_graph_end(rl_rollout* r, void* stream) {
  if (!r) return fail("NULL handle");
  if (!r->capturing) return fail("rl_rollout_graph_end without rl_rollout_graph_begin");
  r->capturing = false;
  const uint32_t n = r->counter - r->snap_counter;
  r->end_step = r->step;
  const bool open_step = r->acted;
  r->counter = r->snap_counter; r->step = r->snap_step; r->acted = r->snap_acted;  // capturing executed nothing
  if (open_step) return fail("the captured loop ends between rl_rollout_act and the record of that step");
  HIP_OK(hipSetDevice(r->device));
  hipLaunchKernelGGL(u32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, r->counter_base, n, 1);
  HIP_OK(hipGetLastError());
  r->graph_n = n;
  return (int)n;
}

int rl_rollout_graph_launching(rl_rollout* r, void* stream) {
  if (!r) return fail("NULL handle");
  if (r->capturing) return fail("rl_rollout_graph_launching inside a capture");
  if (r->acted) return fail("rl_rollout_graph_launching between rl_rollout_act and the record of that step");
  HIP_OK(hipSetDevice(r->device));
  if (r->counter != r->anchor) {  // transitions drawn directly since the last replay: re-anchor
    hipLaunchKernelGGL(u32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, r->counter_base, r->counter, 0);
    HIP_OK(hipGetLastError());
    r->anchor = r->counter;
  }
  r->counter += r->graph_n; r->anchor += r->graph_n;  // what the replay's last node does to the device word
  r->step = r->end_step;
  return 0;
}

int rl_symmetry_create(int32_t n_sym, int32_t dim, const int32_t* perm, const float* sign, int32_t device, rl_symmetry** out) {
  if (!perm || !sign || !out) return fail("NULL argument");
  if (n_sym < 1 || dim < 1) return fail("sizes must be positive");
  for (int i = 0; i < n_sym * dim; ++i)
    if (perm[i] < 0 || perm[i] >= dim) return fail("perm entry out of range");
  HIP_OK(hipSetDevice(device));
  rl_symmetry* s = new rl_symmetry{n_sym, dim, device};
  if (hipMalloc(&s->perm, sizeof(int32_t) * n_sym * dim) != hipSuccess || hipMalloc(&s->sign, sizeof(float) * n_sym * dim) != hipSuccess) {
    rl_symmetry_destroy(s);
    return fail("hipMalloc of the symmetry tables failed");
  }
  HIP_OK(hipMemcpy(s->perm, perm, sizeof(int32_t) * n_sym * dim, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(s->sign, sign, sizeof(float) * n_sym * dim, hipMemcpyHostToDevice));
  *out = s;
  return 0;
}

int rl_symmetry_apply(rl_symmetry* s, const float* in_dev, float* out_dev, int32_t n_rows, void* stream) {
  if (!s || !in_dev || !out_dev) return fail("NULL argument");
  if (n_rows <= 0) return 0;
  HIP_OK(hipSetDevice(s->device));
  const size_t per = (size_t)n_rows * s->dim;
  if (per >= (1ull << 31)) return fail("batch too large for the 32-bit index math of the symmetry kernel");
  const int blocks = (int)std::min<size_t>((per + BLOCK - 1) / BLOCK, 4096);
  hipLaunchKernelGGL(symmetry_kernel, dim3(blocks, s->n_sym), dim3(BLOCK), 0, (hipStream_t)stream, in_dev, out_dev, s->perm, s->sign, (uint32_t)per,
                     (uint32_t)s->dim);
  HIP_OK(hipGetLastError());
  return 0;
}

int rl_symmetry_destroy(rl_symmetry* s) {
  if (!s) return 0;
  if (s->perm) (void)hipFree(s->perm);
  if (s->sign) (void)hipFree(s->sign);
  delete s;
  return 0;
}

int rl_rollout_destroy(rl_rollout* r) {
  if (!r) return 0;
  for (void* p : r->buf)
    if (p) (void)hipFree(p);
  if (r->partial_sum) (void)hipFree(r->partial_sum);
  if (r->partial_sq) (void)hipFree(r->partial_sq);
  if (r->counter_base) (void)hipFree(r->counter_base);
  delete r;
  return 0;
}

}  // extern "C"
