// rl_sample.h - the stochastic policy head's arithmetic, ONE definition for the two kernels that run it: act_kernel (rl_rollout.hip) and the
// epilogue of the actor / critic launch (rl_policy.hip, include/rl_act.h).  Device code; needs rl_math.h (Philox4x32-10) before it.
#pragma once
#include "../../../include/rl_act.h"

namespace rl {

constexpr uint32_t STREAM_POLICY = 7;  // Philox stream of the action noise (the env uses streams 1..6)

// standard normals of one Philox block: (u0, u1) and (u2, u3) -> two Box-Muller pairs; u in [0, 1) -> 1 - u in (0, 1]
__device__ inline void normal4(uint64_t seed, uint32_t env, uint32_t counter, uint32_t blk, float (&z)[4]) {
  float u[4];
  uniform01x4(seed, env, counter, STREAM_POLICY, blk, u);
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float r = sqrtf(-2.0f * logf(1.0f - u[2 * p]));
    float s, c;
    sincosf(6.28318530717958647692f * u[2 * p + 1], &s, &c);
    z[2 * p] = r * c;
    z[2 * p + 1] = r * s;
  }
}

// env e, Philox block blk (actions 4 blk .. 4 blk + 3): sample, write actions_out and the slot's actions / mu / sigma; returns the block's
// share of the log-probability (the caller sums an env's blocks in ascending order)
__device__ inline float act_block(const rl_act_epilogue& a, const float* __restrict__ mean, int e, int blk) {
  const int A = a.act_dim;
  const float* mu = mean + (size_t)e * A;
  float z[4];
  normal4(a.seed, (uint32_t)e, *a.counter_base + a.counter, (uint32_t)blk, z);
  float logp = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = 4 * blk + i;
    if (j >= A) break;
    const float m = mu[j], sd = a.std[j];
    const float act = fmaf(sd, z[i], m);
    const float q = (act - m) / sd;  // as the reference evaluates Normal.log_prob on the sampled action
    logp += -0.5f * q * q - logf(sd) - 0.91893853320467274178f;
    const size_t o = (size_t)e * A + j;
    a.actions_out[o] = a.clip >= 0.f ? fminf(fmaxf(act, -a.clip), a.clip) : act;
    a.s_actions[o] = act;
    a.s_mu[o] = m;
    a.s_sigma[o] = sd;
  }
  return logp;
}

}  // namespace rl
