/* rl_act.h - the stochastic policy head of a rollout step as an EPILOGUE DESCRIPTOR: what `rl_rollout_act` (include/rl_rollout.h) does in a
 * launch of its own - a = mu + sigma * eps, log-prob, slot t of the storage - handed to the kernel that produces mu, so that the step needs
 * no launch between the actor and env.step (rsl_rl PPO.act: `self.transition.actions = self.policy.act(obs).detach()` ... right behind the
 * actor's forward; scripts/reinforcement_learning/rsl_rl/train.py:224 -> OnPolicyRunner.learn).
 *   rl_rollout_act_epilogue (include/rl_rollout.h) fills it for the current step of a storage;
 *   rl_mlp_forward_pair_act (include/rl_policy.h) consumes it in the actor / critic launch.
 * Same numbers as rl_rollout_act, bit for bit (one shared device function: csrc/rl_sample.h).  All pointers are DEVICE pointers. */
#ifndef RL_ACT_H
#define RL_ACT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rl_act_epilogue {
  float* actions_out;          /* [N][act_dim]: what env.step consumes (clamped to +-clip when clip >= 0; the storage keeps the sample) */
  float* s_obs;                /* slot t of the storage: [N][obs_dim] */
  float* s_critic_obs;         /* [N][critic_dim] */
  float* s_actions;            /* [N][act_dim] */
  float* s_mu;                 /* [N][act_dim] */
  float* s_sigma;              /* [N][act_dim] */
  float* s_logp;               /* [N] */
  float* s_values;             /* [N]: hand it to the critic as its output */
  const float* std;            /* [act_dim] */
  const uint32_t* counter_base; /* Philox counter of the launch = *counter_base + counter (include/rl_rollout.h, graph capture) */
  uint64_t seed;
  uint32_t counter;
  int32_t num_envs, obs_dim, critic_dim, act_dim;
  float clip;                  /* < 0: none */
} rl_act_epilogue;

#ifdef __cplusplus
}
#endif
#endif
