/* rl_rollout.h - C-ABI of the on-policy rollout storage: stochastic policy head, transition record, GAE
 * (second half: what sits between `policy(obs)` and `env.step()` in the rollout loop).
 *
 * Replaces (reference call sites; the arithmetic lives in the third-party `rsl-rl-lib`, pinned to 3.0.1 by
 * scripts/reinforcement_learning/rsl_rl/train.py:65-75 and absent from the reference tree):
 *   - `runner.learn(num_learning_iterations=..., init_at_random_ep_len=True)`, train.py:224 - per iteration
 *        `num_steps_per_env` (= 24, .../unitree_a1/agents/rsl_rl_ppo_cfg.py:11) times
 *            actions = alg.act(obs)                      -> rl_rollout_act
 *            obs, rewards, dones, extras = env.step(...)    (include/rl_env.h)
 *            alg.process_env_step(obs, rewards, dones, extras) -> rl_rollout_record
 *        then  alg.compute_returns(obs)                  -> rl_rollout_compute_returns
 *   - `gamma=0.99, lam=0.95`, rsl_rl_ppo_cfg.py:33-34; `init_noise_std=1.0`, :16 (the `std` vector passed to act)
 *
 * Published algorithm restated (rsl_rl/algorithms/ppo.py `act` / `process_env_step`, rsl_rl/storage/rollout_storage.py
 * `add_transitions` / `compute_returns`):
 *   act:      a = mu + sigma * eps, eps ~ N(0, 1);  log_prob = sum_j( -((a_j - mu_j)/sigma_j)^2 / 2 - log sigma_j - log sqrt(2 pi) )
 *   record:   r_t += gamma * V_t * time_out  (bootstrapping on time outs);  done_t = terminated | time_out
 *             (deferred form: a producer that has no V_t yet - rl_env_step_record with values NULL - stores the raw reward and marks
 *              the time out in bit 1 of the done byte; compute_returns adds gamma * V_t there and clears the bit: same numbers)
 *   returns:  for t = T-1 .. 0:  nt = 1 - done_t;  delta = r_t + nt * gamma * V_{t+1} - V_t;
 *             A = delta + nt * gamma * lam * A;  R_t = A + V_t;   adv = R - V;
 *             normalize: adv = (adv - mean(adv)) / (std(adv) + 1e-8)   (unbiased std over all T * N entries)
 * The random stream is this library's own (Philox4x32-10 keyed by the seed, counter = transitions recorded so far,
 * Box-Muller): torch's generator is not reproduced.
 *
 * All array arguments are DEVICE pointers (fp32 unless noted); work is stream-ordered on `stream` (a hipStream_t, or
 * NULL for the default stream).  Storage layout: [num_steps][num_envs][dim], i.e. a flat [T * N, dim] batch. */
#ifndef RL_ROLLOUT_H
#define RL_ROLLOUT_H

#include <stdint.h>

#include "rl_act.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rl_rollout rl_rollout;

enum rl_rollout_buffer {
  RL_RO_OBS = 0,        /* f32 [T][N][obs_dim] */
  RL_RO_CRITIC_OBS = 1, /* f32 [T][N][critic_dim] */
  RL_RO_ACTIONS = 2,    /* f32 [T][N][act_dim] */
  RL_RO_MU = 3,         /* f32 [T][N][act_dim] */
  RL_RO_SIGMA = 4,      /* f32 [T][N][act_dim] */
  RL_RO_LOG_PROB = 5,   /* f32 [T][N] */
  RL_RO_VALUES = 6,     /* f32 [T][N] */
  RL_RO_REWARDS = 7,    /* f32 [T][N] */
  RL_RO_DONES = 8,      /* u8  [T][N] */
  RL_RO_RETURNS = 9,    /* f32 [T][N] */
  RL_RO_ADVANTAGES = 10,/* f32 [T][N] */
  RL_RO_NUM_BUFFERS
};

int rl_rollout_create(int32_t num_envs, int32_t num_steps, int32_t obs_dim, int32_t critic_dim, int32_t act_dim, uint64_t seed,
                      int32_t device, rl_rollout** out);

/* PPO.act + the first half of RolloutStorage.add_transitions, for the current step t:
 *   obs [N][obs_dim], critic_obs [N][critic_dim], mean [N][act_dim] (actor output), std [act_dim] (the policy's noise std),
 *   values [N] (critic output)  ->  actions_out [N][act_dim] (what env.step consumes); everything is copied into slot t.
 * Fails when the storage is full (num_steps transitions recorded and not cleared).
 * `values` and `critic_obs` may each be NULL: that part of slot t is then left to rl_rollout_values_slot / rl_rollout_store_critic_obs,
 * i.e. to a critic that runs on ANOTHER stream while this stream goes on to env.step (the actor gates the env step, the critic does not:
 * its value is needed by the storage, the time-out bootstrap and GAE only). */
int rl_rollout_act(rl_rollout* r, const float* obs, const float* critic_obs, const float* mean, const float* std, const float* values,
                   float* actions_out, void* stream);

/* rl_rollout_act WITHOUT a launch of its own: the current step's sampling, log-prob and slot addresses as a descriptor (include/rl_act.h)
 * for the launch that produces `mean` - rl_mlp_forward_pair_act (include/rl_policy.h) samples in the actor's epilogue, copies obs / critic_obs
 * into the slot and lets the critic write V straight into `out->s_values`.  `clip` >= 0: actions_out is clamped to +-clip (RslRlVecEnvWrapper's
 * clip_actions: the env sees the clamped action, the storage keeps the sample).  No state changes here; rl_rollout_act_done marks the step as
 * acted once that launch is enqueued (then rl_rollout_record / rl_rollout_record_slots close it as after rl_rollout_act).  Same numbers as
 * rl_rollout_act bit for bit. */
int rl_rollout_act_epilogue(rl_rollout* r, const float* std, float* actions_out, float clip, rl_act_epilogue* out);
int rl_rollout_act_done(rl_rollout* r);

/* PPO.process_env_step + the second half of add_transitions: rewards [N] f32, terminated / time_outs [N] u8 (the
 * env's RL_BUF_TERMINATED / RL_BUF_TIME_OUT); closes step t and advances to t + 1. */
int rl_rollout_record(rl_rollout* r, const float* rewards, const uint8_t* terminated, const uint8_t* time_outs, float gamma, void* stream);

/* The critic's half of step t on its own stream (call both BEFORE the step is closed by rl_rollout_record / rl_rollout_record_slots):
 *   rl_rollout_values_slot      values [N] of the current step - hand it to rl_mlp_forward as the critic's output (out_dim 1);
 *   rl_rollout_store_critic_obs copies critic_obs [N][critic_dim] into the current step's slot, stream-ordered on `stream`.
 * The caller orders the streams: the critic of step t must have finished before rl_rollout_compute_returns, and before whoever
 * overwrites critic_obs (env step t + 1 with the env's two alternating observation buffers). */
int rl_rollout_values_slot(rl_rollout* r, float** values);
int rl_rollout_store_critic_obs(rl_rollout* r, const float* critic_obs, void* stream);

/* The same second half without a launch of its own: hands out the current step's slots (values [N] as stored by
 * rl_rollout_act, rewards [N], dones [N] u8) for a producer that writes them itself - rl_env_step_record
 * (include/rl_env.h) does it inside the env kernel - and closes the step like rl_rollout_record. */
int rl_rollout_record_slots(rl_rollout* r, const float** values, float** rewards, uint8_t** dones);

/* RolloutStorage.compute_returns(last_values [N], gamma, lam, normalize_advantage); needs a full storage. */
int rl_rollout_compute_returns(rl_rollout* r, const float* last_values, float gamma, float lam, int32_t normalize_advantage, void* stream);

/* RolloutStorage.clear(): back to step 0 (the random counter keeps running). */
int rl_rollout_clear(rl_rollout* r);

/* ---- hipGraph capture of a collection iteration (include/rl_env.h has the env's half and the protocol) ------------------
 * The only launch argument of this library that changes from one iteration to the next is the random counter; the act kernel
 * takes it as  *device word + launch literal.  A captured stretch [rl_rollout_clear; T x (rl_rollout_act; record); compute_returns]
 * therefore replays with fresh noise every time:
 *     rl_rollout_graph_begin(r, stream);        before hipStreamBeginCapture (anchors the device word)
 *     ... the captured calls ...
 *     rl_rollout_graph_end(r, stream);          inside the capture: appends the node advancing the word; host state rolled back
 *     per replay:  rl_rollout_graph_launching(r, stream);  hipGraphLaunch(...)      (host state := state after the stretch) */
int rl_rollout_graph_begin(rl_rollout* r, void* stream);
int rl_rollout_graph_end(rl_rollout* r, void* stream);
int rl_rollout_graph_launching(rl_rollout* r, void* stream);

int rl_rollout_get_buffer(rl_rollout* r, int32_t which, void** dev_ptr, int64_t* count);
int32_t rl_rollout_step(const rl_rollout* r); /* transitions recorded since the last clear */
int rl_rollout_destroy(rl_rollout* r);
const char* rl_rollout_last_error(void);

/* ---- symmetry data augmentation -------------------------------------------------
 * Replaces `compute_symmetric_states(env, obs, actions)`,
 * source/robot_lab/robot_lab/tasks/manager_based/locomotion/velocity/mdp/symmetry/anymal.py:27-87, which rsl_rl's PPO
 * calls on every mini-batch when `RslRlSymmetryCfg(use_data_augmentation=True, data_augmentation_func=...)` is set
 * (.../config/quadruped/anymal_d/agents/rsl_rl_ppo_cfg.py:100-105): the batch is replicated n_sym (= 4: identity,
 * left-right, front-back, diagonal) times and copy s has its columns permuted and sign-flipped,
 *     out[s * n_rows + r][c] = sign[s][c] * in[r][perm[s][c]]          (anymal.py:97-171 for the policy observation,
 *                                                                        :179-212 for the actions, tables :232-259)
 * One kernel launch for all copies; `perm` / `sign` are given once on the host. */
typedef struct rl_symmetry rl_symmetry;
int rl_symmetry_create(int32_t n_sym, int32_t dim, const int32_t* perm, const float* sign, int32_t device, rl_symmetry** out);
/* in [n_rows][dim] -> out [n_sym * n_rows][dim]; device pointers, stream-ordered. */
int rl_symmetry_apply(rl_symmetry* s, const float* in_dev, float* out_dev, int32_t n_rows, void* stream);
int rl_symmetry_destroy(rl_symmetry* s);

#ifdef __cplusplus
}
#endif
#endif
