/* rl_policy.h - C-ABI of the fused policy / value MLP inference kernel (the
 * component next to the env in the rollout loop).
 *
 * Replaces (reference call sites; the arithmetic itself lives in the third-party rsl_rl `ActorCritic`):
 *   - `policy = runner.get_inference_policy(device=...)`; `actions = policy(obs)`
 *        scripts/reinforcement_learning/rsl_rl/play.py:207,246
 *   - the actor / critic forward inside `runner.learn()` rollouts, scripts/reinforcement_learning/rsl_rl/train.py:206-224
 *   - network shape: `RslRlPpoActorCriticCfg(actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
 *        activation="elu")`, .../unitree_a1/agents/rsl_rl_ppo_cfg.py:15-22  (A1: 45 -> 512 -> 256 -> 128 -> 12 and
 *        235 -> 512 -> 256 -> 128 -> 1; the last layer is linear)
 *
 * y = W_L act(... act(W_1 x + b_1) ...) + b_L in fp32 (exact-f32 MFMA, v_mfma_f32_16x16x4_f32), one kernel launch
 * for the whole network.  All pointers passed to rl_mlp_forward are DEVICE pointers; weights are given once, on the
 * host, in the torch.nn.Linear layout ([out_features][in_features] row-major).  Only inference is provided
 * (no autograd, no rollout storage / GAE - out of scope this round). */
#ifndef RL_POLICY_H
#define RL_POLICY_H

#include <stdint.h>

#include "rl_act.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RL_MLP_MAX_LAYERS 8
#define RL_MLP_MAX_WIDTH 512 /* widest layer input / output */

enum rl_mlp_activation { RL_ACT_ELU = 0, RL_ACT_RELU = 1, RL_ACT_TANH = 2 };

typedef struct rl_mlp rl_mlp;

/* dims[0] = input width, dims[1..n_layers] = layer output widths (n_layers Linear layers; the activation follows
 * every layer but the last).  weights[l] / biases[l]: HOST pointers, nn.Linear layout [dims[l+1]][dims[l]] / [dims[l+1]]. */
int rl_mlp_create(const int32_t* dims, int32_t n_layers, int32_t activation, const float* const* weights,
                  const float* const* biases, int32_t device, rl_mlp** out);

/* New parameters for an existing network, in place (same HOST layout as rl_mlp_create): what a training loop calls after every
 * optimiser step so that the next rollout runs the updated actor / critic (train.py:224 -> rsl_rl OnPolicyRunner.learn: `alg.update()`
 * is followed by `alg.act()` on the same modules).  The device images keep their addresses, so a captured graph that launches this
 * network stays valid.  Waits for EVERY launch queued on the device (any stream), then copies synchronously: when the call returns no
 * forward can be reading a half-updated image; the caller must not launch this network from another thread while the call runs.
 * Not capturable: fails (without disturbing the capture) when `stream` is being captured - push parameters between graph replays. */
int rl_mlp_set_weights(rl_mlp* m, const float* const* weights, const float* const* biases, void* stream);

/* y[n_rows][dims[n_layers]] = MLP(x[n_rows][dims[0]]); x, y: device pointers, row-major; stream-ordered. */
int rl_mlp_forward(rl_mlp* m, const float* x_dev, float* y_dev, int32_t n_rows, void* stream);

/* The same result from the SMALL-FOOTPRINT launch: 16 rows per workgroup of four wavefronts, exact-f32 MFMA (v_mfma_f32_16x16x4_f32), 140
 * registers per lane and 64 KB of LDS - sized to run on a CU BESIDE another kernel's resident workgroup (the env-step kernel of 4096
 * quadruped envs leaves 200 registers per SIMD and 83 KB of LDS per CU): the critic of a rollout step on a second stream under the env
 * step (robot_lab_amd/collect.py).  Slower than rl_mlp_forward when it has the chip to itself; agrees with it to fp32 round-off. */
int rl_mlp_forward_small(rl_mlp* m, const float* x_dev, float* y_dev, int32_t n_rows, void* stream);

/* Two networks over the same n_rows rows in one launch (the actor and the critic of a rollout step,
 * train.py:206-224 -> rsl_rl PPO.act: `policy.act(obs)` and `policy.evaluate(privileged_obs)` back to back):
 * ya = A(xa), yb = B(xb).  Same results as two rl_mlp_forward calls; the two networks' workgroups share the CUs. */
int rl_mlp_forward_pair(rl_mlp* a, const float* xa_dev, float* ya_dev, rl_mlp* b, const float* xb_dev, float* yb_dev, int32_t n_rows,
                        void* stream);

/* rl_mlp_forward_pair with the rollout step's stochastic head in its epilogue (include/rl_act.h; VERDICT r5 item 2: "sampling / log-prob /
 * storage in the actor's epilogue"): the actor's workgroups sample a = mu + sigma eps for their rows, write actions_out and the slot's actions /
 * mu / sigma / log-prob and copy their observation rows into the slot; the critic's workgroups copy their privileged-observation rows; pass
 * `ep->s_values` as yb_dev and V lands in the slot.  What rl_rollout_act launches a kernel for.  Returns 0 when done in the launch, 1 when
 * this size / precision runs a kernel without the epilogue (NOTHING was launched: call rl_mlp_forward_pair and rl_rollout_act), -1 on error. */
int rl_mlp_forward_pair_act(rl_mlp* a, const float* xa_dev, float* ya_dev, rl_mlp* b, const float* xb_dev, float* yb_dev, int32_t n_rows,
                            const rl_act_epilogue* ep, void* stream);

int32_t rl_mlp_in_dim(const rl_mlp* m);
int32_t rl_mlp_out_dim(const rl_mlp* m);
int rl_mlp_destroy(rl_mlp* m);
const char* rl_mlp_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
