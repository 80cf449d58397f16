// env_tables.h - device-side tables and state layout of the env-step lane program.
//
// Lane mapping (DESIGN.md "Lane mapping"): an environment is simulated by NLANE = 4 adjacent lanes of
// a wavefront; lane k owns kinematic chain k (one leg: hip -> thigh -> calf [-> wheel]) and a share of
// the base link's collision spheres.  16 environments per 64-wide wavefront.
//
// HBM layout: wave-tiled structure-of-arrays (see "HBM state layout" below).
#pragma once
#include <stdint.h>

#include "rl_math.h"

namespace rl {

constexpr int NLANE = 4;    // lanes (= chains) per environment
constexpr int MAX_CL = 4;   // joints per chain
constexpr int SPL = 3;      // collision-sphere slots per link group
constexpr int NGRP = MAX_CL + 1;  // link groups per lane: 0 = share of the base link, 1..CL = chain links
constexpr int NBS = 6;      // body slots per lane: 0 = a base-link body (or empty), 1.. = chain bodies
constexpr int MAX_T = 40;   // reward terms
constexpr int MAX_OBS = 12;
constexpr int MAX_BASE_BODIES = 4;
constexpr int ENVS_PER_WAVE = 16;
constexpr int LOG_SIZE = 64;
// log accumulator slots (device LOG buffer)
enum { LOG_RESET_COUNT = 0, LOG_TERM_TIMEOUT = 1, LOG_TERM_OOB = 2, LOG_TERM_ILLEGAL = 3, LOG_METRIC_XY = 4, LOG_METRIC_YAW = 5, LOG_EP_SUM0 = 8 };

struct LaneTab {  // one per chain k
  float origin[MAX_CL][3], axis[MAX_CL][3];
  float lower[MAX_CL], upper[MAX_CL], vel_limit[MAX_CL], armature[MAX_CL];
  float q0[MAX_CL], qd0[MAX_CL], soft_lo[MAX_CL], soft_hi[MAX_CL];
  int32_t act_implicit[MAX_CL];
  float kp0[MAX_CL], kd0[MAX_CL], eff[MAX_CL], sat[MAX_CL], act_vlim[MAX_CL];
  int32_t action_is_vel[MAX_CL];
  float a_scale[MAX_CL], a_off[MAX_CL], a_lo[MAX_CL], a_hi[MAX_CL];
  int32_t joint_id[MAX_CL];            // task joint index (bit in joint masks, column in action/obs)
  float sph_c[NGRP][SPL][3];
  float sph_r[NGRP][SPL];              // <= 0: empty slot
  int32_t sph_slot[NGRP][SPL];         // body slot the sphere reports to
  int32_t slot_body[NBS];              // global body index (bit in body masks), -1 = empty
  int32_t slot_grp[NBS];               // link group the body is attached to
  float slot_pos[NBS][3];              // body frame origin in its link frame
  int32_t base_body_local;             // which base-link body (0..n_base_bodies-1) slot 0 / group 0 belongs to, -1 none
  int32_t owns_base_body;              // 1 if this lane keeps the timers of that base-link body
};

struct RewTab {
  int32_t kind;
  float weight;
  float p[8];
  uint32_t joint_mask;
  uint64_t body_mask;
  int32_t idx_a[16], idx_b[16];
  int32_t n_idx;
};

struct ObsTab {
  int32_t kind;
  float scale, clip_lo, clip_hi, noise_lo, noise_hi;
  int32_t has_noise;
  int32_t offset;  // first column of the term in its group
};

struct Tables {
  LaneTab lane[NLANE];
  int32_t CL, D, n_bodies, n_base_bodies;
  uint32_t slot_valid;  // bit g*SPL+s: some lane has a collision sphere in slot (g, s)
  // sim
  float dt;
  int32_t decimation;
  float gravity, contact_k, contact_c, contact_phi_ref, contact_ct, contact_vdep, contact_vstick, limit_k, limit_c, force_threshold;
  // terrain
  int32_t is_plane, nx, ny;
  float hscale, x0, y0;
  int32_t num_rows, num_cols;
  float tile_size, border;
  int32_t curriculum;
  // task
  float step_dt, max_episode_length_s;
  int32_t max_episode_length;
  float cmd_range[4][2], cmd_resample[2], cmd_rel_standing, cmd_rel_heading, cmd_heading_stiffness, cmd_small_threshold;
  int32_t cmd_heading;
  int32_t n_policy, n_critic, policy_dim, critic_dim, policy_corrupt, critic_corrupt;
  ObsTab policy[MAX_OBS], critic[MAX_OBS];
  int32_t scan_nx, scan_ny;
  float scan_res, scan_offset;
  uint32_t wheel_joint_mask;
  int32_t n_rewards;
  RewTab rew[MAX_T];
  int32_t term_time_out, term_oob, term_illegal;
  float oob_buffer, illegal_threshold;
  uint64_t illegal_body_mask;
  int32_t ev_wrench, ev_reset_joints, ev_gains, ev_reset_base, ev_push;
  float wrench_force[2], wrench_torque[2], reset_jpos[2], reset_jvel[2], gain_kp[2], gain_kd[2];
  float reset_pose[6][2], reset_vel[6][2], push_interval[2], push_vel[6][2];
  float default_root_pos[3], default_root_quat[4];
};

// fields of the per-env "cmd" record
enum { CMD_VX = 0, CMD_VY, CMD_WZ, CMD_HEADING, CMD_TIME_LEFT, CMD_METRIC_XY, CMD_METRIC_YAW, CMD_PUSH_LEFT, CMD_NFIELD };
// link inertia record: mass, com(3), inertia about com (xx yy zz xy xz yz), link frame
constexpr int INERTIA_NF = 10;

// ---- HBM state layout: wave-tiled structure-of-arrays -------------------------------------------
// A tile is the state of one wavefront (16 envs, or 4 envs in the 16-lanes-per-env mapping).  Inside a
// tile every field is one contiguous row: one float per leg for lane fields, one per env for env fields.  A wavefront therefore
// reads/writes whole coalesced rows, and - the reason for tiling rather than [field][N] planes - every
// access is `tile base (SGPR) + lane offset (ONE VGPR) + compile-time row offset`; with [field][N]
// planes hipcc kept ~60 separate 64-bit address pairs live across the kernel (120 VGPRs, profiles/r01).
enum {  // lane fields (rows of 64)
  LF_Q = 0, LF_QD = LF_Q + MAX_CL, LF_KP = LF_QD + MAX_CL, LF_KD = LF_KP + MAX_CL, LF_ACT = LF_KD + MAX_CL,
  LF_INERTIA = LF_ACT + MAX_CL,                    // [MAX_CL][10]
  LF_TIMERS = LF_INERTIA + MAX_CL * INERTIA_NF,    // [NBS][4] current_air, current_contact, last_air, last_contact
  LF_FRICTION = LF_TIMERS + NBS * 4,               // [NBS][3] mu_s, mu_d, restitution
  NF_LANE = LF_FRICTION + NBS * 3
};
enum {  // env fields (rows of 16)
  EF_ROOT = 0,                 // pos(3) quat wxyz(4) lin vel (3, world, link origin) ang vel (3, world)
  EF_WRENCH = EF_ROOT + 13,    // force(3) torque(3), base-body frame
  EF_BASE_INERTIA = EF_WRENCH + 6,
  EF_BASE_COM = EF_BASE_INERTIA + INERTIA_NF,  // COM of the base *body* in the base frame
  EF_CMD = EF_BASE_COM + 3,
  EF_ORIGIN = EF_CMD + CMD_NFIELD,
  NF_ENV = EF_ORIGIN + 3
};
// `ept` = environments per tile (= per wavefront): 16 when one lane simulates a leg, 4 when a leg is
// spread over 4 sub-lanes.  A lane-field row then has 4*ept entries (one per leg), an env-field row ept.
RL_FN size_t lane_index(int e, int k, int f, int ept) { return ((size_t)(e / ept) * NF_LANE + (size_t)f) * (size_t)(NLANE * ept) + (size_t)(e % ept) * NLANE + k; }
RL_FN size_t env_index(int e, int f, int ept) { return ((size_t)(e / ept) * NF_ENV + (size_t)f) * (size_t)ept + (size_t)(e % ept); }

struct KState {
  int32_t N;      // environments the caller sees
  int32_t Npad;   // simulated (multiple of ENVS_PER_WAVE)
  int32_t ept;    // environments per tile / wavefront (16 or 4)
  float* lane_state;  // [Npad/ept][NF_LANE][4*ept]
  float* env_state;   // [Npad/ept][NF_ENV][ept]
  int32_t* flags;    // [Npad] bit0 is_heading_env, bit1 is_standing_env
  int32_t *level, *ttype;  // [Npad]
  int64_t* ep_len;   // [Npad]   (caller-visible int64 [N])
  float* ep_sums;    // [MAX_T][Npad]
  // caller-visible outputs (reference layouts)
  float *obs_policy, *obs_critic;  // [Npad][dim]
  float* reward;                   // [Npad]
  uint8_t *terminated, *time_out;  // [Npad]
  float* rew_terms;                // [MAX_T][Npad]
  float* command_out;              // [Npad][3]
  float* log;                      // [LOG_SIZE]
  // optional inspection buffers (nullptr = skip)
  float *dbg_torque, *dbg_acc;     // [Npad][D]
  float* dbg_cforce;               // [Npad][B][3]
  // inputs
  const float* terrain;            // [nx*ny]
  const float* terrain_origins;    // [rows][cols][3]
  const float* action_in;          // [N][D]
  const uint8_t* reset_mask;       // [Npad] (reset mode)
  uint64_t seed;
  uint32_t step_counter;
};

}  // namespace rl
