// env_tables.h - device-side tables and state layout of the env-step lane program.
//
// Lane mapping (DESIGN.md "Lane mapping"): an environment is simulated by NLANE = 4 lane groups of a
// wavefront; group k owns limb chain k (a quadruped leg hip -> thigh -> calf [-> wheel]; a G1 leg or
// arm) and a share of the collision spheres of a trunk link.  The *trunk* is the base link plus an
// optional serial chain of NW joints hanging off it (G1: waist yaw/roll/pitch -> torso, which carries
// the arms); its joints are simulated redundantly by all lanes.
//
// HBM layout: wave-tiled structure-of-arrays (see "HBM state layout" below).
#pragma once
#include <stdint.h>
#include <type_traits>

#include "rl_math.h"

namespace rl {

constexpr int NLANE = 4;    // lane groups (= limb chains) per environment
constexpr int NLANE_ = 4;
constexpr int MAX_CL = 7;   // joints per limb chain (A1 3, Go2W 4, G1 arm 7)
constexpr int MAX_NW = 6;   // trunk joints (G1: the waist, 3; GR1: waist + head, 6)
constexpr int MAX_JX = MAX_CL + MAX_NW;  // per-lane joint arrays: [0, CL) limb joints, [CL, CL+NW) trunk joints
constexpr int MAX_SPL = 4;  // collision-sphere slots per link group
constexpr int MAX_NGRP = MAX_CL + 1;  // link groups per lane: 0 = share of a trunk link, 1..CL = limb links
constexpr int MAX_NBS = 9;  // body slots per lane: 0 = a trunk-link body (or empty), 1.. = limb bodies / sphere-less trunk bodies

// Compile-time shape of a lane program instance: limb chain length, trunk joints, sphere slots per link
// group, body slots per lane.  Loops are bounded by these (the tables / HBM layout by the MAX_* above).
// M0 = 1 ("merged"): the lane's share of the base link's spheres has no link group of its own - its spheres sit in free sphere
// slots of the limb link groups, flagged in LaneTabT::sph_base_mask (frame, twist and record of the BASE link for those slots).
// A 4-joint limb then has 4 link groups for its 4 sub-lanes instead of 5: one contact pass per substep instead of two.
// RP = 1 ("rotated / padded", quadruped instances): joint frames may be rotated against the parent link and a limb may have fewer than CL
// joints (inert padding joints; a limb may be empty) - what the trunk + limbs instances support anyway.  DDT Tita: two wheeled 4-joint
// legs with rotated hip frames, two empty limbs (round 5; it ran on the trunk + limbs instance before: 185 us at 2048 envs).
template <int CL_, int NW_, int SPL_, int NBS_, int M0_ = 0, int RP_ = 0>
struct Topo {
  static constexpr int CL = CL_, NW = NW_, SPL = SPL_, NBS = NBS_, JX = CL_ + NW_, NB = 6 + NW_, M0 = M0_;
  static_assert(M0_ == 0 || NW_ == 0, "merged base share: quadruped instances only");
  static constexpr bool ROT = NW_ > 0 || RP_ != 0;  // joint frames may be rotated w.r.t. the parent link (URDF joint rpy)
  static constexpr bool PAD = NW_ > 0 || RP_ != 0;  // a limb may have fewer than CL joints: joint_id -1 = inert padding (identity row in the elimination)
  static constexpr int DMAX = NLANE_ * CL_ + NW_;  // joints a model of this shape can have
  static constexpr int OBS_NC = 12 + 3 * DMAX;     // non-scan columns of an observation group (ObsGroupTabT)
};
using TopoQuad3 = Topo<3, 0, 3, 6>;  // A1, Go2
using TopoQuad4 = Topo<4, 0, 3, 6>;  // wheeled quadrupeds whose base spheres do not fit the free limb slots
using TopoQuad4M = Topo<4, 0, 3, 6, 1>;  // Go2W and the other wheeled quadrupeds
using TopoQuad4R = Topo<4, 0, 3, 6, 0, 1>;  // <= 4-joint limbs with rotated joint frames and / or limbs of unequal length (DDT Tita)
using TopoG1 = Topo<7, 3, 4, 9>;     // G1 29-DoF
using TopoGR = Topo<7, 6, 4, 9>;     // FFTAI GR1T1 / GR1T2 (32 DoF): a six-joint spine - waist, then head - with the arms leaving it at depth 3
constexpr int MAX_T = 40;   // reward terms
constexpr int MAX_OBS = 12;
constexpr int MAX_BASE_BODIES = 4;
constexpr int ENVS_PER_WAVE = 16;
constexpr int LOG_SIZE = 64;
constexpr int LOG_RING = 64;  // step k logs into slot k % LOG_RING (one slot = one step's extras["log"]); power of two
// A slot is LOG_PARTS partial rows of LOG_SIZE words, summed by the reader: wavefront w adds into row w % LOG_PARTS.  All adds of a launch on
// ONE row are serialised at ~44 ns each on the same address - nothing when four envs of 4096 reset in a step, 49 of 106 us when a third of
// them do (DDT Tita under random actions: 30 k adds on 26 addresses; every robot early in training) - profiles/r05m_tita_log_atomics.txt
constexpr int LOG_PARTS = 32;
// log accumulator slots (device LOG buffer)
enum { LOG_RESET_COUNT = 0, LOG_TERM_TIMEOUT = 1, LOG_TERM_OOB = 2, LOG_TERM_ILLEGAL = 3, LOG_METRIC_XY = 4, LOG_METRIC_YAW = 5, LOG_EP_SUM0 = 8,
       LOG_FRESH = 63 };  // LOG_FRESH: the resets of THIS very step (LOG_RESET_COUNT is inherited by a slot whose step reset nobody, this word is not)
static_assert(LOG_EP_SUM0 + MAX_T <= LOG_FRESH && LOG_SIZE == 64 && (LOG_PARTS & (LOG_PARTS - 1)) == 0, "log row: one word per lane of a wavefront");

using TopoMax = Topo<MAX_CL, MAX_NW, MAX_SPL, MAX_NBS>;  // shape of the host-side (unpacked) tables

// One per limb chain k.  Joint arrays: [0, CL) the limb, [CL, CL+NW) the trunk joints (same in all lanes).
// Sized by the Topo so that the LDS copy of a quadruped instance stays small: the workgroup's LDS
// footprint decides whether all 4 SIMDs of a CU get a wavefront (4 x 37 KB fit in 160 KB, 4 x 45 KB do not).
// The arrays the HOST fills per joint / sphere and pack_tables turns into the packed vectors below (jc, rota, sph): only the unpacked
// tables (TablesT<TopoMax>) carry them - in an instance's LDS image they would be dead weight (1.4 KB on a 3-joint quadruped: the
// one-lane-per-limb mapping's four-wavefront workgroup sits within 2 KB of a CU's 160 KB).
template <class TP, bool HOST>
struct LaneSrcT {};
template <class TP>
struct LaneSrcT<TP, true> {
  float origin[TP::JX][3], axis[TP::JX][3];
  float lower[TP::JX], upper[TP::JX], armature[TP::JX];
  int32_t act_implicit[TP::JX];
  float eff[TP::JX], sat[TP::JX], act_vlim[TP::JX];
  float sph_c[TP::CL + 1][TP::SPL][3];
  float sph_r[TP::CL + 1][TP::SPL];            // <= 0: empty slot
};
template <class TP>
struct LaneTabT : LaneSrcT<TP, std::is_same<TP, TopoMax>::value> {
  static constexpr bool HOST = std::is_same<TP, TopoMax>::value;
  static constexpr int JXA = TP::JX, RXA = TP::ROT ? TP::JX : 1, NG = TP::CL + 1;
  // What the SUBSTEP reads of a joint, packed as 16-byte vectors (pack_tables builds them from the arrays below): one ds_read_b128 per
  // vector where the arrays cost a ds_read_b32 / ds_read2_b32 per word - a lone wavefront pays an LDS round trip per dependent read.
  //   jc[j]   = [origin.x origin.y origin.z axis.x | axis.y axis.z eff sat | act_vlim flags armature lower | upper vel_limit - -]
  //             (flags: 1 = implicit actuator, 2 = velocity action)
  //   rota[j] = [rot0, row-major (9) | axis.x axis.y axis.z]   (trunk + limbs instances: read with a per-lane j by the dealt kinematics)
  alignas(16) float jc[JXA][16];
  alignas(16) float rota[RXA][12];
  alignas(16) float sph[NG][TP::SPL][4];  // collision spheres of a link group as vectors: [centre.xyz (link frame), radius (<= 0: empty slot)] (pack_tables, from sph_c / sph_r)
  float rot0[RXA][9];                  // joint frame axes in the parent link frame, row-major (only read when TP::ROT)
  float vel_limit[JXA];
  float q0[JXA], qd0[JXA], soft_lo[JXA], soft_hi[JXA];
  float kp0[JXA], kd0[JXA];
  int32_t action_is_vel[JXA];
  float a_scale[JXA], a_off[JXA], a_lo[JXA], a_hi[JXA];
  int32_t joint_id[JXA];               // task joint index (bit in joint masks, column in action/obs); -1 = padding (chain shorter than CL)
  int32_t joint_own[JXA];              // 1: this lane accounts for the joint in reward sums / observation columns / debug views
                                       //    (limb joints: always; trunk joints: lane 0 only)
  int32_t nj;                          // joints of this limb (<= CL)
  int32_t attach;                      // trunk joints that move the limb (0: hangs off the base, NW: off the last trunk link)
  int32_t grp0_depth;                  // trunk joints that move link group 0 of this lane
  int32_t sph_slot[NG][TP::SPL];       // body slot the sphere reports to
  int32_t slot_body[TP::NBS];          // global body index (bit in body masks), -1 = empty
  int32_t slot_grp[TP::NBS];           // link group the body is attached to
  float slot_pos[TP::NBS][3];          // body frame origin in its link frame
  int32_t base_body_local;             // which sphere-carrying trunk body (0..n_base_bodies-1) slot 0 / group 0 belongs to, -1 none
  int32_t owns_base_body;              // 1 if this lane keeps the timers of that trunk body
  uint32_t sph_base_mask;              // bit g*SPL+s: the sphere in slot (g, s) rides on the BASE link (merged instances; else 0)
  static constexpr int MAXOWN = TP::NBS > 6 ? 4 : 2;
  int32_t own_slot[4][MAXOWN];         // 16-lanes-per-env mapping: the body slots sub-lane s updates (ascending, -1 padded)
  static constexpr int MAXOWN2 = 4;
  int32_t own_slot2[2][MAXOWN2];       // 8-lanes-per-env mapping (two sub-lanes per limb; quadrupeds): likewise
  // 32-lanes-per-env mapping (eight sub-lanes per limb; trunk + limbs instances: CL = 7, so sub-lane s evaluates exactly link group
  // s - group 0, the lane's share of a trunk link plus the sphere-less trunk bodies parked with it, on sub-lane 0): at most MAXOWN8
  // slots per sub-lane (TaskTab::sub8_ok), found by the lane itself from slot_grp[] at kernel start - a table of them would have
  // pushed the 16-lane mapping's LDS image past the 80 KB that let two workgroups share a CU
  static constexpr int MAXOWN8 = 3;
  template <int SUB>
  static constexpr int maxown() { return SUB == 8 ? MAXOWN8 : (SUB == 2 ? MAXOWN2 : MAXOWN); }
};
using LaneTab = LaneTabT<TopoMax>;

struct RewTab {  // 48 bytes; index lists (joint_mirror pairs, gait feet) live in TaskTab::idx_pool_*
  int32_t kind;
  float weight;
  float p[4];
  uint32_t joint_mask;
  int32_t n_idx;
  uint64_t body_mask;
  int32_t idx_off;  // first entry of this term in the index pools
  int32_t row;      // joint-sum kinds: row of the env's joint-statistics table the term sums over its joint mask; else -1
};
// per-env reward tables in LDS (env_terms.h compute_rewards): REW_JS_ROWS statistics per task joint, REW_BT_NF words per body
// (a body's row is REW_BT_NS words of contact-sensor state, + REW_BT_NX more - net force, position and velocity relative to the
// root - only for the bodies of TaskTab::rew_ext_mask: the feet, typically.  17 bodies x 14 words x 16 envs per wavefront were
// 15 KB of LDS in the one-lane-per-limb mapping - one wavefront per CU less than fits now.)
constexpr int REW_JS_ROWS = 10, REW_BT_NS = 5, REW_BT_NX = 9, REW_BT_NF = REW_BT_NS + REW_BT_NX;
RL_FN int popcount64(uint64_t m) { return __builtin_popcountll(m); }
RL_FN int rew_tab_words(int D, int n_bodies, uint64_t ext_mask) { return REW_JS_ROWS * D + REW_BT_NS * n_bodies + REW_BT_NX * popcount64(ext_mask); }
// word offset of body b's row in the body table
RL_FN int rew_bt_row(uint64_t ext_mask, int b) { return REW_BT_NS * b + REW_BT_NX * popcount64(ext_mask & ((1ull << b) - 1ull)); }
RL_FN int rew_stage_words(int n_rewards) { return (n_rewards + 3) & ~3; }  // an env's row of the reward stage (per-term values of the step)
constexpr int IDX_POOL = 48;
constexpr int RESET_RAND_WORDS = 160;  // LDS words of an env's table of reset uniforms: Philox blocks 0 .. 39 of stream STREAM_RESET (rl_math.h IDX_*: the last index is IDX_LEVEL = 156)

struct ObsTab {
  int32_t kind;
  float scale, clip_lo, clip_hi, noise_lo, noise_hi;
  int32_t has_noise;
  int32_t offset;  // first column of the term in its group
};

// Observation groups as per-COLUMN descriptors: the host expands the term list (rl_env_host.h build_tables), so the lane program
// never dispatches on observation terms.  A column reads one entry of the env's feature vector (layout below), optionally adds
// uniform noise, clips, scales [UPSTREAM B2: noise -> clip -> scale].  The height scan (at most one per group) is ONE descriptor
// for its scan_n consecutive columns.
struct ObsColTab {  // 24 bytes
  float scale, clip_lo, clip_hi, noise_lo, noise_rng;  // noise_rng = noise_hi - noise_lo; both 0 when the column gets no noise
  int32_t src;                                         // feature index
};
template <int NC>
struct ObsGroupTabT {
  int32_t n_cols;           // non-scan columns, in row order
  int32_t scan_off, scan_n; // the scan occupies columns [scan_off, scan_off + scan_n) of the row (scan_n = 0: no scan)
  int32_t dim, corrupt, pad_[3];
  ObsColTab scan;
  ObsColTab col[NC];
};
// feature vector of an env (LDS, rebuilt by every step): base linear velocity, base angular velocity, projected gravity, velocity
// command, then per task joint index q - q0 | qd - qd0 | last action | q - q0 with the wheel joints zeroed (observations.py:17-27)
enum { FEAT_LIN = 0, FEAT_ANG = 3, FEAT_GRAV = 6, FEAT_CMD = 9, FEAT_JOINT = 12 };
RL_FN constexpr int feat_count(int D) { return FEAT_JOINT + 4 * D; }

struct TaskTab {  // everything that is not per limb
  int32_t CL, NW, SPL, NBS, D, n_bodies, n_base_bodies;
  int32_t nw_used;      // trunk joints the model really has (<= NW; the rest are inert padding)
  uint32_t slot_valid;  // bit g*SPL+s: some lane has a collision sphere in slot (g, s)
  int32_t merged;       // 1: tables built for a Topo<..., M0 = 1> instance (base-share spheres in limb slots, LaneTabT::sph_base_mask)
  int32_t rotpad;       // 1: a quadruped-shaped model (no trunk joints, limbs of <= 4 joints) with rotated joint frames or unequal limbs: Topo<4,0,3,6,0,1>
  int32_t sub8_ok;      // 1: the model fits the 32-lanes-per-env mapping (trunk + limbs instances: at most MAXOWN8 body slots per link group)
  uint32_t trunk_restart;          // bit i: trunk joint i hangs off the base (bit 0 always; Booster T1: waist AND neck on the base - two pieces)
  uint32_t trunk_anc[MAX_NW + 1];  // [d]: the trunk joints between the base and the trunk link at depth d (bit i = trunk joint i); [0] = 0
  int32_t wrench_depth; // trunk link (0 = base, i = after i trunk joints) carrying the body the wrench / COM events address
  int32_t scan_depth;   // trunk link carrying the height-scanner body
  float scan_pos[3];    // scanner body origin in that link's frame
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
  int32_t cur_lin, cur_ang, cur_lin_term, cur_ang_term;  // command_levels_* curricula (KState::cmd_levels holds the live ranges)
  int32_t n_policy, n_critic, policy_dim, critic_dim, policy_corrupt, critic_corrupt;
  ObsTab policy[MAX_OBS], critic[MAX_OBS];
  int32_t scan_nx, scan_ny;
  float scan_res, scan_offset;
  uint32_t wheel_joint_mask;
  int32_t n_rewards;
  // evaluation schedule of the reward terms (env_terms.h compute_rewards): slot s is evaluated as term rew_slot[s]; slots below
  // n_main by the full term_value(), slots from n_main on - they hold "scalar" kinds only, terms of the env's own scalars - by the
  // short scalar_term_value().  With 16 lanes per env, 17 - 32 terms cost one full trip + one scalar mini-trip instead of two full trips.
  int32_t n_main;
  int32_t rew_slot[MAX_T];
  uint64_t rew_rel_mask;  // bodies whose position / velocity relative to the root some reward term reads
  uint64_t rew_ext_mask;  // bodies with a long row in the reward body table: rew_rel_mask + the bodies whose net force a term reads
  int32_t idx_pool_a[IDX_POOL], idx_pool_b[IDX_POOL];
  int32_t term_time_out, term_oob, term_illegal;
  float oob_buffer, illegal_threshold;
  uint64_t illegal_body_mask;
  int32_t ev_wrench, ev_reset_joints, ev_gains, ev_reset_base, ev_push;
  float wrench_force[2], wrench_torque[2], reset_jpos[2], reset_jvel[2], gain_kp[2], gain_kd[2];
  float reset_pose[6][2], reset_vel[6][2], push_interval[2], push_vel[6][2];
  float default_root_pos[3], default_root_quat[4];
};

// Self-collision (trunk + limbs instance; include/rl_env.h rl_model_desc.self_pair).  The work of an env is dealt to 16 "virtual
// lanes" v = 4 k + s (limb k, sub-lane s; with fewer sub-lanes per limb a lane plays several): virtual lane v places at most one
// capsule - one of a link of ITS limb, or of a trunk link - and tests up to SELF_PPL capsule pairs.  These constants are read from
// the table image in HBM once per launch and kept in registers (the trunk + limbs instance has no LDS left for them: DESIGN.md 2).
constexpr int SELF_PPL = 5;   // pairs per virtual lane (RL_MAX_SELF_PAIRS / 16)
constexpr int SELF_CAPS = 16; // capsule slots of an env (RL_MAX_CAPSULES)
constexpr int SELF_CAP_WORDS = 8;  // env-shared LDS words of a placed capsule: centre, half axis (base coordinates), radius, bounding radius
struct SelfLaneTab {  // 64 bytes
  int32_t cap;          // capsule slot this virtual lane places, -1: none
  int32_t frame;        // chain slot of the capsule's link frame: limb joint j -> j, trunk joint i -> CL + i, -1: the base
  float p0[3], p1[3], r;  // link frame
  // pair word: a | b << 4 | ka << 8 | ga << 11 | kb << 15 | gb << 18, -1: none.  a, b: capsule slots; k: limb whose record takes the
  // force (7: a trunk link), g: link group of that limb (limb link j -> j + 1) or the trunk depth
  int32_t pair[SELF_PPL];
  int32_t pad_[2];
};
template <bool ON>
struct SelfTail {
  SelfLaneTab self_lane[SELF_CAPS];
};
template <>
struct SelfTail<false> {};

template <class TP>
struct TablesBody : TaskTab {
  LaneTabT<TP> lane[NLANE];
  ObsGroupTabT<TP::OBS_NC> obs[2];  // policy, critic
  RewTab rew[MAX_T];  // last of what is STAGED: only the first n_rewards entries go to LDS (KState::table_bytes)
};
template <class TP>
struct TablesT : TablesBody<TP>, SelfTail<(TP::NW > 0)> {};  // (the tail sits behind rew[] in the image: never staged)
using Tables = TablesT<TopoMax>;  // host side / export-import kernels; env kernels read the packed TablesT<TP>

// host: unpacked -> the instance's compact layout (trunk joints already sit at [CL, CL + NW) of the joint arrays)
template <class TP>
inline void pack_tables(const Tables& s, TablesT<TP>& d) {
  static_cast<TaskTab&>(d) = static_cast<const TaskTab&>(s);
  for (int t = 0; t < MAX_T; ++t) d.rew[t] = s.rew[t];
  if constexpr (TP::NW > 0)
    for (int v = 0; v < SELF_CAPS; ++v) d.self_lane[v] = s.self_lane[v];
  for (int g = 0; g < 2; ++g) {
    d.obs[g].n_cols = s.obs[g].n_cols; d.obs[g].scan_off = s.obs[g].scan_off; d.obs[g].scan_n = s.obs[g].scan_n;
    d.obs[g].dim = s.obs[g].dim; d.obs[g].corrupt = s.obs[g].corrupt; d.obs[g].scan = s.obs[g].scan;
    for (int c = 0; c < TP::OBS_NC && c < s.obs[g].n_cols; ++c) d.obs[g].col[c] = s.obs[g].col[c];
  }
  for (int k = 0; k < NLANE; ++k) {
    const LaneTab& a = s.lane[k];
    LaneTabT<TP>& b = d.lane[k];
    for (int j = 0; j < TP::JX; ++j) {
      if constexpr (LaneTabT<TP>::HOST) {
        for (int c = 0; c < 3; ++c) { b.origin[j][c] = a.origin[j][c]; b.axis[j][c] = a.axis[j][c]; }
        b.lower[j] = a.lower[j]; b.upper[j] = a.upper[j]; b.armature[j] = a.armature[j];
        b.act_implicit[j] = a.act_implicit[j]; b.eff[j] = a.eff[j]; b.sat[j] = a.sat[j]; b.act_vlim[j] = a.act_vlim[j];
      }
      if (TP::ROT) for (int c = 0; c < 9; ++c) b.rot0[TP::ROT ? j : 0][c] = a.rot0[j][c];
      b.vel_limit[j] = a.vel_limit[j];
      b.q0[j] = a.q0[j]; b.qd0[j] = a.qd0[j]; b.soft_lo[j] = a.soft_lo[j]; b.soft_hi[j] = a.soft_hi[j];
      b.kp0[j] = a.kp0[j]; b.kd0[j] = a.kd0[j]; b.action_is_vel[j] = a.action_is_vel[j];
      b.a_scale[j] = a.a_scale[j]; b.a_off[j] = a.a_off[j]; b.a_lo[j] = a.a_lo[j]; b.a_hi[j] = a.a_hi[j];
      b.joint_id[j] = a.joint_id[j]; b.joint_own[j] = a.joint_own[j];
      const float jc[16] = {a.origin[j][0], a.origin[j][1], a.origin[j][2], a.axis[j][0], a.axis[j][1], a.axis[j][2], a.eff[j], a.sat[j],
                            a.act_vlim[j], (float)((a.act_implicit[j] ? 1 : 0) + (a.action_is_vel[j] ? 2 : 0)), a.armature[j], a.lower[j],
                            a.upper[j], a.vel_limit[j], 0.f, 0.f};
      for (int c = 0; c < 16; ++c) b.jc[j][c] = jc[c];
      if (TP::ROT) {
        for (int c = 0; c < 9; ++c) b.rota[TP::ROT ? j : 0][c] = a.rot0[j][c];
        for (int c = 0; c < 3; ++c) b.rota[TP::ROT ? j : 0][9 + c] = a.axis[j][c];
      }
    }
    b.nj = a.nj; b.attach = a.attach; b.grp0_depth = a.grp0_depth;
    for (int g = 0; g <= TP::CL; ++g)
      for (int q = 0; q < TP::SPL; ++q) {
        if constexpr (LaneTabT<TP>::HOST) {
          for (int c = 0; c < 3; ++c) b.sph_c[g][q][c] = a.sph_c[g][q][c];
          b.sph_r[g][q] = a.sph_r[g][q];
        }
        b.sph_slot[g][q] = a.sph_slot[g][q];
        for (int c = 0; c < 3; ++c) b.sph[g][q][c] = a.sph_c[g][q][c];
        b.sph[g][q][3] = a.sph_r[g][q];
      }
    for (int q = 0; q < TP::NBS; ++q) {
      b.slot_body[q] = a.slot_body[q]; b.slot_grp[q] = a.slot_grp[q];
      for (int c = 0; c < 3; ++c) b.slot_pos[q][c] = a.slot_pos[q][c];
    }
    b.base_body_local = a.base_body_local; b.owns_base_body = a.owns_base_body; b.sph_base_mask = a.sph_base_mask;
    for (int q = 0; q < 4; ++q)
      for (int i = 0; i < LaneTabT<TP>::MAXOWN; ++i) b.own_slot[q][i] = a.own_slot[q][i];
    for (int q = 0; q < 2; ++q)
      for (int i = 0; i < LaneTabT<TP>::MAXOWN2; ++i) b.own_slot2[q][i] = a.own_slot2[q][i];
  }
}

// fields of the per-env "cmd" record
enum { CMD_VX = 0, CMD_VY, CMD_WZ, CMD_HEADING, CMD_TIME_LEFT, CMD_METRIC_XY, CMD_METRIC_YAW, CMD_PUSH_LEFT, CMD_NFIELD };
// link inertia record: mass, com(3), inertia about com (xx yy zz xy xz yz), link frame
constexpr int INERTIA_NF = 10;

// ---- HBM state layout: wave-tiled structure-of-arrays -------------------------------------------
// A tile is the state of one wavefront (16 envs; 8 / 4 / 2 envs in the 8- / 16- / 32-lanes-per-env mappings).  Rows are
// laid out for the MAX_* shape (unused rows are never touched, so they cost address space only).  Inside a
// tile every field is one contiguous row: one float per leg for lane fields, one per env for env fields.  A wavefront therefore
// reads/writes whole coalesced rows, and - the reason for tiling rather than [field][N] planes - every
// access is `tile base (SGPR) + lane offset (ONE VGPR) + compile-time row offset`; with [field][N]
// planes hipcc kept ~60 separate 64-bit address pairs live across the kernel (120 VGPRs, profiles/r01).
// Field rows of a tile for an instance shape (CL limb joints, NW trunk joints, NBS body slots): only the
// rows the instance uses exist, so a tile is one dense block of HBM.
struct Layout {
  // lane fields (one float per limb)
  int LF_Q, LF_QD, LF_KP, LF_KD, LF_ACT;
  int LF_INERTIA;   // [CL][10]
  int LF_TIMERS;    // [NBS][4] current_air, current_contact, last_air, last_contact
  int LF_FRICTION;  // [NBS][3] mu_s, mu_d, restitution
  int NF_LANE;
  // env fields (one float per env)
  int EF_ROOT;          // pos(3) quat wxyz(4) lin vel (3, world, link origin) ang vel (3, world)
  int EF_WRENCH;        // force(3) torque(3), base-body frame
  int EF_BASE_INERTIA;  // [1 + NW][10] trunk links: base, then the link after each trunk joint
  int EF_BASE_COM;      // COM of the root *body* in the base frame (root COM velocity)
  int EF_WR_COM;        // COM of the wrench body in its trunk link frame
  int EF_TQ, EF_TQD, EF_TKP, EF_TKD, EF_TACT;  // trunk joints
  int EF_CMD, EF_ORIGIN, NF_ENV;
  RL_FN constexpr Layout(int CL, int NW, int NBS)
      : LF_Q(0), LF_QD(CL), LF_KP(2 * CL), LF_KD(3 * CL), LF_ACT(4 * CL), LF_INERTIA(5 * CL), LF_TIMERS(5 * CL + CL * INERTIA_NF),
        LF_FRICTION(5 * CL + CL * INERTIA_NF + NBS * 4), NF_LANE(5 * CL + CL * INERTIA_NF + NBS * 7),
        EF_ROOT(0), EF_WRENCH(13), EF_BASE_INERTIA(19), EF_BASE_COM(19 + (1 + NW) * INERTIA_NF), EF_WR_COM(EF_BASE_COM + 3),
        EF_TQ(EF_WR_COM + 3), EF_TQD(EF_TQ + NW), EF_TKP(EF_TQD + NW), EF_TKD(EF_TKP + NW), EF_TACT(EF_TKD + NW), EF_CMD(EF_TACT + NW),
        EF_ORIGIN(EF_CMD + CMD_NFIELD), NF_ENV(EF_ORIGIN + 3) {}
};
// `ept` = environments per tile (= per wavefront): 16 when one lane simulates a leg, 4 when a leg is
// spread over 4 sub-lanes.  A lane-field row then has 4*ept entries (one per leg), an env-field row ept.
RL_FN size_t lane_index(const Layout& ly, int e, int k, int f, int ept) {
  return ((size_t)(e / ept) * ly.NF_LANE + (size_t)f) * (size_t)(NLANE * ept) + (size_t)(e % ept) * NLANE + k;
}
RL_FN size_t env_index(const Layout& ly, int e, int f, int ept) { return ((size_t)(e / ept) * ly.NF_ENV + (size_t)f) * (size_t)ept + (size_t)(e % ept); }

// observation group g of a one-lane-per-limb launch is written straight to HBM (no LDS staging row) when it carries no noise
// (env_terms.h write_group<DIRECT>; the LDS sizing in rl_env.hip follows the same predicate)
template <class TabT>
RL_FN bool direct_group(const TabT& T, int g) { return T.obs[g].corrupt == 0; }

// layout of KState::cmd_levels / RL_BUF_CMD_LEVELS
enum CmdLevels { CL_LIN_X = 0, CL_LIN_Y = 2, CL_ANG_Z = 4, CL_SUM_LIN = 8, CL_CNT_LIN = 9, CL_SUM_ANG = 10, CL_CNT_ANG = 11, CL_WORDS = 16 };
struct CmdLevelParams {  // what the end-of-step decision needs (VEL/mdp/curriculums.py:21-94)
  int32_t lin, ang;
  float lin_weight, ang_weight, max_episode_length_s;
  float final_x[2], final_y[2], final_z[2];
};
// the decision itself, run once after a step whose counter is a multiple of the episode length (one thread / the host)
RL_FN void apply_cmd_levels(float* lv, const CmdLevelParams& P) {
  if (P.lin && lv[CL_CNT_LIN] > 0.f && lv[CL_SUM_LIN] / lv[CL_CNT_LIN] / P.max_episode_length_s > 0.8f * P.lin_weight) {
    lv[CL_LIN_X + 0] = fminf(fmaxf(lv[CL_LIN_X + 0] - 0.1f, P.final_x[0]), P.final_x[1]);
    lv[CL_LIN_X + 1] = fminf(fmaxf(lv[CL_LIN_X + 1] + 0.1f, P.final_x[0]), P.final_x[1]);
    lv[CL_LIN_Y + 0] = fminf(fmaxf(lv[CL_LIN_Y + 0] - 0.1f, P.final_y[0]), P.final_y[1]);
    lv[CL_LIN_Y + 1] = fminf(fmaxf(lv[CL_LIN_Y + 1] + 0.1f, P.final_y[0]), P.final_y[1]);
  }
  if (P.ang && lv[CL_CNT_ANG] > 0.f && lv[CL_SUM_ANG] / lv[CL_CNT_ANG] / P.max_episode_length_s > 0.8f * P.ang_weight) {
    lv[CL_ANG_Z + 0] = fminf(fmaxf(lv[CL_ANG_Z + 0] - 0.1f, P.final_z[0]), P.final_z[1]);
    lv[CL_ANG_Z + 1] = fminf(fmaxf(lv[CL_ANG_Z + 1] + 0.1f, P.final_z[0]), P.final_z[1]);
  }
  lv[CL_SUM_LIN] = lv[CL_CNT_LIN] = lv[CL_SUM_ANG] = lv[CL_CNT_ANG] = 0.f;
}

// what a launch of the env kernels runs (KState::mode): a whole step, the reset entry, or one half of a step that is split around
// the decision of the command_levels_* curricula (env_terms.h step_head / step_tail)
enum KMode { KMODE_STEP = 0, KMODE_RESET = 1, KMODE_STEP_HEAD = 2, KMODE_STEP_TAIL = 3 };

struct KState {
  int32_t N;      // environments the caller sees
  int32_t Npad;   // simulated (multiple of ENVS_PER_WAVE)
  int32_t ept;    // environments per tile / wavefront (16, 8, 4 or 2)
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
  float* log;                      // [LOG_RING][LOG_PARTS][LOG_SIZE]
  // optional inspection buffers (nullptr = skip)
  float *dbg_torque, *dbg_acc;     // [Npad][D]
  float* dbg_cforce;               // [Npad][B][3]
  // inputs
  const float* terrain;            // [nx*ny]
  const float* terrain_origins;    // [rows][cols][3]
  const float* action_in;          // [N][D]
  const uint8_t* reset_mask;       // [Npad] (reset mode)
  // optional rollout sink of rl_env_step_record (all NULL for a plain step): what PPO.process_env_step stores for this step
  const float* ro_values;          // [N] V(s_t) of the critic
  float* ro_rewards;               // [N] <- reward + gamma * V * time_out
  uint8_t* ro_dones;               // [N] <- terminated | time_out
  float ro_gamma;
  float* cmd_levels;               // [16] live command ranges + decision accumulators of the command_levels_* curricula (CmdLevels)
  uint64_t seed;
  const uint32_t* step_base;       // the step count of a launch = *step_base + step_counter (hipGraph replays advance the device word, rl_env_graph_*)
  uint32_t step_counter;
  uint32_t table_bytes;  // bytes of the packed table image the env kernels stage into LDS (multiple of 16)
  int32_t mode;          // KMode
  float self_k;          // self-collision penalty stiffness; 0: no self-collision pass (rl_sim_desc.self_k when the model lists capsule pairs)
  int32_t self_trips;    // pair slots per virtual lane that are in use: ceil(pairs / 16) <= SELF_PPL
};

}  // namespace rl
