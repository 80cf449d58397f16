// env_spec.h - TASK-SPECIALISED term stack: the reward terms and observation groups of one task as compile-time constants.
//
// The generic lane program (env_terms.h compute_rewards / write_group) INTERPRETS the task: reward descriptors read from the LDS table
// image, a 39-way switch per lane, joint / body statistics published to LDS so that any term can read anything, observation columns
// gathered through per-column tables.  That costs 37 % of the A1 Rough step for ~7 % of its flops (profiles/r04l_phase_clock_a1.txt).
// The task is known at rl_env_create, and for the tasks whose term lists are compiled in here (spec/env_specs_gen.h, written by
// tools/gen_specs.py from the same descriptor -> Tables compile the product runs) the step kernel is instantiated on a `Spec`: term
// kinds, weights, parameters, joint / body masks and index lists are constant expressions, every term is evaluated from the lanes'
// own registers, joint sums are one pass over the lane's own joints + one DPP sum per term, nothing is published, no switch remains.
//
// rl_env_create picks a Spec only when spec_matches<Spec>() finds the env's compiled Tables IDENTICAL, field by field and bit by bit,
// to the constants the Spec was generated from (a cfg with one edited weight runs the interpreter - slower, never wrong);
// RL_ENV_SPEC=0 forces the interpreter (A/B runs, and the tests that hold the two paths against each other).
#pragma once
#include <string.h>

#include "env_tables.h"

namespace rl {

struct RewSpec {  // constexpr twin of RewTab (+ its index lists, which RewTab keeps in TaskTab::idx_pool_*)
  int kind;
  float weight;
  float p[4];
  uint32_t joint_mask;
  uint64_t body_mask;
  int n_idx;
  int idx_a[16], idx_b[16];
};
struct ObsSpec {  // constexpr twin of ObsTab
  int kind;
  float scale, clip_lo, clip_hi, noise_lo, noise_hi;
  int has_noise, offset;
};
struct NoSpec {  // the interpreter
  static constexpr bool ON = false;
  static constexpr int ID = 0;
};

// joints of limb k (bit j: the lane's joint slot j - [0, CL) limb joints, [CL, CL + NW) trunk joints, which limb 0 accounts for) that
// reward term t sums over
template <class SP>
constexpr uint32_t spec_lmask(int t, int k) {
  uint32_t m = 0;
  for (int jid = 0; jid < SP::D; ++jid)
    if (((SP::REW[t].joint_mask >> jid) & 1u) && SP::JOINT_K[jid] == k) m |= 1u << SP::JOINT_J[jid];
  return m;
}
template <class SP>
constexpr uint64_t spec_rel_mask() {  // bodies whose position / velocity relative to the root some term reads (= TaskTab::rew_rel_mask of these kinds)
  uint64_t m = 0;
  for (int t = 0; t < SP::N_REW; ++t) {
    const int kd = SP::REW[t].kind;
    if (kd == 15 /*FEET_HEIGHT_BODY*/ || kd == 19 /*FEET_SLIDE*/ || kd == 27 /*FEET_HEIGHT*/ || kd == 31 /*HANDSTAND_FEET_HEIGHT_EXP*/) m |= SP::REW[t].body_mask;
    if (kd == 37 /*FEET_DISTANCE_Y_EXP*/ || kd == 38 /*FEET_DISTANCE_XY_EXP*/)
      for (int i = 0; i < SP::REW[t].n_idx; ++i) m |= 1ull << SP::REW[t].idx_a[i];
  }
  return m;
}
// Axis kinds of the limb joints (round 6): SP::AXIS_KIND[j] = 0 / 1 / 2 when limb joint j of EVERY limb turns about +-e_x / e_y / e_z of its
// joint frame (exactly: components 0 and +-1 after the tables' normalisation), 3 otherwise.  The kinematics of a Spec then compose
// R_parent * Rot(e_i, q) in its sparse form (12 multiply-adds and no Rodrigues matrix instead of ~50 instructions, env_step.h
// chain_kinematics) - five kinematics passes per step on the quadrupeds.  The interpreter (NoSpec) and the trunk + limbs instances: general.
template <class SP>
constexpr int spec_axis_kind(int j) {
  if constexpr (SP::ON) return SP::AXIS_KIND[j];
  else return 3;
}
// the kind the tables of limb k give joint slot j (host: spec generator and spec_matches)
template <class LT>
inline int table_axis_kind(const LT& L, int j) {
  int kind = 3;
  for (int i = 0; i < 3; ++i) {
    const float a = L.axis[j][i], b = L.axis[j][(i + 1) % 3], c = L.axis[j][(i + 2) % 3];
    if ((a == 1.0f || a == -1.0f) && b == 0.0f && c == 0.0f) kind = i;
  }
  return kind;
}
// reward kinds the specialised evaluation implements (a task with any other runs the interpreter: tools/gen_specs.py / robot_lab_amd/jit.py
// say so).  0 .. 38; not action_mirror (39) / action_sync (40): weight 0 in every shipped cfg.
constexpr bool spec_kind_supported(int kd) {
  return kd >= 0 && kd <= 38;
}

// host: is the env's compiled table image exactly what the Spec was generated from?
template <class SP>
inline bool spec_matches(const TablesT<TopoMax>& T) {
  auto same = [](float a, float b) { return memcmp(&a, &b, 4) == 0; };
  if (T.CL != SP::TP::CL || T.NW != SP::TP::NW || (T.merged != 0) != (SP::TP::M0 != 0) || (T.rotpad != 0) != (SP::TP::NW == 0 && SP::TP::PAD) || T.D != SP::D ||
      T.n_bodies != SP::N_BODIES)
    return false;
  if (T.n_rewards != SP::N_REW || T.cur_lin || T.cur_ang) return false;
  for (int j = 0; j < SP::TP::CL; ++j)  // a joint the Spec composes in sparse form must be axis-aligned in every limb of THIS env's tables
    for (int k = 0; k < NLANE; ++k)
      if (SP::AXIS_KIND[j] != 3 && table_axis_kind(T.lane[k], j) != SP::AXIS_KIND[j]) return false;
  // the joint map: task joint jid = slot JOINT_J[jid] of limb JOINT_K[jid], and nothing else is owned
  int owned = 0;
  for (int k = 0; k < NLANE; ++k)
    for (int j = 0; j < T.CL + T.NW; ++j) {  // (trunk joints sit at [CL, CL + NW) of the unpacked tables, too)
      const int jid = T.lane[k].joint_own[j] ? T.lane[k].joint_id[j] : -1;
      if (jid < 0) continue;
      ++owned;
      if (jid >= SP::D || SP::JOINT_K[jid] != k || SP::JOINT_J[jid] != j) return false;
    }
  if (owned != SP::D) return false;
  for (int t = 0; t < SP::N_REW; ++t) {
    const RewTab& R = T.rew[t];
    const RewSpec& Q = SP::REW[t];
    if (R.kind != Q.kind || !same(R.weight, Q.weight) || R.joint_mask != Q.joint_mask || R.body_mask != Q.body_mask || R.n_idx != Q.n_idx) return false;
    for (int i = 0; i < 4; ++i)
      if (!same(R.p[i], Q.p[i])) return false;
    const int nidx = R.kind == 20 /*FEET_GAIT*/ ? 4 : R.n_idx;
    for (int i = 0; i < nidx; ++i)
      if (T.idx_pool_a[R.idx_off + i] != Q.idx_a[i] || T.idx_pool_b[R.idx_off + i] != Q.idx_b[i]) return false;
  }
  if (T.n_policy != SP::N_OBS[0] || T.n_critic != SP::N_OBS[1] || T.policy_dim != SP::OBS_DIM[0] || T.critic_dim != SP::OBS_DIM[1] ||
      (T.policy_corrupt != 0 && SP::OBS_CORRUPT[0] == 0) || (T.critic_corrupt != 0 && SP::OBS_CORRUPT[1] == 0))  // (a group the Spec corrupts may
    return false;                                                                                                  // run clean: the play variant)
  for (int g = 0; g < 2; ++g)
    for (int i = 0; i < SP::N_OBS[g]; ++i) {
      const ObsTab& O = g == 0 ? T.policy[i] : T.critic[i];
      const ObsSpec& Q = SP::OBS[g][i];
      if (O.kind != Q.kind || !same(O.scale, Q.scale) || !same(O.clip_lo, Q.clip_lo) || !same(O.clip_hi, Q.clip_hi) || !same(O.noise_lo, Q.noise_lo) ||
          !same(O.noise_hi, Q.noise_hi) || (O.has_noise != 0) != (Q.has_noise != 0) || O.offset != Q.offset)
        return false;
    }
  return true;
}

}  // namespace rl

#include "spec/env_specs_gen.h"
