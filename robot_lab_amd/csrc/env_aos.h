// env_aos.h - gather/scatter between the SoA simulator state and the AoS inspection buffers of the
// C-ABI (RL_BUF_ROOT_STATE, RL_BUF_JOINT_*, RL_BUF_CONTACT_TIMERS, ...).  One call per environment;
// not part of step().
#pragma once
#include "env_tables.h"

namespace rl {

struct AosPtrs {
  float *root_state, *joint_pos, *joint_vel, *ctimers, *action, *env_origin;
};

RL_FN void export_env(const KState& S, const Tables& T, const AosPtrs& A, int e) {
  const Layout ly(T.CL, T.NW, T.NBS);
  for (int f = 0; f < 13; ++f) A.root_state[e * 13 + f] = S.env_state[env_index(ly, e, ly.EF_ROOT + f, S.ept)];
  for (int a = 0; a < 3; ++a) A.env_origin[e * 3 + a] = S.env_state[env_index(ly, e, ly.EF_ORIGIN + a, S.ept)];
  for (int k = 0; k < NLANE; ++k) {
    const LaneTab& L = T.lane[k];
    for (int j = 0; j < L.nj; ++j) {
      A.joint_pos[e * T.D + L.joint_id[j]] = S.lane_state[lane_index(ly, e, k, ly.LF_Q + j, S.ept)];
      A.joint_vel[e * T.D + L.joint_id[j]] = S.lane_state[lane_index(ly, e, k, ly.LF_QD + j, S.ept)];
      A.action[e * T.D + L.joint_id[j]] = S.lane_state[lane_index(ly, e, k, ly.LF_ACT + j, S.ept)];
    }
    for (int i = 0; k == 0 && i < T.nw_used; ++i) {  // trunk joints live in the env record
      const int jt = L.joint_id[T.CL + i];
      A.joint_pos[e * T.D + jt] = S.env_state[env_index(ly, e, ly.EF_TQ + i, S.ept)];
      A.joint_vel[e * T.D + jt] = S.env_state[env_index(ly, e, ly.EF_TQD + i, S.ept)];
      A.action[e * T.D + jt] = S.env_state[env_index(ly, e, ly.EF_TACT + i, S.ept)];
    }
    for (int s = 0; s < T.NBS; ++s) {
      int b = L.slot_body[s];
      if (b < 0 || (s == 0 && !L.owns_base_body)) continue;
      for (int t = 0; t < 4; ++t) A.ctimers[(e * T.n_bodies + b) * 4 + t] = S.lane_state[lane_index(ly, e, k, ly.LF_TIMERS + s * 4 + t, S.ept)];
    }
  }
}

// any of root_state / joint_pos / joint_vel may be null
RL_FN void import_env(const KState& S, const Tables& T, const float* root_state, const float* joint_pos, const float* joint_vel, int e) {
  const Layout ly(T.CL, T.NW, T.NBS);
  if (root_state)
    for (int f = 0; f < 13; ++f) S.env_state[env_index(ly, e, ly.EF_ROOT + f, S.ept)] = root_state[e * 13 + f];
  for (int k = 0; k < NLANE; ++k) {
    const LaneTab& L = T.lane[k];
    for (int j = 0; j < L.nj; ++j) {
      if (joint_pos) S.lane_state[lane_index(ly, e, k, ly.LF_Q + j, S.ept)] = joint_pos[e * T.D + L.joint_id[j]];
      if (joint_vel) S.lane_state[lane_index(ly, e, k, ly.LF_QD + j, S.ept)] = joint_vel[e * T.D + L.joint_id[j]];
    }
    for (int i = 0; k == 0 && i < T.nw_used; ++i) {
      const int jt = L.joint_id[T.CL + i];
      if (joint_pos) S.env_state[env_index(ly, e, ly.EF_TQ + i, S.ept)] = joint_pos[e * T.D + jt];
      if (joint_vel) S.env_state[env_index(ly, e, ly.EF_TQD + i, S.ept)] = joint_vel[e * T.D + jt];
    }
  }
}

}  // namespace rl
