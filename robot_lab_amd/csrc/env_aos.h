// env_aos.h - gather/scatter between the SoA simulator state and the AoS inspection buffers of the
// C-ABI (RL_BUF_ROOT_STATE, RL_BUF_JOINT_*, RL_BUF_CONTACT_TIMERS, RL_BUF_TASK_STATE, ...).  One call per
// environment; not part of step().  export_env / commit_env are exact inverses over the state step() carries
// between calls (include/rl_env.h: rl_env_export_state / rl_env_commit_state).
#pragma once
#include "env_tables.h"

namespace rl {

struct AosPtrs {
  float *root_state, *joint_pos, *joint_vel, *ctimers, *action, *env_origin;
  float *task_state;  // [N][TASK_NF]
  float *gains;       // [N][2][D]
};

// one row of RL_BUF_TASK_STATE (include/rl_env.h rl_task_state_field)
enum { TS_CMD = 0, TS_HEADING = 3, TS_CMD_TIME = 4, TS_METRIC_XY = 5, TS_METRIC_YAW = 6, TS_PUSH = 7, TS_IS_HEADING = 8, TS_IS_STANDING = 9, TS_EXT_F = 10, TS_EXT_T = 13, TASK_NF = 16 };

RL_FN void export_env(const KState& S, const Tables& T, const AosPtrs& A, int e) {
  const Layout ly(T.CL, T.NW, T.NBS);
  auto EF = [&](int f) -> float& { return S.env_state[env_index(ly, e, f, S.ept)]; };
  auto LF = [&](int k, int f) -> float& { return S.lane_state[lane_index(ly, e, k, f, S.ept)]; };
  for (int f = 0; f < 13; ++f) A.root_state[e * 13 + f] = EF(ly.EF_ROOT + f);
  for (int a = 0; a < 3; ++a) A.env_origin[e * 3 + a] = EF(ly.EF_ORIGIN + a);
  float* ts = A.task_state + (size_t)e * TASK_NF;
  ts[TS_CMD + 0] = EF(ly.EF_CMD + CMD_VX); ts[TS_CMD + 1] = EF(ly.EF_CMD + CMD_VY); ts[TS_CMD + 2] = EF(ly.EF_CMD + CMD_WZ);
  ts[TS_HEADING] = EF(ly.EF_CMD + CMD_HEADING); ts[TS_CMD_TIME] = EF(ly.EF_CMD + CMD_TIME_LEFT);
  ts[TS_METRIC_XY] = EF(ly.EF_CMD + CMD_METRIC_XY); ts[TS_METRIC_YAW] = EF(ly.EF_CMD + CMD_METRIC_YAW); ts[TS_PUSH] = EF(ly.EF_CMD + CMD_PUSH_LEFT);
  ts[TS_IS_HEADING] = (S.flags[e] & 1) ? 1.f : 0.f; ts[TS_IS_STANDING] = (S.flags[e] & 2) ? 1.f : 0.f;
  for (int a = 0; a < 6; ++a) ts[TS_EXT_F + a] = EF(ly.EF_WRENCH + a);
  float* gp = A.gains + (size_t)e * 2 * T.D;
  for (int k = 0; k < NLANE; ++k) {
    const LaneTab& L = T.lane[k];
    for (int j = 0; j < L.nj; ++j) {
      const int jt = L.joint_id[j];
      A.joint_pos[e * T.D + jt] = LF(k, ly.LF_Q + j);
      A.joint_vel[e * T.D + jt] = LF(k, ly.LF_QD + j);
      A.action[e * T.D + jt] = LF(k, ly.LF_ACT + j);
      gp[jt] = LF(k, ly.LF_KP + j);
      gp[T.D + jt] = LF(k, ly.LF_KD + j);
    }
    for (int i = 0; k == 0 && i < T.nw_used; ++i) {  // trunk joints live in the env record
      const int jt = L.joint_id[T.CL + i];
      A.joint_pos[e * T.D + jt] = EF(ly.EF_TQ + i);
      A.joint_vel[e * T.D + jt] = EF(ly.EF_TQD + i);
      A.action[e * T.D + jt] = EF(ly.EF_TACT + i);
      gp[jt] = EF(ly.EF_TKP + i);
      gp[T.D + jt] = EF(ly.EF_TKD + i);
    }
    for (int s = 0; s < T.NBS; ++s) {
      int b = L.slot_body[s];
      if (b < 0 || (s == 0 && !L.owns_base_body)) continue;
      for (int t = 0; t < 4; ++t) A.ctimers[(e * T.n_bodies + b) * 4 + t] = LF(k, ly.LF_TIMERS + s * 4 + t);
    }
  }
}

RL_FN void commit_env(const KState& S, const Tables& T, const AosPtrs& A, int e) {
  const Layout ly(T.CL, T.NW, T.NBS);
  auto EF = [&](int f) -> float& { return S.env_state[env_index(ly, e, f, S.ept)]; };
  auto LF = [&](int k, int f) -> float& { return S.lane_state[lane_index(ly, e, k, f, S.ept)]; };
  for (int f = 0; f < 13; ++f) EF(ly.EF_ROOT + f) = A.root_state[e * 13 + f];
  for (int a = 0; a < 3; ++a) EF(ly.EF_ORIGIN + a) = A.env_origin[e * 3 + a];
  const float* ts = A.task_state + (size_t)e * TASK_NF;
  EF(ly.EF_CMD + CMD_VX) = ts[TS_CMD + 0]; EF(ly.EF_CMD + CMD_VY) = ts[TS_CMD + 1]; EF(ly.EF_CMD + CMD_WZ) = ts[TS_CMD + 2];
  EF(ly.EF_CMD + CMD_HEADING) = ts[TS_HEADING]; EF(ly.EF_CMD + CMD_TIME_LEFT) = ts[TS_CMD_TIME];
  EF(ly.EF_CMD + CMD_METRIC_XY) = ts[TS_METRIC_XY]; EF(ly.EF_CMD + CMD_METRIC_YAW) = ts[TS_METRIC_YAW]; EF(ly.EF_CMD + CMD_PUSH_LEFT) = ts[TS_PUSH];
  S.flags[e] = (ts[TS_IS_HEADING] != 0.f ? 1 : 0) | (ts[TS_IS_STANDING] != 0.f ? 2 : 0);
  S.command_out[e * 3 + 0] = ts[TS_CMD + 0]; S.command_out[e * 3 + 1] = ts[TS_CMD + 1]; S.command_out[e * 3 + 2] = ts[TS_CMD + 2];
  for (int a = 0; a < 6; ++a) EF(ly.EF_WRENCH + a) = ts[TS_EXT_F + a];
  const float* gp = A.gains + (size_t)e * 2 * T.D;
  for (int k = 0; k < NLANE; ++k) {
    const LaneTab& L = T.lane[k];
    for (int j = 0; j < L.nj; ++j) {
      const int jt = L.joint_id[j];
      LF(k, ly.LF_Q + j) = A.joint_pos[e * T.D + jt];
      LF(k, ly.LF_QD + j) = A.joint_vel[e * T.D + jt];
      LF(k, ly.LF_ACT + j) = A.action[e * T.D + jt];
      LF(k, ly.LF_KP + j) = gp[jt];
      LF(k, ly.LF_KD + j) = gp[T.D + jt];
    }
    for (int i = 0; k == 0 && i < T.nw_used; ++i) {
      const int jt = L.joint_id[T.CL + i];
      EF(ly.EF_TQ + i) = A.joint_pos[e * T.D + jt];
      EF(ly.EF_TQD + i) = A.joint_vel[e * T.D + jt];
      EF(ly.EF_TACT + i) = A.action[e * T.D + jt];
      EF(ly.EF_TKP + i) = gp[jt];
      EF(ly.EF_TKD + i) = gp[T.D + jt];
    }
    for (int s = 0; s < T.NBS; ++s) {
      int b = L.slot_body[s];
      if (b < 0 || (s == 0 && !L.owns_base_body)) continue;
      for (int t = 0; t < 4; ++t) LF(k, ly.LF_TIMERS + s * 4 + t) = A.ctimers[(e * T.n_bodies + b) * 4 + t];
    }
  }
}

}  // namespace rl
