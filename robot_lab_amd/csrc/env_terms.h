// env_terms.h - the manager term stack of ManagerBasedRLEnv.step() (SURVEY.md section 3.2 stages 1, 3-9) as
// the second half of the lane program.  Each reward/termination/command/observation/event term cites
// the reference function it restates; the same arithmetic in fp64 numpy is oracle/env.py, which is
// pinned to the reference's own functions by tests/golden/.
#pragma once
#include "env_step.h"

namespace rl {

enum RewKind {
  REW_TRACK_LIN_VEL_XY_EXP = 0, REW_TRACK_ANG_VEL_Z_EXP, REW_LIN_VEL_Z_L2, REW_ANG_VEL_XY_L2, REW_JOINT_TORQUES_L2,
  REW_JOINT_ACC_L2, REW_JOINT_POS_LIMITS, REW_JOINT_POWER, REW_STAND_STILL, REW_JOINT_POS_PENALTY, REW_JOINT_MIRROR,
  REW_ACTION_RATE_L2, REW_UNDESIRED_CONTACTS, REW_CONTACT_FORCES, REW_FEET_CONTACT_WITHOUT_CMD, REW_FEET_HEIGHT_BODY,
  REW_UPWARD, REW_FEET_AIR_TIME, REW_FEET_AIR_TIME_VARIANCE, REW_FEET_SLIDE, REW_FEET_GAIT, REW_FLAT_ORIENTATION_L2,
  REW_IS_TERMINATED, REW_JOINT_DEVIATION_L1, REW_JOINT_VEL_L2, REW_FEET_CONTACT, REW_FEET_STUMBLE, REW_FEET_HEIGHT,
  REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP, REW_TRACK_ANG_VEL_Z_WORLD_EXP, REW_FEET_AIR_TIME_POSITIVE_BIPED
};
enum ObsKind {
  OBS_BASE_LIN_VEL = 0, OBS_BASE_ANG_VEL, OBS_PROJECTED_GRAVITY, OBS_VELOCITY_COMMANDS, OBS_JOINT_POS_REL,
  OBS_JOINT_VEL_REL, OBS_LAST_ACTION, OBS_HEIGHT_SCAN, OBS_JOINT_POS_REL_NO_WHEEL
};

// Term descriptors as the lane program consumes them: filled from the LDS tables at run time
// (wave-uniform, pinned into SGPRs).  `Spec` lets a build provide them as `static constexpr` data
// instead (compile-time term lists); measured on MI355X that bought nothing (85.0 vs 82.9 us per step:
// the term bodies, not the descriptor decode, are the cost), so only the generic form is instantiated.
struct RewDesc {
  int kind, n_idx;
  float weight, p[4];
  uint32_t joint_mask;
  uint64_t body_mask;
  int idx_a[16], idx_b[16];
};
struct ObsDesc {
  int kind, has_noise, offset;
  float scale, clip_lo, clip_hi, noise_lo, noise_hi;
};
struct GenericSpec {
  static constexpr bool generic = true;
};

template <class Ctx, class TP, class Spec = GenericSpec>
struct EnvProgram : EnvLane<Ctx, TP> {
  using Base = EnvLane<Ctx, TP>;
  using ChainTP = typename Base::ChainTP;
  static constexpr Layout LY = Base::LY;
  static constexpr int CL = TP::CL, NW = TP::NW, JX = TP::JX, NBS = TP::NBS;
  using Base::ctx; using Base::S; using Base::T; using Base::L; using Base::e; using Base::k; using Base::sub; using Base::li; using Base::Np;
  static constexpr int SUB = Base::SUB;
  static constexpr int LPE = Base::LPE;
  using Base::pos; using Base::quat; using Base::vlin; using Base::vang; using Base::q; using Base::qd; using Base::kp; using Base::kd;
  using Base::act; using Base::prev_act; using Base::tim; using Base::cf; using Base::hist_n; using Base::tau_app; using Base::qacc;
  using Base::extF; using Base::extT; using Base::base_com; using Base::wr_com;

  // command / bookkeeping registers (identical in the 4 lanes of a group)
  V3 cmd;
  float heading_target, cmd_time_left, metric_xy, metric_yaw, push_left;
  bool is_heading, is_standing;
  int level, ttype;
  V3 origin;
  long long ep_len;
  // derived articulation data [UPSTREAM B3]
  M3 Rwb;
  V3 lin_b, ang_b, grav_b, lin_w;
  float heading_w;

  RL_FN EnvProgram(Ctx& c, const KState& s) : Base(c, s) {}

  RL_FN void load_task() {
    cmd = {this->EF(LY.EF_CMD + CMD_VX), this->EF(LY.EF_CMD + CMD_VY), this->EF(LY.EF_CMD + CMD_WZ)};
    heading_target = this->EF(LY.EF_CMD + CMD_HEADING);
    cmd_time_left = this->EF(LY.EF_CMD + CMD_TIME_LEFT);
    metric_xy = this->EF(LY.EF_CMD + CMD_METRIC_XY);
    metric_yaw = this->EF(LY.EF_CMD + CMD_METRIC_YAW);
    push_left = this->EF(LY.EF_CMD + CMD_PUSH_LEFT);
    int f = S.flags[e];
    is_heading = f & 1;
    is_standing = (f >> 1) & 1;
    level = S.level[e];
    ttype = S.ttype[e];
    origin = {this->EF(LY.EF_ORIGIN + 0), this->EF(LY.EF_ORIGIN + 1), this->EF(LY.EF_ORIGIN + 2)};
    ep_len = S.ep_len[e];
  }
  RL_FN void store_task() {
    if (li != 0) return;
    this->EF(LY.EF_CMD + CMD_VX) = cmd.x; this->EF(LY.EF_CMD + CMD_VY) = cmd.y; this->EF(LY.EF_CMD + CMD_WZ) = cmd.z;
    this->EF(LY.EF_CMD + CMD_HEADING) = heading_target; this->EF(LY.EF_CMD + CMD_TIME_LEFT) = cmd_time_left;
    this->EF(LY.EF_CMD + CMD_METRIC_XY) = metric_xy; this->EF(LY.EF_CMD + CMD_METRIC_YAW) = metric_yaw; this->EF(LY.EF_CMD + CMD_PUSH_LEFT) = push_left;
    S.flags[e] = (is_heading ? 1 : 0) | (is_standing ? 2 : 0);
    S.level[e] = level;
    this->EF(LY.EF_ORIGIN + 0) = origin.x; this->EF(LY.EF_ORIGIN + 1) = origin.y; this->EF(LY.EF_ORIGIN + 2) = origin.z;
    S.ep_len[e] = ep_len;
    S.command_out[e * 3 + 0] = cmd.x; S.command_out[e * 3 + 1] = cmd.y; S.command_out[e * 3 + 2] = cmd.z;
  }

  RL_FN void derive() {
    Rwb = quat_to_mat(quat);
    V3 com_w = mul(Rwb, base_com);
    lin_w = vlin + cross(vang, com_w);  // root COM velocity
    lin_b = mulT(Rwb, lin_w);
    ang_b = mulT(Rwb, vang);
    grav_b = mulT(Rwb, V3{0.f, 0.f, -1.f});
    heading_w = atan2f(Rwb.r1.x, Rwb.r0.x);
  }

  RL_FN float U(uint32_t stream, uint32_t idx, float lo, float hi) const {
    return uniform_range(S.seed, (uint32_t)e, S.step_counter, stream, idx, lo, hi);
  }

  // UniformVelocityCommand._resample_command [UPSTREAM B7] + threshold (VEL/mdp/commands.py:43-47)
  RL_FN void resample_command(uint32_t stream, uint32_t idx) {
    float vx = U(stream, idx + 0, T.cmd_range[0][0], T.cmd_range[0][1]);
    float vy = U(stream, idx + 1, T.cmd_range[1][0], T.cmd_range[1][1]);
    float wz = U(stream, idx + 2, T.cmd_range[2][0], T.cmd_range[2][1]);
    float hd = U(stream, idx + 3, T.cmd_range[3][0], T.cmd_range[3][1]);
    bool ih = U(stream, idx + 4, 0.f, 1.f) <= T.cmd_rel_heading;
    bool is = U(stream, idx + 5, 0.f, 1.f) <= T.cmd_rel_standing;
    float keep = fsqrt(vx * vx + vy * vy) > T.cmd_small_threshold ? 1.f : 0.f;
    cmd = {vx * keep, vy * keep, wz};
    if (T.cmd_heading) { heading_target = hd; is_heading = ih; }
    is_standing = is;
  }

  // ---------------------------------------------------------------- reset of one env (all 4 lanes) [UPSTREAM B1]
  RL_FN void reset_env(bool log_episode) {
    // curriculum: terrain_levels_vel [UPSTREAM isaaclab_tasks] (velocity_env_cfg.py:671)
    if (T.curriculum && !T.is_plane) {
      float dx = pos.x - origin.x, dy = pos.y - origin.y;
      float dist = fsqrt(dx * dx + dy * dy);
      bool up = dist > T.tile_size * 0.5f;
      bool down = (dist < fsqrt(cmd.x * cmd.x + cmd.y * cmd.y) * T.max_episode_length_s * 0.5f) && !up;
      int lv = level + (up ? 1 : 0) - (down ? 1 : 0);
      int rnd = (int)fminf(floorf(U(STREAM_RESET, IDX_LEVEL, 0.f, 1.f) * (float)T.num_rows), (float)(T.num_rows - 1));
      level = lv >= T.num_rows ? rnd : (lv < 0 ? 0 : lv);
      const float* o = S.terrain_origins + ((size_t)level * T.num_cols + ttype) * 3;
      origin = {o[0], o[1], o[2]};
    }
    // scene.reset: sensor / wrench buffers
#pragma unroll
    for (int b = 0; b < NBS; ++b) {
      tim[b][0] = tim[b][1] = tim[b][2] = tim[b][3] = 0.f;
      cf[b][0] = cf[b][1] = cf[b][2] = 0.f;
      hist_n[b][0] = hist_n[b][1] = hist_n[b][2] = 0.f;
    }
    extF = {0.f, 0.f, 0.f};
    extT = {0.f, 0.f, 0.f};
    // reset events in declaration order (velocity_env_cfg.py:316-363)
    if (T.ev_wrench) {
      extF = {U(STREAM_RESET, IDX_WRENCH + 0, T.wrench_force[0], T.wrench_force[1]), U(STREAM_RESET, IDX_WRENCH + 1, T.wrench_force[0], T.wrench_force[1]),
              U(STREAM_RESET, IDX_WRENCH + 2, T.wrench_force[0], T.wrench_force[1])};
      extT = {U(STREAM_RESET, IDX_WRENCH + 3, T.wrench_torque[0], T.wrench_torque[1]), U(STREAM_RESET, IDX_WRENCH + 4, T.wrench_torque[0], T.wrench_torque[1]),
              U(STREAM_RESET, IDX_WRENCH + 5, T.wrench_torque[0], T.wrench_torque[1])};
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      uint32_t ji = (uint32_t)((NW > 0 && L.joint_id[j] < 0) ? 0 : L.joint_id[j]);  // padding joints: q0 = qd0 = kp0 = kd0 = 0
      float qn = L.q0[j], qdn = L.qd0[j];
      if (T.ev_reset_joints) {  // reset_joints_by_scale [UPSTREAM B8]
        qn = clampf(L.q0[j] * U(STREAM_RESET, IDX_JPOS + ji, T.reset_jpos[0], T.reset_jpos[1]), L.soft_lo[j], L.soft_hi[j]);
        qdn = clampf(L.qd0[j] * U(STREAM_RESET, IDX_JVEL + ji, T.reset_jvel[0], T.reset_jvel[1]), -L.vel_limit[j], L.vel_limit[j]);
      }
      q[j] = qn;
      qd[j] = qdn;
      if (T.ev_gains) {  // randomize_actuator_gains(operation="scale") [UPSTREAM B4]
        kp[j] = L.kp0[j] * U(STREAM_RESET, IDX_KP + ji, T.gain_kp[0], T.gain_kp[1]);
        kd[j] = L.kd0[j] * U(STREAM_RESET, IDX_KD + ji, T.gain_kd[0], T.gain_kd[1]);
      }
      act[j] = 0.f;
      prev_act[j] = 0.f;
      tau_app[j] = 0.f;
      qacc[j] = 0.f;
    }
    {  // reset_root_state_uniform (VEL/mdp/events.py:205-271), non-pit branch
      float ps[6], vs[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        ps[a] = T.ev_reset_base ? U(STREAM_RESET, IDX_POSE + a, T.reset_pose[a][0], T.reset_pose[a][1]) : 0.f;
        vs[a] = T.ev_reset_base ? U(STREAM_RESET, IDX_VEL + a, T.reset_vel[a][0], T.reset_vel[a][1]) : 0.f;
      }
      pos = V3{T.default_root_pos[0], T.default_root_pos[1], T.default_root_pos[2]} + origin + V3{ps[0], ps[1], ps[2]};
      Q4 q0{T.default_root_quat[0], T.default_root_quat[1], T.default_root_quat[2], T.default_root_quat[3]};
      quat = quat_mul(q0, quat_from_euler_xyz(ps[3], ps[4], ps[5]));
      vlin = {vs[0], vs[1], vs[2]};
      vang = {vs[3], vs[4], vs[5]};
    }
    // manager resets: episode-sum log + zero, command metrics log + resample, interval timer
    for (int t = li; t < T.n_rewards; t += LPE) {
      float* p = S.ep_sums + (size_t)t * Np + e;
      if (log_episode && e < S.N) ctx.atomic_add(S.log + LOG_EP_SUM0 + t, *p);
      *p = 0.f;
    }
    if (li == 0 && log_episode && e < S.N) {
      ctx.atomic_add(S.log + LOG_RESET_COUNT, 1.0f);
      ctx.atomic_add(S.log + LOG_METRIC_XY, metric_xy);
      ctx.atomic_add(S.log + LOG_METRIC_YAW, metric_yaw);
    }
    metric_xy = 0.f;
    metric_yaw = 0.f;
    cmd_time_left = U(STREAM_RESET, IDX_CMD_TIME, T.cmd_resample[0], T.cmd_resample[1]);
    resample_command(STREAM_RESET, IDX_CMD);
    if (T.ev_push) push_left = U(STREAM_RESET, IDX_PUSH_TIME, T.push_interval[0], T.push_interval[1]);
    ep_len = 0;
  }

  // ---------------------------------------------------------------- rewards
  RL_FN bool body_bit(uint64_t mask, int slot) const {
    int b = L.slot_body[slot];
    if (b < 0) return false;
    if (slot == 0 && !L.owns_base_body) return false;
    return (mask >> b) & 1ull;
  }
  RL_FN float hist_max(int slot) const { return fmaxf(hist_n[slot][0], fmaxf(hist_n[slot][1], hist_n[slot][2])); }

  // position (base coords) and velocity relative to the root COM velocity (base coords) of body slot s
  RL_FN void body_rel(const ChainTP& C, int s, V3& relp, V3& relv) const {
    int g = L.slot_grp[s];
    V3 bp = ld3(L.slot_pos[s]);
    V3 x = bp;
    if (NW > 0) {
      M3 Rf;
      V3 pf;
      trunk_frame<TP>(C, L.grp0_depth, Rf, pf);
      x = pf + mul(Rf, bp);
    }
#pragma unroll
    for (int j = 0; j < CL; ++j)
      if (g == j + 1) x = C.p(j) + mul(C.R(j), bp);
    relp = x;
    relv = point_velocity<TP, ChainTP>(C, this->wdepth(g), g, x, SV{ang_b, cross(base_com, ang_b)}, qd);  // relative to the root COM velocity
  }

  struct RewCtx {
    float gate, cmd_norm, bv, fc_hi;
    bool terminated;
    ChainTP C;
    int sbody[NBS], jid[JX];  // jid: task joint index of the joints this lane accounts for, else -1
    float hmax[NBS], t_ca[NBS], t_cc[NBS], t_la[NBS], t_lc[NBS], q0j[JX], slo[JX], shi[JX];
  };

  // one reward term: unweighted value f (all lanes of the env return the same number)
  template <class RD>
  RL_FN float reward_term(const RD& R, const RewCtx& rc) {
    const float gate = rc.gate, cmd_norm = rc.cmd_norm, bv = rc.bv;
    const bool terminated = rc.terminated;
    const ChainTP& C = rc.C;
    const int(&sbody)[NBS] = rc.sbody;
    const int(&jid)[JX] = rc.jid;
    const float(&hmax)[NBS] = rc.hmax;
    const float(&t_ca)[NBS] = rc.t_ca;
    const float(&t_cc)[NBS] = rc.t_cc;
    const float(&t_la)[NBS] = rc.t_la;
    const float(&t_lc)[NBS] = rc.t_lc;
    const float(&q0j)[JX] = rc.q0j;
    const float(&slo)[JX] = rc.slo;
    const float(&shi)[JX] = rc.shi;
    const float fc_hi = rc.fc_hi;
    auto in_mask = [&](uint64_t mask, int s) { return sbody[s] >= 0 && ((mask >> sbody[s]) & 1ull); };
    auto first_c = [&](int s) { return t_cc[s] > 0.f && t_cc[s] < fc_hi; };
    float f = 0.f;
    switch (R.kind) {
        case REW_TRACK_LIN_VEL_XY_EXP: {  // VEL/mdp/rewards.py:22-35
          float ex = cmd.x - lin_b.x, ey = cmd.y - lin_b.y;
          f = expf(-(ex * ex + ey * ey) / R.p[0]) * gate;
        } break;
        case REW_TRACK_ANG_VEL_Z_EXP: {  // rewards.py:38-48
          float ez = cmd.z - ang_b.z;
          f = expf(-(ez * ez) / R.p[0]) * gate;
        } break;
        case REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP: {  // rewards.py:51-66: root COM velocity in the yaw-only frame
          const float cy = cosf(heading_w), sy = sinf(heading_w);
          float ex = cmd.x - (cy * lin_w.x + sy * lin_w.y), ey = cmd.y - (-sy * lin_w.x + cy * lin_w.y);
          f = expf(-(ex * ex + ey * ey) / R.p[0]) * gate;
        } break;
        case REW_TRACK_ANG_VEL_Z_WORLD_EXP: {  // rewards.py:69-78
          float ez = cmd.z - vang.z;
          f = expf(-(ez * ez) / R.p[0]) * gate;
        } break;
        case REW_FEET_AIR_TIME_POSITIVE_BIPED: {  // rewards.py:363-383
          float nc = 0.f, mn = 1e30f;
#pragma unroll
          for (int s = 0; s < NBS; ++s) {
            if (!in_mask(R.body_mask, s)) continue;
            bool inc = t_cc[s] > 0.f;
            nc += inc ? 1.f : 0.f;
            mn = fminf(mn, inc ? t_cc[s] : t_ca[s]);
          }
          nc = ctx.esum(nc);
          mn = ctx.emin(mn);
          f = (nc == 1.f ? fminf(mn, R.p[0]) : 0.f) * (cmd_norm > 0.1f ? 1.f : 0.f) * gate;
        } break;
        case REW_LIN_VEL_Z_L2: f = lin_b.z * lin_b.z * gate; break;                         // rewards.py:647-653
        case REW_ANG_VEL_XY_L2: f = (ang_b.x * ang_b.x + ang_b.y * ang_b.y) * gate; break;  // rewards.py:656-662
        case REW_FLAT_ORIENTATION_L2: f = (grav_b.x * grav_b.x + grav_b.y * grav_b.y) * gate; break;  // rewards.py:678-687
        case REW_UPWARD: f = (1.f - grav_b.z) * (1.f - grav_b.z); break;                     // rewards.py:608-613
        case REW_IS_TERMINATED: f = terminated ? 1.f : 0.f; break;
        case REW_JOINT_TORQUES_L2: case REW_JOINT_ACC_L2: case REW_JOINT_VEL_L2: case REW_JOINT_POS_LIMITS:
        case REW_JOINT_POWER: case REW_JOINT_DEVIATION_L1: case REW_STAND_STILL: case REW_JOINT_POS_PENALTY:
        case REW_ACTION_RATE_L2: {
          float part = 0.f;
#pragma unroll
          for (int j = 0; j < JX; ++j) {
            bool in = (NW == 0 || jid[j] >= 0) && ((R.joint_mask >> (jid[j] & 31)) & 1u);
            float v = 0.f;
            switch (R.kind) {
              case REW_JOINT_TORQUES_L2: v = tau_app[j] * tau_app[j]; break;
              case REW_JOINT_ACC_L2: v = qacc[j] * qacc[j]; break;
              case REW_JOINT_VEL_L2: v = qd[j] * qd[j]; break;
              case REW_JOINT_POS_LIMITS: v = fmaxf(slo[j] - q[j], 0.f) + fmaxf(q[j] - shi[j], 0.f); break;
              case REW_JOINT_POWER: v = fabsf(qd[j] * tau_app[j]); break;  // rewards.py:81-90
              case REW_JOINT_DEVIATION_L1: case REW_STAND_STILL: v = fabsf(q[j] - q0j[j]); break;
              case REW_JOINT_POS_PENALTY: v = (q[j] - q0j[j]) * (q[j] - q0j[j]); break;
              case REW_ACTION_RATE_L2: v = (act[j] - prev_act[j]) * (act[j] - prev_act[j]); in = NW == 0 || jid[j] >= 0; break;
              default: break;
            }
            part += in ? v : 0.f;
          }
          f = ctx.gsum(part);
          if (R.kind == REW_STAND_STILL) f *= (cmd_norm < R.p[0] ? 1.f : 0.f) * gate;  // rewards.py:93-104
          if (R.kind == REW_JOINT_POS_PENALTY) {                                        // rewards.py:107-129
            float run = fsqrt(f);
            f = ((cmd_norm > R.p[2] || bv > R.p[1]) ? run : R.p[0] * run) * gate;
          }
        } break;
        case REW_JOINT_MIRROR: {  // rewards.py:259-278
          float qa[NLANE][CL];
#pragma unroll
          for (int kk = 0; kk < NLANE; ++kk)
#pragma unroll
            for (int j = 0; j < CL; ++j) qa[kk][j] = ctx.gshfl(q[j], kk);
          float s = 0.f;
          for (int i = 0; i < R.n_idx; ++i) {
            float va = 0.f, vb = 0.f;
#pragma unroll
            for (int kk = 0; kk < NLANE; ++kk)
#pragma unroll
              for (int j = 0; j < CL; ++j) {
                int id = T.lane[kk].joint_id[j];
                va = id == R.idx_a[i] ? qa[kk][j] : va;
                vb = id == R.idx_b[i] ? qa[kk][j] : vb;
              }
            s += (va - vb) * (va - vb);
          }
          f = s * R.p[0] * gate;
        } break;
        case REW_UNDESIRED_CONTACTS: case REW_CONTACT_FORCES: case REW_FEET_CONTACT_WITHOUT_CMD: case REW_FEET_CONTACT:
        case REW_FEET_AIR_TIME: case REW_FEET_HEIGHT_BODY: case REW_FEET_SLIDE: case REW_FEET_HEIGHT: case REW_FEET_STUMBLE: {
          float part = 0.f;
#pragma unroll
          for (int s = 0; s < NBS; ++s) {
            if (!in_mask(R.body_mask, s)) continue;
            float hm = hmax[s];
            switch (R.kind) {
              case REW_UNDESIRED_CONTACTS: part += hm > R.p[0] ? 1.f : 0.f; break;          // rewards.py:665-675
              case REW_CONTACT_FORCES: part += fmaxf(hm - R.p[0], 0.f); break;              // [UPSTREAM] contact_forces
              case REW_FEET_CONTACT_WITHOUT_CMD: case REW_FEET_CONTACT: part += first_c(s) ? 1.f : 0.f; break;
              case REW_FEET_AIR_TIME: part += first_c(s) ? t_la[s] - R.p[0] : 0.f; break;  // rewards.py:340-360
              case REW_FEET_STUMBLE: {                                                       // rewards.py:428-436
                float fx = cf[s][0], fy = cf[s][1];
                part += fsqrt(fx * fx + fy * fy) > 4.f * fabsf(cf[s][2]) ? 1.f : 0.f;
              } break;
              default: {
                V3 relp, relv;
                body_rel(C, s, relp, relv);
                if (R.kind == REW_FEET_HEIGHT_BODY) {  // rewards.py:527-554
                  float er = relp.z - R.p[0];
                  part += er * er * tanhf(R.p[1] * fsqrt(relv.x * relv.x + relv.y * relv.y));
                } else if (R.kind == REW_FEET_SLIDE) {  // rewards.py:557-587
                  part += hm > 1.0f ? fsqrt(relv.x * relv.x + relv.y * relv.y) : 0.f;
                } else {  // feet_height, world frame (rewards.py:507-524)
                  V3 pw = pos + mul(Rwb, relp);
                  V3 vw = lin_w + mul(Rwb, relv);
                  float er = pw.z - R.p[0];
                  part += er * er * tanhf(R.p[1] * fsqrt(vw.x * vw.x + vw.y * vw.y));
                }
              }
            }
          }
          f = ctx.esum(part);
          switch (R.kind) {
            case REW_UNDESIRED_CONTACTS: case REW_FEET_SLIDE: f *= gate; break;
            case REW_FEET_CONTACT_WITHOUT_CMD: f *= (cmd_norm < 0.1f ? 1.f : 0.f) * gate; break;  // rewards.py:416-425
            case REW_FEET_CONTACT: f = (f != R.p[0] ? 1.f : 0.f) * (cmd_norm > 0.1f ? 1.f : 0.f) * gate; break;
            case REW_FEET_AIR_TIME: case REW_FEET_HEIGHT_BODY: case REW_FEET_HEIGHT: f *= (cmd_norm > 0.1f ? 1.f : 0.f) * gate; break;
            case REW_FEET_STUMBLE: f = (f > 0.f ? 1.f : 0.f) * gate; break;
            default: break;
          }
        } break;
        case REW_FEET_AIR_TIME_VARIANCE: {  // rewards.py:386-397 (torch.var is unbiased)
          float n = 0.f, sa = 0.f, saa = 0.f, sc = 0.f, scc = 0.f;
#pragma unroll
          for (int s = 0; s < NBS; ++s) {
            if (!in_mask(R.body_mask, s)) continue;
            float la = fminf(t_la[s], 0.5f), lc = fminf(t_lc[s], 0.5f);
            n += 1.f; sa += la; saa += la * la; sc += lc; scc += lc * lc;
          }
          n = ctx.esum(n); sa = ctx.esum(sa); saa = ctx.esum(saa); sc = ctx.esum(sc); scc = ctx.esum(scc);
          float den = fmaxf(n - 1.f, 1.f);
          f = ((saa - sa * sa / n) / den + (scc - sc * sc / n) / den) * gate;
        } break;
        case REW_FEET_GAIT: {  // GaitReward, rewards.py:156-256
          float air[4], con[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float a = 0.f, c = 0.f;
#pragma unroll
            for (int s = 0; s < NBS; ++s)
              if (sbody[s] == R.idx_a[i]) { a = t_ca[s]; c = t_cc[s]; }
            air[i] = ctx.esum(a);
            con[i] = ctx.esum(c);
          }
          const float std = R.p[0], me2 = R.p[1] * R.p[1];
          auto sync = [&](int a, int b) {
            float da = air[a] - air[b], dc = con[a] - con[b];
            return expf(-(fminf(da * da, me2) + fminf(dc * dc, me2)) / std);
          };
          auto asyn = [&](int a, int b) {
            float d0 = air[a] - con[b], d1 = con[a] - air[b];
            return expf(-(fminf(d0 * d0, me2) + fminf(d1 * d1, me2)) / std);
          };
          float sr = sync(0, 1) * sync(2, 3);
          float ar = asyn(0, 2) * asyn(1, 3) * asyn(0, 3) * asyn(2, 1);
          f = ((cmd_norm > R.p[3] || bv > R.p[2]) ? sr * ar : 0.f) * gate;
        } break;
        default: break;
      }
    return f;
  }

  RL_FN float compute_rewards(bool terminated) {
    RewCtx rc{0.f, 0.f, 0.f, 0.f, false, this->new_chain()};
    rc.gate = clampf(-grav_b.z, 0.f, 0.7f) / 0.7f;
    rc.cmd_norm = norm(cmd);
    rc.bv = fsqrt(lin_b.x * lin_b.x + lin_b.y * lin_b.y);
    rc.terminated = terminated;
    chain_kinematics<TP>(L, q, rc.C);
    // per-slot sensor data and per-joint constants: one batch of LDS reads up front instead of dependent
    // reads inside every term
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
      int b = L.slot_body[s];
      rc.sbody[s] = ((s == 0 && !L.owns_base_body) || !this->owns_slot(s)) ? -1 : b;
      rc.hmax[s] = fmaxf(hist_n[s][0], fmaxf(hist_n[s][1], hist_n[s][2]));
      rc.t_ca[s] = tim[s][0]; rc.t_cc[s] = tim[s][1]; rc.t_la[s] = tim[s][2]; rc.t_lc[s] = tim[s][3];
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      rc.jid[j] = (NW == 0 || L.joint_own[j]) ? L.joint_id[j] : -1; rc.q0j[j] = L.q0[j]; rc.slo[j] = L.soft_lo[j]; rc.shi[j] = L.soft_hi[j];
    }
    rc.fc_hi = T.step_dt + 1e-8f;
    float total = 0.f;
    float* rstage = ctx.rew_stage();
    const float step_dt = ctx.uniform(T.step_dt);
    int n_rewards;
    if constexpr (Spec::generic) {
      n_rewards = ctx.uniform_i(T.n_rewards);
      for (int t = 0; t < n_rewards; ++t) {
        // the term descriptor is wave-uniform: pin it into SGPRs so that the dispatch is scalar branching
        const RewTab& Rl = T.rew[t];
        struct {
          int kind, n_idx;
          float weight, p[4];
          uint32_t joint_mask;
          uint64_t body_mask;
          const int32_t *idx_a, *idx_b;
        } R;
        R.kind = ctx.uniform_i(Rl.kind); R.n_idx = ctx.uniform_i(Rl.n_idx); R.weight = ctx.uniform(Rl.weight);
        R.p[0] = ctx.uniform(Rl.p[0]); R.p[1] = ctx.uniform(Rl.p[1]); R.p[2] = ctx.uniform(Rl.p[2]); R.p[3] = ctx.uniform(Rl.p[3]);
        R.joint_mask = (uint32_t)ctx.uniform_i((int)Rl.joint_mask);
        R.body_mask = (uint64_t)(uint32_t)ctx.uniform_i((int)(uint32_t)Rl.body_mask) | ((uint64_t)(uint32_t)ctx.uniform_i((int)(uint32_t)(Rl.body_mask >> 32)) << 32);
        R.idx_a = Rl.idx_a; R.idx_b = Rl.idx_b;
        float val = reward_term(R, rc) * R.weight * step_dt;  // RewardManager [UPSTREAM B2]
        total += val;
        if (li == 0) rstage[t] = val;
      }
    } else {
      n_rewards = Spec::n_rewards;
#pragma unroll
      for (int t = 0; t < Spec::n_rewards; ++t) {  // fully unrolled: every descriptor is a compile-time constant
        float val = reward_term(Spec::rewards[t], rc) * Spec::rewards[t].weight * step_dt;
        total += val;
        if (li == 0) rstage[t] = val;
      }
    }
    // per-term outputs + episode sums: staged through LDS so that each lane's read-modify-writes of
    // `ep_sums` (terms t = k, k+4, ...) are issued as one batch instead of one HBM round trip per term
    ctx.group_sync();
    {
      constexpr int NB = (MAX_T + LPE - 1) / LPE;
      float acc[NB];
      const int nrew = n_rewards;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int t = li + LPE * i;
        acc[i] = t < nrew ? S.ep_sums[(uint32_t)t * (uint32_t)Np + (uint32_t)e] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int t = li + LPE * i;
        if (t < nrew) {
          const float v = rstage[t];
          S.rew_terms[(uint32_t)t * (uint32_t)Np + (uint32_t)e] = v;
          S.ep_sums[(uint32_t)t * (uint32_t)Np + (uint32_t)e] = acc[i] + v;
        }
      }
    }
    ctx.group_sync();
    return total;
  }

  // ---------------------------------------------------------------- observations [UPSTREAM B2 / B6]
  // one observation term: value -> +noise -> clip -> scale -> its columns of the LDS-staged row
  template <class OD>
  RL_FN void obs_term(const OD& O, float* stage, bool corrupt, uint32_t noise_base, float cy, float sy, V3 scan_p) {
    auto put = [&](int col, float v) {
        if (corrupt && O.has_noise) v += U(STREAM_NOISE, noise_base + (uint32_t)col, O.noise_lo, O.noise_hi);
        stage[col] = clampf(v, O.clip_lo, O.clip_hi) * O.scale;
      };
      switch (O.kind) {
        case OBS_BASE_LIN_VEL: if (li < 3) put(O.offset + li, comp(lin_b, li)); break;
        case OBS_BASE_ANG_VEL: if (li < 3) put(O.offset + li, comp(ang_b, li)); break;
        case OBS_PROJECTED_GRAVITY: if (li < 3) put(O.offset + li, comp(grav_b, li)); break;
        case OBS_VELOCITY_COMMANDS: if (li < 3) put(O.offset + li, comp(cmd, li)); break;
        case OBS_JOINT_POS_REL: case OBS_JOINT_POS_REL_NO_WHEEL: case OBS_JOINT_VEL_REL: case OBS_LAST_ACTION:
#pragma unroll
          for (int j = 0; j < JX; ++j) {
            if (SUB > 1 && (j % SUB) != sub) continue;  // the leg's sub-lanes share its joints
            if (NW > 0 && !L.joint_own[j]) continue;    // padding / trunk joints accounted for by lane 0
            float v = O.kind == OBS_JOINT_VEL_REL ? qd[j] - L.qd0[j] : O.kind == OBS_LAST_ACTION ? act[j] : q[j] - L.q0[j];
            if (O.kind == OBS_JOINT_POS_REL_NO_WHEEL && ((T.wheel_joint_mask >> L.joint_id[j]) & 1u)) v = 0.f;
            put(O.offset + L.joint_id[j], v);
          }
          break;
        case OBS_HEIGHT_SCAN: {  // yaw-aligned grid, x fastest; z_base - hit_z - offset.  12 rays per trip so 24 8-byte loads overlap
          const int nr = ctx.uniform_i(T.scan_nx * T.scan_ny), snx = ctx.uniform_i(T.scan_nx);
          const float inv_snx = ctx.uniform(1.0f / (float)T.scan_nx);
          const float res = T.scan_res, cx0 = 0.5f * (float)(T.scan_nx - 1), cy0 = 0.5f * (float)(T.scan_ny - 1), soff = T.scan_offset;
          constexpr int RB = 12 / SUB;  // rays per lane per trip
          for (int r0 = li; r0 < nr; r0 += RB * LPE) {
            TerrainPatch tp[RB];
#pragma unroll
            for (int i = 0; i < RB; ++i) {
              int r = r0 + i * LPE;
              r = r < nr ? r : nr - 1;
              int iy = (int)(((float)r + 0.5f) * inv_snx), ix = r - iy * snx;  // exact for r < 2^20
              float lx = ((float)ix - cx0) * res, ly = ((float)iy - cy0) * res;
              tp[i] = terrain_fetch(this->u, S.terrain, scan_p.x + cy * lx - sy * ly, scan_p.y + sy * lx + cy * ly);
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
              int r = r0 + i * LPE;
              float hz;
              V3 nn;
              terrain_eval(this->u, tp[i], hz, nn);
              if (r < nr) put(O.offset + r, scan_p.z - hz - soff);
            }
          }
        } break;
        default: break;
      }
  }

  // pose of the height scanner: the root link on the quadrupeds, the torso on G1 (rides on trunk link scan_depth)
  RL_FN void scanner_pose(float& cy, float& sy, V3& scan_p) {
    if (NW == 0) {
      cy = cosf(heading_w); sy = sinf(heading_w); scan_p = pos;
      return;
    }
    ChainTP C = this->new_chain();
    chain_kinematics<TP>(L, q, C);
    M3 Rf;
    V3 pf;
    trunk_frame<TP>(C, T.scan_depth, Rf, pf);
    const M3 Rs = mul(Rwb, Rf);
    const float yaw = atan2f(Rs.r1.x, Rs.r0.x);
    cy = cosf(yaw); sy = sinf(yaw);
    scan_p = pos + mul(Rwb, pf + mul(Rf, V3{T.scan_pos[0], T.scan_pos[1], T.scan_pos[2]}));
  }

  RL_FN void write_obs(float* stage, const ObsTab* terms, int n, bool corrupt, uint32_t noise_base) {
    float cy, sy;
    V3 scan_p;
    scanner_pose(cy, sy, scan_p);
    n = ctx.uniform_i(n);
    for (int i = 0; i < n; ++i) {
      const ObsTab& Ol = terms[i];
      ObsDesc O;
      O.kind = ctx.uniform_i(Ol.kind); O.has_noise = ctx.uniform_i(Ol.has_noise); O.offset = ctx.uniform_i(Ol.offset);
      O.scale = ctx.uniform(Ol.scale); O.clip_lo = ctx.uniform(Ol.clip_lo); O.clip_hi = ctx.uniform(Ol.clip_hi);
      O.noise_lo = ctx.uniform(Ol.noise_lo); O.noise_hi = ctx.uniform(Ol.noise_hi);
      obs_term(O, stage, corrupt, noise_base, cy, sy, scan_p);
    }
  }

  RL_FN void observations() {
    derive();
    float* sp = ctx.obs_stage(0);
    float* sc = ctx.obs_stage(1);
    if constexpr (Spec::generic) {
      write_obs(sp, T.policy, T.n_policy, T.policy_corrupt != 0, 0u);
      write_obs(sc, T.critic, T.n_critic, T.critic_corrupt != 0, 1024u);
    } else {
      float cy, sy;
      V3 scan_p;
      scanner_pose(cy, sy, scan_p);
#pragma unroll
      for (int i = 0; i < Spec::n_policy; ++i) obs_term(Spec::policy[i], sp, Spec::policy_corrupt, 0u, cy, sy, scan_p);
#pragma unroll
      for (int i = 0; i < Spec::n_critic; ++i) obs_term(Spec::critic[i], sc, Spec::critic_corrupt, 1024u, cy, sy, scan_p);
    }
    ctx.flush_obs(S.obs_policy, T.policy_dim, 0);
    ctx.flush_obs(S.obs_critic, T.critic_dim, 1);
  }

  // ---------------------------------------------------------------- step()
  RL_FN void step() {
    this->load();
    // 1 ActionManager.process_action [UPSTREAM B2]; JointPosition/VelocityAction (velocity_env_cfg.py:124-126)
    float q_tgt[JX], qd_tgt[JX];
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      prev_act[j] = act[j];
      float a = (e < S.N && (NW == 0 || L.joint_id[j] >= 0)) ? S.action_in[(size_t)e * T.D + L.joint_id[j]] : 0.f;
      act[j] = a;
      float pr = clampf(a * L.a_scale[j] + L.a_off[j], L.a_lo[j], L.a_hi[j]);
      q_tgt[j] = L.action_is_vel[j] ? 0.f : pr;
      qd_tgt[j] = L.action_is_vel[j] ? pr : 0.f;
    }
    // 2 decimation loop: actuators -> physics -> contact sensor
    for (int s = 0; s < T.decimation; ++s) this->substep(q_tgt, qd_tgt);
    if (S.dbg_torque != nullptr) {
      if (sub == 0)
#pragma unroll
      for (int j = 0; j < JX; ++j) {
        if (NW > 0 && !L.joint_own[j]) continue;
        S.dbg_torque[(size_t)e * T.D + L.joint_id[j]] = tau_app[j];
        S.dbg_acc[(size_t)e * T.D + L.joint_id[j]] = qacc[j];
      }
#pragma unroll
      for (int s = 0; s < NBS; ++s) {
        int b = L.slot_body[s];
        if (b >= 0 && (s != 0 || L.owns_base_body) && this->owns_slot(s)) {
          float* o = S.dbg_cforce + ((size_t)e * T.n_bodies + b) * 3;
          o[0] = cf[s][0]; o[1] = cf[s][1]; o[2] = cf[s][2];
        }
      }
    }
    // 3 counters (the task registers are only loaded now: nothing above needs them)
    load_task();
    ep_len += 1;
    derive();
    // 4 terminations (velocity_env_cfg.py:648-664)
    bool t_timeout = T.term_time_out && ep_len >= (long long)T.max_episode_length;
    bool t_oob = false;
    if (T.term_oob && !T.is_plane) {
      float mw = (float)T.num_rows * T.tile_size + 2.f * T.border, mh = (float)T.num_cols * T.tile_size + 2.f * T.border;
      t_oob = fabsf(pos.x) > 0.5f * mw - T.oob_buffer || fabsf(pos.y) > 0.5f * mh - T.oob_buffer;
    }
    bool t_illegal = false;
    if (T.term_illegal) {
      float c = 0.f;
#pragma unroll
      for (int s = 0; s < NBS; ++s)
        if (body_bit(T.illegal_body_mask, s) && this->owns_slot(s) && hist_max(s) > T.illegal_threshold) c += 1.f;
      t_illegal = ctx.esum(c) > 0.f;
    }
    bool terminated = t_illegal, time_out = t_timeout || t_oob;
    // 5 rewards
    float rew = compute_rewards(terminated);
    if (li == 0) {
      S.reward[e] = rew;
      S.terminated[e] = terminated ? 1 : 0;
      S.time_out[e] = time_out ? 1 : 0;
    }
    // 6 reset done envs
    if (terminated || time_out) {
      if (li == 0 && e < S.N) {
        if (t_timeout) ctx.atomic_add(S.log + LOG_TERM_TIMEOUT, 1.f);
        if (t_oob) ctx.atomic_add(S.log + LOG_TERM_OOB, 1.f);
        if (t_illegal) ctx.atomic_add(S.log + LOG_TERM_ILLEGAL, 1.f);
      }
      reset_env(true);
      derive();
    }
    // 7 CommandManager.compute [UPSTREAM B7]
    {
      float max_step = T.cmd_resample[1] / T.step_dt;
      float ex = cmd.x - lin_b.x, ey = cmd.y - lin_b.y;
      metric_xy += fsqrt(ex * ex + ey * ey) / max_step;
      metric_yaw += fabsf(cmd.z - ang_b.z) / max_step;
      cmd_time_left -= T.step_dt;
      if (cmd_time_left <= 0.f) {
        cmd_time_left = U(STREAM_COMMAND, 6, T.cmd_resample[0], T.cmd_resample[1]);
        resample_command(STREAM_COMMAND, 0);
      }
      if (T.cmd_heading && is_heading)
        cmd.z = clampf(T.cmd_heading_stiffness * wrap_to_pi(heading_target - heading_w), T.cmd_range[2][0], T.cmd_range[2][1]);
      if (is_standing) cmd = {0.f, 0.f, 0.f};
      // the "pits" branch of commands.py:61-85 never fires: ROUGH_TERRAINS_CFG has no sub-terrain of that name (utils.py:27-28)
    }
    // 8 interval event: push_by_setting_velocity (velocity_env_cfg.py:366-371) [UPSTREAM B2/B8]
    if (T.ev_push) {
      push_left -= T.step_dt;
      if (push_left < 1e-6f) {
        push_left = U(STREAM_PUSH, 6, T.push_interval[0], T.push_interval[1]);
        vlin += V3{U(STREAM_PUSH, 0, T.push_vel[0][0], T.push_vel[0][1]), U(STREAM_PUSH, 1, T.push_vel[1][0], T.push_vel[1][1]),
                   U(STREAM_PUSH, 2, T.push_vel[2][0], T.push_vel[2][1])};
        vang += V3{U(STREAM_PUSH, 3, T.push_vel[3][0], T.push_vel[3][1]), U(STREAM_PUSH, 4, T.push_vel[4][0], T.push_vel[4][1]),
                   U(STREAM_PUSH, 5, T.push_vel[5][0], T.push_vel[5][1])};
      }
    }
    // 9 observations
    observations();
    this->store();
    store_task();
  }

  // ---------------------------------------------------------------- reset() entry: reset masked envs, recompute obs
  RL_FN void reset_entry() {
    this->load();
    load_task();
#pragma unroll
    for (int j = 0; j < JX; ++j) prev_act[j] = act[j];
    if (S.reset_mask == nullptr || S.reset_mask[e]) reset_env(false);
    observations();
    this->store();
    store_task();
  }
};

}  // namespace rl
