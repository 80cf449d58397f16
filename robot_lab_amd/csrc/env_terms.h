// env_terms.h - the manager term stack of ManagerBasedRLEnv.step() (SURVEY.md section 3.2 stages 1, 3-9) as
// the second half of the lane program.  Each reward/termination/command/observation/event term cites
// the reference function it restates; the same arithmetic in fp64 numpy is oracle/env.py, which is
// pinned to the reference's own functions by tests/golden/.
#pragma once
#include "env_spec.h"
#include "env_step.h"

namespace rl {

enum RewKind {
  REW_TRACK_LIN_VEL_XY_EXP = 0, REW_TRACK_ANG_VEL_Z_EXP, REW_LIN_VEL_Z_L2, REW_ANG_VEL_XY_L2, REW_JOINT_TORQUES_L2,
  REW_JOINT_ACC_L2, REW_JOINT_POS_LIMITS, REW_JOINT_POWER, REW_STAND_STILL, REW_JOINT_POS_PENALTY, REW_JOINT_MIRROR,
  REW_ACTION_RATE_L2, REW_UNDESIRED_CONTACTS, REW_CONTACT_FORCES, REW_FEET_CONTACT_WITHOUT_CMD, REW_FEET_HEIGHT_BODY,
  REW_UPWARD, REW_FEET_AIR_TIME, REW_FEET_AIR_TIME_VARIANCE, REW_FEET_SLIDE, REW_FEET_GAIT, REW_FLAT_ORIENTATION_L2,
  REW_IS_TERMINATED, REW_JOINT_DEVIATION_L1, REW_JOINT_VEL_L2, REW_FEET_CONTACT, REW_FEET_STUMBLE, REW_FEET_HEIGHT,
  REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP, REW_TRACK_ANG_VEL_Z_WORLD_EXP, REW_FEET_AIR_TIME_POSITIVE_BIPED,
  REW_HANDSTAND_FEET_HEIGHT_EXP, REW_HANDSTAND_FEET_ON_AIR, REW_HANDSTAND_FEET_AIR_TIME, REW_HANDSTAND_ORIENTATION_L2, REW_BASE_HEIGHT_L2, REW_WHEEL_VEL_PENALTY,
  REW_FEET_DISTANCE_Y_EXP, REW_FEET_DISTANCE_XY_EXP, REW_ACTION_MIRROR, REW_ACTION_SYNC
};
enum ObsKind {
  OBS_BASE_LIN_VEL = 0, OBS_BASE_ANG_VEL, OBS_PROJECTED_GRAVITY, OBS_VELOCITY_COMMANDS, OBS_JOINT_POS_REL,
  OBS_JOINT_VEL_REL, OBS_LAST_ACTION, OBS_HEIGHT_SCAN, OBS_JOINT_POS_REL_NO_WHEEL
};

// ---- reward terms: tables, env scalars, and the evaluation of ONE term by ONE lane (used by the lane program and by the terms
// kernel of the split path) ------------------------------------------------------------------------------------------------
enum { JS_TAU2 = 0, JS_ACC2, JS_QD2, JS_LIMIT, JS_POWER, JS_DEV1, JS_DEV2, JS_DA2, JS_Q, JS_ABSQD, JS_ROWS };  // = env_tables.h REW_JS_ROWS
enum { BT_HMAX = 0, BT_CA, BT_CC, BT_LA, BT_LC, BT_FX, BT_FY, BT_FZ, BT_PX, BT_PY, BT_PZ, BT_VX, BT_VY, BT_VZ, BT_NF };  // = REW_BT_NF
static_assert(JS_ROWS == REW_JS_ROWS && BT_NF == REW_BT_NF && BT_FX == REW_BT_NS, "reward tables: LDS sizing in env_tables.h");

struct RewEnv {
  float gate, cmd_norm, bv, fc_hi, moving;
  bool terminated;
  const float* JT;  // [JS_ROWS][D]
  const float* BT;  // body rows: BT_FX.. only for the bodies of ext_mask (row(b), env_tables.h rew_bt_row)
  uint64_t ext_mask;
  RL_FN const float* row(int b) const { return BT + rew_bt_row(ext_mask, b); }
  int D;
  const float* action;  // this env's row of the action buffer (the raw action of this step, [D]); nullptr: a padding env (all zero)
  // the env's own scalars a term may read (the lane program's registers, or the record of the split path)
  V3 cmd, lin_b, ang_b, lin_w, vang, grav_b, pos;
  float yaw_c, yaw_s;
  M3 Rwb;
};

// "scalar" kinds: functions of the env's own scalars (velocities, gravity vector, command) and the term's parameters - no tables
RL_FN constexpr bool is_scalar_reward_kind(int kind) {
  return kind == REW_TRACK_LIN_VEL_XY_EXP || kind == REW_TRACK_ANG_VEL_Z_EXP || kind == REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP ||
         kind == REW_TRACK_ANG_VEL_Z_WORLD_EXP || kind == REW_LIN_VEL_Z_L2 || kind == REW_ANG_VEL_XY_L2 || kind == REW_FLAT_ORIENTATION_L2 ||
         kind == REW_UPWARD || kind == REW_IS_TERMINATED || kind == REW_HANDSTAND_ORIENTATION_L2;
}
// unweighted value of a scalar-kind term (the same arithmetic as the corresponding cases of term_value, which it serves too)
RL_FN float scalar_term_value(const RewTab& R, const RewEnv& E) {
  const float gate = E.gate;
  float f = 0.f;
  switch (R.kind) {
    case REW_TRACK_LIN_VEL_XY_EXP: {  // VEL/mdp/rewards.py:22-35
      float ex = E.cmd.x - E.lin_b.x, ey = E.cmd.y - E.lin_b.y;
      f = fexp(-(ex * ex + ey * ey) * frcp(R.p[0])) * gate;
    } break;
    case REW_TRACK_ANG_VEL_Z_EXP: {  // rewards.py:38-48
      float ez = E.cmd.z - E.ang_b.z;
      f = fexp(-(ez * ez) * frcp(R.p[0])) * gate;
    } break;
    case REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP: {  // rewards.py:51-66: root COM velocity in the yaw-only frame
      float ex = E.cmd.x - (E.yaw_c * E.lin_w.x + E.yaw_s * E.lin_w.y), ey = E.cmd.y - (-E.yaw_s * E.lin_w.x + E.yaw_c * E.lin_w.y);
      f = fexp(-(ex * ex + ey * ey) * frcp(R.p[0])) * gate;
    } break;
    case REW_TRACK_ANG_VEL_Z_WORLD_EXP: {  // rewards.py:69-78
      float ez = E.cmd.z - E.vang.z;
      f = fexp(-(ez * ez) * frcp(R.p[0])) * gate;
    } break;
    case REW_LIN_VEL_Z_L2: f = E.lin_b.z * E.lin_b.z * gate; break;                         // rewards.py:647-653
    case REW_ANG_VEL_XY_L2: f = (E.ang_b.x * E.ang_b.x + E.ang_b.y * E.ang_b.y) * gate; break;  // rewards.py:656-662
    case REW_FLAT_ORIENTATION_L2: f = (E.grav_b.x * E.grav_b.x + E.grav_b.y * E.grav_b.y) * gate; break;  // rewards.py:678-687
    case REW_UPWARD: f = (1.f - E.grav_b.z) * (1.f - E.grav_b.z); break;                     // rewards.py:608-613
    case REW_IS_TERMINATED: f = E.terminated ? 1.f : 0.f; break;
    case REW_HANDSTAND_ORIENTATION_L2: {  // config/others/unitree_a1_handstand/env/rewards.py:50-59
      const float dx = E.grav_b.x - R.p[0], dy = E.grav_b.y - R.p[1], dz = E.grav_b.z - R.p[2];
      f = dx * dx + dy * dy + dz * dz;
    } break;
    default: break;
  }
  return f;
}

// The term's descriptor.  -DRL_PIN_DESC reads it from the LDS table image into REGISTERS before the evaluation diverges by kind (left
// to itself the compiler sinks every field's ds_read into the case that uses it: each of the ~16 kinds a wavefront walks through
// then starts with its own LDS round trip + s_waitcnt).  NOT the default: measured 42.85 vs 42.93 us on A1 Rough 4096 (noise), and
// the eleven pinned registers at the kernel's pressure peak were enough to bring the reload-under-a-narrowed-EXEC miscompile
// back on the four-wavefront 3-joint variant (profiles/r03d_pin_desc_miscompile.txt: commands / heading flags of ~1 % of the envs).
#if defined(__HIP_DEVICE_COMPILE__) && defined(RL_PIN_DESC)
#define RL_PIN_REG(x) asm volatile("" : "+v"(x))
#else
#define RL_PIN_REG(x) ((void)0)
#endif
RL_FN RewTab load_rew_desc(const RewTab& src) {
  RewTab R = src;
  RL_PIN_REG(R.kind); RL_PIN_REG(R.weight); RL_PIN_REG(R.p[0]); RL_PIN_REG(R.p[1]); RL_PIN_REG(R.p[2]); RL_PIN_REG(R.p[3]);
  RL_PIN_REG(R.joint_mask); RL_PIN_REG(R.n_idx); RL_PIN_REG(R.body_mask); RL_PIN_REG(R.idx_off); RL_PIN_REG(R.row);
  return R;
}

// unweighted value of one term, evaluated by ONE lane (`R`: a register copy of the descriptor, load_rew_desc)
template <class TabT>
RL_FN float term_value(const TabT& T, const Uni& u, const float* __restrict__ terrain, const RewTab& R, const RewEnv& E) {
  const float gate = E.gate, cmd_norm = E.cmd_norm, bv = E.bv, moving = E.moving;
  const float* BT = E.BT;
  const int32_t* ia = T.idx_pool_a + R.idx_off;
  const int32_t* ib = T.idx_pool_b + R.idx_off;
  auto first_c = [&](const float* r) { return r[BT_CC] > 0.f && r[BT_CC] < E.fc_hi; };   // ContactSensor.compute_first_contact(step_dt)
  auto first_a = [&](const float* r) { return r[BT_CA] > 0.f && r[BT_CA] < E.fc_hi; };   // compute_first_air(step_dt)
  // joint-sum kinds share one loop: R.row = row of the joint-statistics table (host: rl_env_host.h), -1 for every other kind
  // (8 columns per trip, all LDS reads of a trip in flight together: a lone wavefront cannot hide a round trip per joint)
  float js = 0.f;
  if (R.row >= 0) {
    const float* g = E.JT + R.row * E.D;
    for (int j0 = 0; j0 < E.D; j0 += 8) {
      float v[8];
#pragma unroll
      for (int w = 0; w < 8; ++w) v[w] = g[j0 + w];  // columns >= D: reads inside the tables, dropped by the mask (the host keeps joint masks below 1 << D)
#pragma unroll
      for (int w = 0; w < 8; ++w) js += ((R.joint_mask >> (j0 + w)) & 1u) ? v[w] : 0.f;
    }
  }
  float f = 0.f;
  switch (R.kind) {
    case REW_TRACK_LIN_VEL_XY_EXP: case REW_TRACK_ANG_VEL_Z_EXP: case REW_TRACK_LIN_VEL_XY_YAW_FRAME_EXP: case REW_TRACK_ANG_VEL_Z_WORLD_EXP:
    case REW_LIN_VEL_Z_L2: case REW_ANG_VEL_XY_L2: case REW_FLAT_ORIENTATION_L2: case REW_UPWARD: case REW_IS_TERMINATED:
    case REW_HANDSTAND_ORIENTATION_L2:
      f = scalar_term_value(R, E);
      break;
    // joint sums [UPSTREAM isaaclab.envs.mdp] + rewards.py:81-90: the statistic is in the table, the mask picked the joints
    case REW_JOINT_TORQUES_L2: case REW_JOINT_ACC_L2: case REW_JOINT_VEL_L2: case REW_JOINT_POS_LIMITS: case REW_JOINT_POWER:
    case REW_JOINT_DEVIATION_L1: case REW_ACTION_RATE_L2:
      f = js;
      break;
    case REW_STAND_STILL: f = js * (cmd_norm < R.p[0] ? 1.f : 0.f) * gate; break;  // rewards.py:93-104
    case REW_JOINT_POS_PENALTY: {  // rewards.py:107-129
      float run = fsqrt(js);
      f = ((cmd_norm > R.p[2] || bv > R.p[1]) ? run : R.p[0] * run) * gate;
    } break;
    case REW_JOINT_MIRROR: {  // rewards.py:259-278
      const float* qt = E.JT + JS_Q * E.D;
      float part = 0.f;
      for (int i0 = 0; i0 < R.n_idx; i0 += 4) {  // 4 pairs per trip: index reads, then the 8 position reads, in flight together
        int a4[4], b4[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) { a4[w] = ia[i0 + w < R.n_idx ? i0 + w : i0]; b4[w] = ib[i0 + w < R.n_idx ? i0 + w : i0]; }
        float qa[4], qb[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) { qa[w] = qt[a4[w]]; qb[w] = qt[b4[w]]; }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float d = qa[w] - qb[w];
          part += i0 + w < R.n_idx ? d * d : 0.f;
        }
      }
      f = part * R.p[0] * gate;
    } break;
    // (E.action is null for PADDING envs only - e >= N, compute_rewards - whose rewards nobody reads: the 0 below is theirs.  The specialised
    // evaluation does not implement these two kinds (spec_kind_supported): a task that lists them runs this interpreter.)
    case REW_ACTION_MIRROR: {  // rewards.py:281-302: joint_mirror's form on |action| (weight 0 in every shipped cfg: straight from the action buffer)
      float part = 0.f;
      for (int i = 0; i < R.n_idx; ++i) {
        const float d = E.action != nullptr ? fabsf(E.action[ia[i]]) - fabsf(E.action[ib[i]]) : 0.f;
        part += d * d;
      }
      f = part * R.p[0] * gate;
    } break;
    case REW_ACTION_SYNC: {  // rewards.py:305-337: per joint group the (biased) variance of |action| - mean first, then the squared deviations, as the reference
      float part = 0.f;
      for (int g = 0; g < 8; ++g) {
        float s1 = 0.f, n = 0.f;
        for (int i = 0; i < R.n_idx; ++i)
          if (ib[i] == g) { s1 += E.action != nullptr ? fabsf(E.action[ia[i]]) : 0.f; n += 1.f; }
        if (n < 2.f) continue;
        const float mean = s1 / n;
        float s2 = 0.f;
        for (int i = 0; i < R.n_idx; ++i)
          if (ib[i] == g) { const float d = (E.action != nullptr ? fabsf(E.action[ia[i]]) : 0.f) - mean; s2 += d * d; }
        part += s2 / n;
      }
      f = part * R.p[0] * gate;
    } break;
    case REW_WHEEL_VEL_PENALTY: {  // rewards.py:132-153: pairs (wheel body, wheel joint)
      const float* aq = E.JT + JS_ABSQD * E.D;
      const bool running = cmd_norm > R.p[1] || bv > R.p[0];
      float part = 0.f;
      for (int i = 0; i < R.n_idx; ++i) part += (running ? (first_a(E.row(ia[i])) ? 1.f : 0.f) : 1.f) * aq[ib[i]];
      f = part;
    } break;
    case REW_FEET_GAIT: {  // GaitReward, rewards.py:156-256: product of exponentials = exponential of the sum
      float air[4], con[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* r = E.row(ia[i]);
        air[i] = r[BT_CA];
        con[i] = r[BT_CC];
      }
      const float inv_std = frcp(R.p[0]), me2 = R.p[1] * R.p[1];
      auto se = [&](float a, float b) { float d = a - b; return fminf(d * d, me2); };
      float acc = se(air[0], air[1]) + se(con[0], con[1]) + se(air[2], air[3]) + se(con[2], con[3]);
      acc += se(air[0], con[2]) + se(con[0], air[2]) + se(air[1], con[3]) + se(con[1], air[3]);
      acc += se(air[0], con[3]) + se(con[0], air[3]) + se(air[2], con[1]) + se(con[2], air[1]);
      f = ((cmd_norm > R.p[3] || bv > R.p[2]) ? fexp(-acc * inv_std) : 0.f) * gate;
    } break;
    case REW_FEET_DISTANCE_Y_EXP:
    case REW_FEET_DISTANCE_XY_EXP: {  // rewards.py:439-461, 464-505: feet (link frames) against a stance rectangle in the base frame
      float part = 0.f;
      for (int i = 0; i < R.n_idx; ++i) {
        const float* r = E.row(ia[i]);
        const float ey = ((i & 1) ? -0.5f : 0.5f) * R.p[1] - r[BT_PY];
        const float ex = R.kind == REW_FEET_DISTANCE_XY_EXP ? (i < 2 ? 0.5f : -0.5f) * R.p[2] - r[BT_PX] : 0.f;
        part += ex * ex + ey * ey;
      }
      f = fexp(-part * frcp(R.p[0])) * gate;
    } break;
    case REW_BASE_HEIGHT_L2: {  // rewards.py:616-644; the 3 x 3 base ray caster (velocity_env_cfg.py:78-85)
      float tgt = R.p[0];
      if (R.p[1] > 0.5f) {
        float hsum = 0.f;
        for (int r9 = 0; r9 < 9; ++r9) {
          const int iy = r9 / 3, ix = r9 - 3 * iy;
          const float lx = (float)(ix - 1) * 0.05f, ly = (float)(iy - 1) * 0.05f;
          float hz;
          V3 nn;
          terrain_sample(u, terrain, E.pos.x, E.pos.y, E.yaw_c * lx - E.yaw_s * ly, E.yaw_s * lx + E.yaw_c * ly, hz, nn);
          hsum += hz;
        }
        tgt += hsum * (1.0f / 9.0f);
      }
      f = (E.pos.z - tgt) * (E.pos.z - tgt) * gate;
    } break;
    default: {  // sums over the bodies of the term's body mask
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, mn = 1e30f;
      // 4 bodies per trip, their table rows read together (a lone wavefront cannot hide an LDS round trip per body)
      for (uint64_t m = R.body_mask; m != 0ull;) {
        const float* rr[4];
        bool on[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          on[w] = m != 0ull;
          rr[w] = E.row(on[w] ? __builtin_ctzll(m) : 0);
          m &= m - 1ull;  // (0 stays 0)
        }
        float hm[4], ca[4], cc[4], la[4], lc[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) { hm[w] = rr[w][BT_HMAX]; ca[w] = rr[w][BT_CA]; cc[w] = rr[w][BT_CC]; la[w] = rr[w][BT_LA]; lc[w] = rr[w][BT_LC]; }
        switch (R.kind) {
          case REW_UNDESIRED_CONTACTS:  // rewards.py:665-675
#pragma unroll
            for (int w = 0; w < 4; ++w) a0 += on[w] && hm[w] > R.p[0] ? 1.f : 0.f;
            break;
          case REW_CONTACT_FORCES:  // [UPSTREAM] contact_forces
#pragma unroll
            for (int w = 0; w < 4; ++w) a0 += on[w] ? fmaxf(hm[w] - R.p[0], 0.f) : 0.f;
            break;
          case REW_FEET_CONTACT_WITHOUT_CMD: case REW_FEET_CONTACT:  // rewards.py:416-425, 399-413
#pragma unroll
            for (int w = 0; w < 4; ++w) a0 += on[w] && cc[w] > 0.f && cc[w] < E.fc_hi ? 1.f : 0.f;
            break;
          case REW_FEET_AIR_TIME: case REW_HANDSTAND_FEET_AIR_TIME:  // rewards.py:340-360
#pragma unroll
            for (int w = 0; w < 4; ++w) a0 += on[w] && cc[w] > 0.f && cc[w] < E.fc_hi ? la[w] - R.p[0] : 0.f;
            break;
          case REW_FEET_AIR_TIME_POSITIVE_BIPED:  // rewards.py:363-383
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const bool inc = cc[w] > 0.f;
              a0 += on[w] && inc ? 1.f : 0.f;
              mn = on[w] ? fminf(mn, inc ? cc[w] : ca[w]) : mn;
            }
            break;
          case REW_HANDSTAND_FEET_ON_AIR:  // .../env/rewards.py:31-37: counts the feet that have NOT just lifted
#pragma unroll
            for (int w = 0; w < 4; ++w) a0 += on[w] && !(ca[w] > 0.f && ca[w] < E.fc_hi) ? 1.f : 0.f;
            break;
          case REW_FEET_AIR_TIME_VARIANCE:  // rewards.py:386-397 (torch.var is unbiased)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const float xa = fminf(la[w], 0.5f), xc = fminf(lc[w], 0.5f), o = on[w] ? 1.f : 0.f;
              a0 += o; a1 += o * xa; a2 += o * xa * xa; a3 += o * xc; a4 += o * xc * xc;
            }
            break;
          case REW_FEET_STUMBLE:  // rewards.py:428-436
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const float fx = rr[w][BT_FX], fy = rr[w][BT_FY], fz = rr[w][BT_FZ];
              a0 += on[w] && fsqrt(fx * fx + fy * fy) > 4.f * fabsf(fz) ? 1.f : 0.f;
            }
            break;
          default: {  // the kinds that look at a foot's position / velocity relative to the root
            V3 relp[4], relv[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              relp[w] = {rr[w][BT_PX], rr[w][BT_PY], rr[w][BT_PZ]};
              relv[w] = {rr[w][BT_VX], rr[w][BT_VY], rr[w][BT_VZ]};
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              float v = 0.f;
              if (R.kind == REW_FEET_HEIGHT_BODY) {  // rewards.py:527-554
                const float er = relp[w].z - R.p[0];
                v = er * er * ftanh(R.p[1] * fsqrt(relv[w].x * relv[w].x + relv[w].y * relv[w].y));
              } else if (R.kind == REW_FEET_SLIDE) {  // rewards.py:557-587
                v = hm[w] > 1.0f ? fsqrt(relv[w].x * relv[w].x + relv[w].y * relv[w].y) : 0.f;
              } else if (R.kind == REW_FEET_HEIGHT) {  // feet_height, world frame (rewards.py:507-524)
                const V3 vw = E.lin_w + mul(E.Rwb, relv[w]);
                const float er = E.pos.z + dot(E.Rwb.r2, relp[w]) - R.p[0];
                v = er * er * ftanh(R.p[1] * fsqrt(vw.x * vw.x + vw.y * vw.y));
              } else if (R.kind == REW_HANDSTAND_FEET_HEIGHT_EXP) {  // .../env/rewards.py:18-28
                const float dz = E.pos.z + dot(E.Rwb.r2, relp[w]) - R.p[1];
                v = dz * dz;
              }
              a0 += on[w] ? v : 0.f;
            }
          } break;
        }
      }
      switch (R.kind) {
        case REW_UNDESIRED_CONTACTS: f = a0 * gate; break;
        case REW_CONTACT_FORCES: f = a0; break;
        case REW_FEET_CONTACT_WITHOUT_CMD: f = a0 * (cmd_norm < 0.1f ? 1.f : 0.f) * gate; break;
        case REW_FEET_CONTACT: f = (a0 != R.p[0] ? 1.f : 0.f) * moving * gate; break;
        case REW_FEET_AIR_TIME: f = a0 * moving * gate; break;
        case REW_HANDSTAND_FEET_AIR_TIME: f = a0; break;
        case REW_FEET_STUMBLE: f = (a0 > 0.f ? 1.f : 0.f) * gate; break;
        case REW_FEET_HEIGHT_BODY: case REW_FEET_HEIGHT: f = a0 * moving * gate; break;
        case REW_FEET_SLIDE: f = a0 * gate; break;
        case REW_FEET_AIR_TIME_POSITIVE_BIPED: f = (a0 == 1.f ? fminf(mn, R.p[0]) : 0.f) * moving * gate; break;
        case REW_HANDSTAND_FEET_HEIGHT_EXP: f = fexp(-a0 * frcp(R.p[0])); break;
        case REW_HANDSTAND_FEET_ON_AIR: f = a0 == 0.f ? 1.f : 0.f; break;
        case REW_FEET_AIR_TIME_VARIANCE: {
          const float inv_n = frcp(a0), inv_den = frcp(fmaxf(a0 - 1.f, 1.f));
          f = ((a2 - a1 * a1 * inv_n) + (a4 - a3 * a3 * inv_n)) * inv_den * gate;
        } break;
        default: break;
      }
    } break;
  }
  return f;
}


template <class Ctx, class TP, class SP = NoSpec>
struct EnvProgram : EnvLane<Ctx, TP, SP> {
  using Base = EnvLane<Ctx, TP, SP>;
  using ChainTP = typename Base::ChainTP;
  static constexpr Layout LY = Base::LY;
  static constexpr int CL = TP::CL, NW = TP::NW, JX = TP::JX, NBS = TP::NBS;
  using Base::ctx; using Base::S; using Base::T; using Base::L; using Base::e; using Base::k; using Base::sub; using Base::li; using Base::Np;
  static constexpr int SUB = Base::SUB;
  static constexpr int LPE = Base::LPE;
  using Base::pos; using Base::quat; using Base::vlin; using Base::vang; using Base::q; using Base::qd; using Base::kp; using Base::kd;
  using Base::act; using Base::prev_act; using Base::tim; using Base::cf; using Base::hist_n; using Base::tau_app; using Base::qacc;
  using Base::extF; using Base::extT; using Base::base_com; using Base::wr_com;

  // command / bookkeeping registers (identical in all lanes of an env)
  V3 cmd;
  float heading_target, cmd_time_left, metric_xy, metric_yaw, push_left;
  // The episode log's atomic adds of a done env go out at the very END of step() (flush_log), from registers: nothing waits for them there.
  // Issued where the values arise (reward write-back, reset) they sit in front of the observation stage's loads in the wavefront's memory
  // queue, and the wait for those loads is a wait for the adds - ~1.3 k ticks of a resetting wavefront when few envs reset, and the whole
  // same-address queue when many do (DDT Tita under random actions: 0.29 resets per env-step, 30 k adds on 26 addresses per launch, 49 of
  // its 106 us; profiles/r05f_reset_cost.txt).  One lane per limb keeps the immediate form (ten more live registers in a kernel that has none).
  static constexpr bool DEFER_LOG = SUB > 1;
  static constexpr int NACC_M = (MAX_T + LPE - 1) / LPE;
  float log_sum[NACC_M], log_mxy, log_myaw;
  bool is_heading, is_standing;
  int level, ttype;
  V3 origin;
  long long ep_len;
  // derived articulation data [UPSTREAM B3]
  M3 Rwb;
  V3 lin_b, ang_b, grav_b, lin_w;
  float yaw_c, yaw_s;  // cos / sin of the heading

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
    const float hn = frsqrt(fmaxf(Rwb.r0.x * Rwb.r0.x + Rwb.r1.x * Rwb.r1.x, 1e-30f));  // cos / sin of the heading without trigonometry
    yaw_c = Rwb.r0.x * hn;
    yaw_s = Rwb.r1.x * hn;
  }

  RL_FN uint64_t uniform_u64(uint64_t v) const {  // a wave-uniform 64-bit value: both halves pinned (SGPR pair on the GPU)
    return ((uint64_t)(uint32_t)ctx.uniform_i((int)(uint32_t)(v >> 32)) << 32) | (uint64_t)(uint32_t)ctx.uniform_i((int)(uint32_t)v);
  }
  RL_FN float U(uint32_t stream, uint32_t idx, float lo, float hi) const {
    return uniform_range(S.seed, (uint32_t)e, S.step_counter, stream, idx, lo, hi);
  }

  // UniformVelocityCommand._resample_command [UPSTREAM B7] + threshold (VEL/mdp/commands.py:43-47), from its six uniforms (indices idx .. idx + 5
  // of the stream)
  RL_FN void apply_command_draws(const float (&uw)[6]) {
    // ranges: the table's, or the live ones of the command_levels_* curricula (VEL/mdp/curriculums.py:21-94) - read under ONE branch, so
    // that without the curricula (every shipped cfg) the table words and the draws are straight-line code
    float rx0 = T.cmd_range[0][0], rx1 = T.cmd_range[0][1], ry0 = T.cmd_range[1][0], ry1 = T.cmd_range[1][1], rz0 = T.cmd_range[2][0], rz1 = T.cmd_range[2][1];
    float rh0 = T.cmd_range[3][0], rh1 = T.cmd_range[3][1], rel_h = T.cmd_rel_heading, rel_s = T.cmd_rel_standing, small = T.cmd_small_threshold;
    int use_heading = T.cmd_heading;
    rl_pin(rx0); rl_pin(rx1); rl_pin(ry0); rl_pin(ry1); rl_pin(rz0); rl_pin(rz1); rl_pin(rh0); rl_pin(rh1); rl_pin(rel_h); rl_pin(rel_s); rl_pin(small); rl_pin(use_heading);
    if (T.cur_lin != 0 || T.cur_ang != 0) {
      const float* lv = S.cmd_levels;
      if (T.cur_lin) { rx0 = lv[CL_LIN_X]; rx1 = lv[CL_LIN_X + 1]; ry0 = lv[CL_LIN_Y]; ry1 = lv[CL_LIN_Y + 1]; }
      if (T.cur_ang) { rz0 = lv[CL_ANG_Z]; rz1 = lv[CL_ANG_Z + 1]; }
    }
    const float vx = lerp_draw(rx0, rx1, uw[0]), vy = lerp_draw(ry0, ry1, uw[1]), wz = lerp_draw(rz0, rz1, uw[2]), hd = lerp_draw(rh0, rh1, uw[3]);
    const bool ih = lerp_draw(0.f, 1.f, uw[4]) <= rel_h, is = lerp_draw(0.f, 1.f, uw[5]) <= rel_s;
    const float keep = fsqrt(vx * vx + vy * vy) > small ? 1.f : 0.f;
    cmd = {vx * keep, vy * keep, wz};
    heading_target = select1(use_heading != 0, hd, heading_target);
    is_heading = (use_heading != 0 && ih) || (use_heading == 0 && is_heading);
    is_standing = is;
  }
  RL_FN void resample_command(uint32_t stream, uint32_t idx) {
    float uw[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) uw[i] = uniform01(S.seed, (uint32_t)e, S.step_counter, stream, idx + (uint32_t)i);
    apply_command_draws(uw);
  }

  // ---------------------------------------------------------------- reset of one env (all its lanes) [UPSTREAM B1]
  // The uniforms a reset draws (stream STREAM_RESET of this env and step; indices IDX_WRENCH .. IDX_LEVEL, < RESET_RAND_WORDS) as a
  // table in LDS, computed ONCE per env by its lanes together: lane l takes Philox blocks l, l + LPE, ... of the needed ones (the
  // env-level blocks and ceil(D / 4) blocks per joint stream) and stores their four words; a draw is then `lo + (hi - lo) * RT[index]`.
  // Same numbers as one uniform_range() per draw (what a lane per limb still does): a Philox4x32-10 block is ~900 cycles of
  // quarter-rate integer multiplies, a reset evaluated ~20 (A1) / ~50 (G1) of them per lane one after the other - and in steady
  // state some env of the launch resets on EVERY step, so that tail (8 us on A1 Rough 4096, profiles/r04b_cold_vs_steady.txt) was
  // part of every step's kernel time: the launch ends with its slowest wavefront.
  RL_FN const float* reset_uniforms() {
    if constexpr (SUB == 1) return nullptr;
    else {
      float* RT = ctx.rand_tab();
      // the joint streams whose range is a single point are not drawn (reset_joints_by_scale of every shipped cfg: position x U(1, 1),
      // velocity x U(0, 0) - half of the blocks; a draw is `lo` then, whatever the table holds: UR below): the needed blocks of A1 / Go2W /
      // G1 then fit ONE round over the env's lanes (14 / 16 / 24 blocks on 16 / 16 / 32 lanes) instead of two
      const bool skip_jpos = !T.ev_reset_joints || !(T.reset_jpos[1] > T.reset_jpos[0]), skip_jvel = !T.ev_reset_joints || !(T.reset_jvel[1] > T.reset_jvel[0]);
      const int first_js = ctx.uniform_i(skip_jpos ? (skip_jvel ? 2 : 1) : 0);  // joint streams are dropped from the front only (JPOS, then JVEL): the usual case
      const int n_js = 4 - first_js;
      const int nb = (ctx.uniform_i(T.D) + 3) >> 2, n_need = 8 + n_js * nb;
      // (not unrolled: left to itself the compiler unrolled this loop in the single-wavefront kernel and not in the four-wavefront one, and
      // the two then contracted the draws' multiply-adds differently - one ulp in a reset env's heading target, caught by the bit-equality
      // canary of tests/test_gpu_canary.py; tools/isa_shape_arith.py compares the arithmetic of the kernel shapes at build time)
#pragma unroll 1
      for (int n = li; n < n_need; n += LPE) {
        int b = n < 2 ? n : 34 + (n - 2 - n_js * nb);  // blocks 0, 1: wrench; 34 .. 39: pose, velocity, command, timers, level
        if (n >= 2 && n < 2 + n_js * nb) {             // 2 + 8 s + i: block i of joint stream s (IDX_JPOS, IDX_JVEL, IDX_KP, IDX_KD: 32 indices each)
          const int m = n - 2, st = first_js + (m >= nb ? 1 : 0) + (m >= 2 * nb ? 1 : 0) + (m >= 3 * nb ? 1 : 0);
          b = 2 + 8 * st + (m - (st - first_js) * nb);
        }
        float un[4];
        uniform01x4(S.seed, (uint32_t)e, S.step_counter, STREAM_RESET, (uint32_t)b, un);
        RT[4 * b + 0] = un[0]; RT[4 * b + 1] = un[1]; RT[4 * b + 2] = un[2]; RT[4 * b + 3] = un[3];
      }
      ctx.group_sync();
      return RT;
    }
  }

  // With the table of uniforms (every mapping but one lane per limb) a reset is ONE batch of LDS reads - the table words, the event ranges and
  // the lane's joint constants, pinned in registers (rl_pin) - followed by arithmetic in which event flags and single-point ranges select
  // results.  Written with a branch per event and a read per draw, every read sat in a block of its own behind the read its condition came
  // from (the compiler sinks a load into the branch that uses it): 93 reads, 77 waits, one after the other in a lone wavefront - A1: 7.5 k
  // ticks of the resetting wavefront = 4 us, G1: 12.8 k = 6.8 us (profiles/r05g_phase_clock_*_reset_env0.txt) - and a launch ends with its
  // slowest wavefront, which in steady state is one that resets an env.  Same draws, same arithmetic.
  // `sums_folded`: the reward stage has logged and zeroed the episode sums already (compute_rewards).
  RL_FN void reset_env(bool log_episode, bool sums_folded = false) {
    RL_PHASE(26, "reset.uniforms");
    const float* RT = reset_uniforms();
    RL_PHASE(27, "reset.state");
    constexpr bool TAB = SUB > 1;  // the draws are table words; else (one lane per limb) a Philox block per draw, each worth its branch
    // ---- the batch.  Uniforms: indices IDX_WRENCH .. + 5 and IDX_POSE .. IDX_LEVEL (contiguous), then four per joint
    constexpr int NHI = (int)IDX_LEVEL - (int)IDX_POSE + 1;
    float w_lo[6] = {}, w_hi[NHI] = {};
    int jid[JX];
#pragma unroll
    for (int j = 0; j < JX; ++j) jid[j] = (TP::PAD && L.joint_id[j] < 0) ? 0 : L.joint_id[j];  // padding joints: q0 = qd0 = kp0 = kd0 = 0
    if constexpr (TAB) {
#pragma unroll
      for (int i = 0; i < 6; ++i) w_lo[i] = RT[IDX_WRENCH + i];
#pragma unroll
      for (int i = 0; i < NHI; ++i) w_hi[i] = RT[IDX_POSE + i];
    }
    // the event ranges
    float rng_w[4] = {T.wrench_force[0], T.wrench_force[1], T.wrench_torque[0], T.wrench_torque[1]};
    float rng_j[8] = {T.reset_jpos[0], T.reset_jpos[1], T.reset_jvel[0], T.reset_jvel[1], T.gain_kp[0], T.gain_kp[1], T.gain_kd[0], T.gain_kd[1]};
    float rng_t[4] = {T.cmd_resample[0], T.cmd_resample[1], T.push_interval[0], T.push_interval[1]};
    float tile_half = T.tile_size * 0.5f, ep_half = T.max_episode_length_s * 0.5f;
    int n_rows = T.num_rows, n_cols = T.num_cols;
    int flags = (T.ev_wrench ? 1 : 0) | (T.ev_reset_joints ? 2 : 0) | (T.ev_gains ? 4 : 0) | (T.ev_reset_base ? 8 : 0) | (T.ev_push ? 16 : 0) | ((T.curriculum && !T.is_plane) ? 32 : 0);
    if constexpr (TAB) {
      rl_pin(w_lo); rl_pin(w_hi);
      rl_pin(rng_w); rl_pin(rng_j); rl_pin(rng_t);
      rl_pin(tile_half); rl_pin(ep_half); rl_pin(n_rows); rl_pin(n_cols); rl_pin(flags);
    }
    const bool ev_wrench = flags & 1, ev_joints = flags & 2, ev_gains = flags & 4, ev_base = flags & 8, ev_push = flags & 16, cur = flags & 32;
    // a draw from its table word (TAB) / its index (else).  (hi > lo: a single-point range is `lo` - reset_uniforms does not fill the blocks
    // of such streams, whatever the word holds)
    auto UR = [&](uint32_t idx, float word, float lo, float hi) __attribute__((always_inline)) {
      if constexpr (TAB) return hi > lo ? lerp_draw(lo, hi, word) : lo;
      else return U(STREAM_RESET, idx, lo, hi);
    };
    auto HI = [&](uint32_t idx) __attribute__((always_inline)) { return TAB ? w_hi[idx - IDX_POSE] : 0.f; };
    // ---- curriculum: terrain_levels_vel [UPSTREAM isaaclab_tasks] (velocity_env_cfg.py:671).  The origin of the new tile is the one HBM read
    // of a reset: issued first, consumed last (the root position below)
    V3 origin_new = origin;
    if (TAB || cur) {
      float dx = pos.x - origin.x, dy = pos.y - origin.y;
      float dist = fsqrt(dx * dx + dy * dy);
      bool up = dist > tile_half;
      bool down = (dist < fsqrt(cmd.x * cmd.x + cmd.y * cmd.y) * ep_half) && !up;
      int lv = level + (up ? 1 : 0) - (down ? 1 : 0);
      int rnd = (int)fminf(floorf(UR(IDX_LEVEL, HI(IDX_LEVEL), 0.f, 1.f) * (float)n_rows), (float)(n_rows - 1));
      const int lv_new = lv >= n_rows ? rnd : (lv < 0 ? 0 : lv);
      level = cur ? lv_new : level + 0;
      // (a plane has a one-tile origin table of zeros: rl_env_host.h create)
      const float* o = S.terrain_origins + (cur ? ((size_t)level * n_cols + ttype) * 3 : (size_t)0);
      const V3 ot{o[0], o[1], o[2]};
      origin_new = select3(cur, ot, origin);  // (a plane, or no curriculum: the env keeps its origin)
    }
    // scene.reset: sensor / wrench buffers
#pragma unroll
    for (int i = 0; i < Base::MAXOWN; ++i) {
      const int b = this->own[i];
      if (b < 0) continue;
      tim.st(b, F4{0.f, 0.f, 0.f, 0.f});
      cf.st(b, F4{0.f, 0.f, 0.f, 0.f});
      hist_n.st(b, F4{0.f, 0.f, 0.f, 0.f});
    }
    extF = {0.f, 0.f, 0.f};
    extT = {0.f, 0.f, 0.f};
    // reset events in declaration order (velocity_env_cfg.py:316-363)
    if (TAB || ev_wrench) {
      const V3 wf{UR(IDX_WRENCH + 0, w_lo[0], rng_w[0], rng_w[1]), UR(IDX_WRENCH + 1, w_lo[1], rng_w[0], rng_w[1]), UR(IDX_WRENCH + 2, w_lo[2], rng_w[0], rng_w[1])};
      const V3 wt{UR(IDX_WRENCH + 3, w_lo[3], rng_w[2], rng_w[3]), UR(IDX_WRENCH + 4, w_lo[4], rng_w[2], rng_w[3]), UR(IDX_WRENCH + 5, w_lo[5], rng_w[2], rng_w[3])};
      extF = select3(ev_wrench, wf, extF);
      extT = select3(ev_wrench, wt, extT);
    }
    // the joints, five at a time: a batch of their table words and the lane's constants (eleven reads per joint), then the arithmetic (all
    // thirteen joints of the six-joint-trunk instance at once were 140 live values more than its registers hold)
    constexpr int JB = 5;
    static_for<0, (JX + JB - 1) / JB>([&](auto cc) __attribute__((always_inline)) {
      constexpr int j0 = decltype(cc)::value * JB, NJ = JX - j0 < JB ? JX - j0 : JB;
      float wj[4][NJ] = {}, jq0[NJ], jqd0[NJ], jlo[NJ], jhi[NJ], jvl[NJ], jkp0[NJ], jkd0[NJ];
#pragma unroll
      for (int i = 0; i < NJ; ++i) {
        const int j = j0 + i;
        if constexpr (TAB) {
          wj[0][i] = RT[IDX_JPOS + (uint32_t)jid[j]]; wj[1][i] = RT[IDX_JVEL + (uint32_t)jid[j]];
          wj[2][i] = RT[IDX_KP + (uint32_t)jid[j]]; wj[3][i] = RT[IDX_KD + (uint32_t)jid[j]];
        }
        jq0[i] = this->L.q0[j]; jqd0[i] = this->L.qd0[j]; jlo[i] = this->L.soft_lo[j]; jhi[i] = this->L.soft_hi[j]; jvl[i] = this->L.vel_limit[j];
        jkp0[i] = this->L.kp0[j]; jkd0[i] = this->L.kd0[j];
      }
      if constexpr (TAB) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) rl_pin(wj[s4]);
        rl_pin(jq0); rl_pin(jqd0); rl_pin(jlo); rl_pin(jhi); rl_pin(jvl); rl_pin(jkp0); rl_pin(jkd0);
      }
#pragma unroll
      for (int i = 0; i < NJ; ++i) {
        const int j = j0 + i;
        const uint32_t ji = (uint32_t)jid[j];
        float qn = jq0[i], qdn = jqd0[i];
        if (TAB || ev_joints) {  // reset_joints_by_scale [UPSTREAM B8]
          const float qa = clampf(jq0[i] * UR(IDX_JPOS + ji, wj[0][i], rng_j[0], rng_j[1]), jlo[i], jhi[i]);
          const float qb = clampf(jqd0[i] * UR(IDX_JVEL + ji, wj[1][i], rng_j[2], rng_j[3]), -jvl[i], jvl[i]);
          qn = select1(ev_joints, qa, qn);
          qdn = select1(ev_joints, qb, qdn);
        }
        this->q[j] = qn;
        this->qd[j] = qdn;
        if (TAB || ev_gains) {  // randomize_actuator_gains(operation="scale") [UPSTREAM B4]
          const float ka = jkp0[i] * UR(IDX_KP + ji, wj[2][i], rng_j[4], rng_j[5]), kb = jkd0[i] * UR(IDX_KD + ji, wj[3][i], rng_j[6], rng_j[7]);
          this->kp[j] = select1(ev_gains, ka, this->kp[j]);
          this->kd[j] = select1(ev_gains, kb, this->kd[j]);
        }
        this->act[j] = 0.f;
        this->prev_act[j] = 0.f;
        this->tau_app[j] = 0.f;
        this->qacc[j] = 0.f;
      }
    });
    {  // reset_root_state_uniform (VEL/mdp/events.py:205-271), non-pit branch.  Its ranges: a batch of their own
      float rng_p[12], rng_v[12], root0[7];
#pragma unroll
      for (int a = 0; a < 6; ++a) { rng_p[2 * a] = T.reset_pose[a][0]; rng_p[2 * a + 1] = T.reset_pose[a][1]; rng_v[2 * a] = T.reset_vel[a][0]; rng_v[2 * a + 1] = T.reset_vel[a][1]; }
#pragma unroll
      for (int a = 0; a < 3; ++a) root0[a] = T.default_root_pos[a];
#pragma unroll
      for (int a = 0; a < 4; ++a) root0[3 + a] = T.default_root_quat[a];
      if constexpr (TAB) { rl_pin(rng_p); rl_pin(rng_v); rl_pin(root0); }
      float ps[6], vs[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        ps[a] = vs[a] = 0.f;
        if (TAB || ev_base) {
          const float pa = UR(IDX_POSE + a, HI(IDX_POSE + a), rng_p[2 * a], rng_p[2 * a + 1]), va = UR(IDX_VEL + a, HI(IDX_VEL + a), rng_v[2 * a], rng_v[2 * a + 1]);
          ps[a] = ev_base ? pa : 0.f;
          vs[a] = ev_base ? va : 0.f;
        }
      }
      origin = origin_new;
      pos = V3{root0[0], root0[1], root0[2]} + origin + V3{ps[0], ps[1], ps[2]};
      Q4 q0{root0[3], root0[4], root0[5], root0[6]};
      quat = quat_mul(q0, quat_from_euler_xyz(ps[3], ps[4], ps[5]));
      vlin = {vs[0], vs[1], vs[2]};
      vang = {vs[3], vs[4], vs[5]};
    }
    // manager resets: episode-sum log + zero, command metrics log + resample, interval timer
    // (command_levels_* curricula: the live ranges read by the command resampling were decided between the two launches of this step,
    // from the sums collect_cmd_levels gathered in the first one - step_head)
    if (!sums_folded) {
      RL_PHASE(28, "reset.log");
      for (int t = li; t < T.n_rewards; t += LPE) {
        float* p = S.ep_sums + (size_t)t * Np + e;
        if (log_episode && e < S.N) ctx.atomic_add(log_slot() + LOG_EP_SUM0 + t, *p);
        *p = 0.f;
      }
      RL_PHASE(27, "reset.state");
    }
    if (DEFER_LOG && sums_folded) {  // (flush_log)
      log_mxy = metric_xy;
      log_myaw = metric_yaw;
    } else if (li == 0 && log_episode && e < S.N) {
      ctx.atomic_add(log_slot() + LOG_RESET_COUNT, 1.0f);
      ctx.atomic_add(log_slot() + LOG_FRESH, 1.0f);
      ctx.atomic_add(log_slot() + LOG_METRIC_XY, metric_xy);
      ctx.atomic_add(log_slot() + LOG_METRIC_YAW, metric_yaw);
    }
    metric_xy = 0.f;
    metric_yaw = 0.f;
    cmd_time_left = UR(IDX_CMD_TIME, HI(IDX_CMD_TIME), rng_t[0], rng_t[1]);
    if constexpr (TAB) {
      float uw[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) uw[i] = HI(IDX_CMD + i);
      apply_command_draws(uw);
    } else {
      resample_command(STREAM_RESET, IDX_CMD);
    }
    if (TAB || ev_push) {
      const float pl = UR(IDX_PUSH_TIME, HI(IDX_PUSH_TIME), rng_t[2], rng_t[3]);
      push_left = select1(ev_push, pl, push_left);
    }
    ep_len = 0;
  }

  // ---------------------------------------------------------------- rewards
  RL_FN bool body_bit(uint64_t mask, int slot) const {
    int b = L.slot_body[slot];
    if (b < 0) return false;
    if (slot == 0 && !L.owns_base_body) return false;
    return (mask >> b) & 1ull;
  }
  RL_FN float hist_max(int slot) const {
    const F4 h = hist_n.ld(slot);
    return fmaxf(h.x, fmaxf(h.y, h.z));
  }

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
    relv = point_velocity<TP, ChainTP>(C, this->trunk_anc(g), g, x, SV{ang_b, cross(base_com, ang_b)}, qd);  // relative to the root COM velocity
  }

  // Reward evaluation: LANE PER TERM.  The lanes that own joints / body slots first publish two small per-env tables in LDS -
  // joint statistics (one row per statistic, one column per task joint) and a body table (contact sensor state, net force,
  // foot position / velocity relative to the root) - and then lane l of the env evaluates terms l, l + LPE, ... completely:
  // it reads its term's descriptor from the LDS table image, sums over the joints of its joint mask / the bodies of its body
  // mask from the tables and applies the term's own arithmetic.  The terms of an env are thus evaluated side by side (the wavefront
  // executes each reward KIND that occurs once, for all environments and all terms of that kind), instead of one after the other
  // with a descriptor pinned into SGPRs, a scalar dispatch and a cross-lane reduction per term (round 1: ~1100 cycles per term,
  // 20 k of a 125 k-cycle step).  Every term cites the reference function it restates; oracle/env.py has the same arithmetic in fp64.
  // `fold_done`: the env is done and this launch resets it (step(): not the head launch of a split step) - the write-back then logs the
  // final episode sums and stores zeros, from the registers that hold them; reset_env() would read them back from HBM first (a dependent
  // round trip on the resetting wavefront's critical path)
  RL_FN float compute_rewards(bool terminated, bool fold_done = false) {
    const int D = ctx.uniform_i(T.D), n_rewards = ctx.uniform_i(T.n_rewards);
    float* JT = ctx.rew_tab();
    float* BT = JT + JS_ROWS * D;
    // episode sums of the terms this lane writes back (t = li, li + LPE, ...): loaded now, consumed after the
    // terms - the HBM round trip overlaps the term arithmetic
    constexpr int NACC = (MAX_T + LPE - 1) / LPE;
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int t = li + LPE * i;
      acc[i] = t < n_rewards ? S.ep_sums[(uint32_t)t * (uint32_t)Np + (uint32_t)e] : 0.f;
    }
    // ---- publish the joint statistics (first sub-lane of a limb: its joints; limb 0 also the trunk joints) ...
#ifdef RL_ABL_NO_REW_PUBLISH
    if (false) {
#else
    if (sub == 0) {
#endif
#pragma unroll
      for (int j = 0; j < JX; ++j) {
        const int jid = (!TP::PAD || L.joint_own[j]) ? L.joint_id[j] : -1;
        if (jid < 0) continue;
        const float dq = q[j] - L.q0[j], da = act[j] - prev_act[j];
        JT[JS_TAU2 * D + jid] = tau_app[j] * tau_app[j];
        JT[JS_ACC2 * D + jid] = qacc[j] * qacc[j];
        JT[JS_QD2 * D + jid] = qd[j] * qd[j];
        JT[JS_LIMIT * D + jid] = fmaxf(L.soft_lo[j] - q[j], 0.f) + fmaxf(q[j] - L.soft_hi[j], 0.f);
        JT[JS_POWER * D + jid] = fabsf(qd[j] * tau_app[j]);
        JT[JS_DEV1 * D + jid] = fabsf(dq);
        JT[JS_DEV2 * D + jid] = dq * dq;
        JT[JS_DA2 * D + jid] = da * da;
        JT[JS_Q * D + jid] = q[j];
        JT[JS_ABSQD * D + jid] = fabsf(qd[j]);
      }
    }
    // ---- ... and the body table (the lane that owns a body slot: sensor state; position / velocity relative to the root for
    // the bodies some term looks at that way - T.rew_rel_mask)
    {
      const uint64_t rel_mask = T.rew_rel_mask;
      const uint64_t ext_mask = uniform_u64(T.rew_ext_mask);
      const bool any_rel = ctx.uniform_i((int)(rel_mask != 0ull)) != 0;
      ChainTP C = this->new_chain();
      // (trunk + limbs instance: the chain words in LDS already hold the kinematics of the final joint positions, step() stage 3)
      if (any_rel && NW == 0) chain_kinematics<TP, SP>(L, q, C);
#pragma unroll
      for (int i = 0; i < Base::MAXOWN; ++i) {
        const int s = this->own[i];
        if (s < 0) continue;
        const int b = L.slot_body[s];
        if (b < 0 || (s == 0 && !L.owns_base_body)) continue;
        float* r = BT + rew_bt_row(ext_mask, b);
        const F4 hs = hist_n.ld(s), ts = tim.ld(s);
        r[BT_HMAX] = fmaxf(hs.x, fmaxf(hs.y, hs.z));
        r[BT_CA] = ts.x; r[BT_CC] = ts.y; r[BT_LA] = ts.z; r[BT_LC] = ts.w;
        if ((ext_mask >> b) & 1ull) {
          const F4 fs = cf.ld(s);
          r[BT_FX] = fs.x; r[BT_FY] = fs.y; r[BT_FZ] = fs.z;
        }
        if (any_rel && ((rel_mask >> b) & 1ull)) {
          V3 relp, relv;
          body_rel(C, s, relp, relv);
          r[BT_PX] = relp.x; r[BT_PY] = relp.y; r[BT_PZ] = relp.z;
          r[BT_VX] = relv.x; r[BT_VY] = relv.y; r[BT_VZ] = relv.z;
        }
      }
    }
    ctx.group_sync();
    // ---- lane per term
    RL_PHASE(17, "rewards.terms");
    RewEnv E;
    E.gate = clampf(-grav_b.z, 0.f, 0.7f) * (1.0f / 0.7f);
    E.cmd_norm = norm(cmd);
    E.bv = fsqrt(lin_b.x * lin_b.x + lin_b.y * lin_b.y);
    E.fc_hi = T.step_dt + 1e-8f;
    E.moving = E.cmd_norm > 0.1f ? 1.f : 0.f;
    E.terminated = terminated;
    E.JT = JT; E.BT = BT; E.D = D;
    E.action = e < S.N ? S.action_in + (size_t)e * (size_t)D : nullptr;
    E.ext_mask = uniform_u64(T.rew_ext_mask);
    E.cmd = cmd; E.lin_b = lin_b; E.ang_b = ang_b; E.lin_w = lin_w; E.vang = vang; E.grav_b = grav_b; E.pos = pos;
    E.yaw_c = yaw_c; E.yaw_s = yaw_s; E.Rwb = Rwb;
    const float step_dt = ctx.uniform(T.step_dt);
    float* rstage = ctx.rew_stage();
    float mine = 0.f;
    // slots [0, n_main): any kind; slots [n_main, n_rewards): scalar kinds (the host's schedule, TaskTab::rew_slot)
#ifdef RL_ABL_NO_REW_TERMS
    const int n_main = 0, n_all = 0;
#else
    const int n_main = ctx.uniform_i(T.n_main), n_all = n_rewards;
#endif
    for (int sl = li; sl < n_main; sl += LPE) {
      const int t = T.rew_slot[sl];
      const RewTab R = load_rew_desc(T.rew[t]);
      const float val = term_value(T, this->u, S.terrain, R, E) * R.weight * step_dt;  // RewardManager [UPSTREAM B2]
      rstage[t] = val;
      mine += val;
    }
    for (int sl = n_main + li; sl < n_all; sl += LPE) {
      const int t = T.rew_slot[sl];
      const RewTab R = load_rew_desc(T.rew[t]);
      const float val = scalar_term_value(R, E) * R.weight * step_dt;
      rstage[t] = val;
      mine += val;
    }
    const float total = ctx.esum(mine);
    // per-term outputs + episode sums: staged through LDS so that each lane's read-modify-writes of
    // `ep_sums` (terms t = li, li + LPE, ...) are issued as one batch instead of one HBM round trip per term
    RL_PHASE(18, "rewards.writeback");
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int t = li + LPE * i;
      if (t < n_rewards) {
        const float v = rstage[t];
        S.rew_terms[(uint32_t)t * (uint32_t)Np + (uint32_t)e] = v;
        const float ns = acc[i] + v;
        S.ep_sums[(uint32_t)t * (uint32_t)Np + (uint32_t)e] = fold_done ? 0.f : ns;
        if constexpr (DEFER_LOG) log_sum[i] = ns;
        else if (fold_done && e < S.N) ctx.atomic_add(log_slot() + LOG_EP_SUM0 + t, ns);
      }
    }
    ctx.group_sync();  // the tables share LDS with the observation rows written next
    return total;
  }

  // ---------------------------------------------------------------- rewards, SPECIALISED on the task (env_spec.h)
  // The same terms - term for term the arithmetic of term_value() above - with the task's term list a constant expression: every lane
  // evaluates every term from its own registers.  Joint kinds: one pass over the lane's own joints (the sub-lanes of a limb hold them
  // redundantly; a term's joint mask is a compile-time constant per limb), one DPP sum over the four limbs per term.  Body kinds: each
  // lane over the body slots it owns (contact-sensor rows from its scratchpad), one DPP sum over the env's lanes per term.  Scalar
  // kinds: replicated.  No statistics published to LDS, no descriptor reads, no dispatch; the per-term values end in registers of
  // every lane, and the lane that writes term t back picks it with a select chain.
  template <uint64_t MASK>
  RL_FN static bool in_body_mask(int b) {  // b: a valid body index (< SP::N_BODIES)
    if constexpr (MASK == 0ull) return false;
    else if constexpr (SP::N_BODIES <= 32) return (((uint32_t)MASK >> (uint32_t)b) & 1u) != 0u;
    else return ((MASK >> (uint64_t)b) & 1ull) != 0ull;
  }
  RL_FN float compute_rewards_spec(bool terminated, bool fold_done = false) {
    constexpr int NT = SP::N_REW;
    constexpr int NACC = (NT + LPE - 1) / LPE;
    Ctx& cx = ctx;                  // (plain locals for the nested generic lambdas below: g++ does not find names that come from
    const int my_k = k, my_li = li; // using-declarations of the dependent base inside them)
    float acc[NACC];  // episode sums of the terms this lane writes back: loaded now, consumed after the terms
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int t = li + LPE * i;
      acc[i] = t < NT ? S.ep_sums[(uint32_t)t * (uint32_t)Np + (uint32_t)e] : 0.f;
    }
    RewEnv E;
    E.gate = clampf(-grav_b.z, 0.f, 0.7f) * (1.0f / 0.7f);
    E.cmd_norm = norm(cmd);
    E.bv = fsqrt(lin_b.x * lin_b.x + lin_b.y * lin_b.y);
    E.fc_hi = T.step_dt + 1e-8f;
    E.moving = E.cmd_norm > 0.1f ? 1.f : 0.f;
    E.terminated = terminated;
    E.JT = nullptr; E.BT = nullptr; E.D = SP::D; E.ext_mask = 0ull; E.action = nullptr;
    E.cmd = cmd; E.lin_b = lin_b; E.ang_b = ang_b; E.lin_w = lin_w; E.vang = vang; E.grav_b = grav_b; E.pos = pos;
    E.yaw_c = yaw_c; E.yaw_s = yaw_s; E.Rwb = Rwb;
    const float gate = E.gate, cmd_norm = E.cmd_norm, bv = E.bv, moving = E.moving, fc_hi = E.fc_hi;
    // partial sums of this lane: a0 a term's main accumulator, ax its extra ones (variance: four moments; biped: the minimum)
    float a0[NT], ax[NT][4];
    float gait_air[4] = {0.f, 0.f, 0.f, 0.f}, gait_con[4] = {0.f, 0.f, 0.f, 0.f};  // the four feet of a gait term (one such term at most is specialised)
    float wq[4] = {0.f, 0.f, 0.f, 0.f};  // |qd| of the wheel joints of a wheel_vel_penalty term (one such term at most)
    static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      a0[t] = 0.f;
      ax[t][0] = SP::REW[t].kind == REW_FEET_AIR_TIME_POSITIVE_BIPED ? 1e30f : 0.f;
      ax[t][1] = ax[t][2] = ax[t][3] = 0.f;
    });
    // ---- joint kinds: this limb's share (identical in the limb's sub-lanes)
    static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      constexpr int kd = SP::REW[t].kind;
      constexpr bool joint_kind = kd == REW_JOINT_TORQUES_L2 || kd == REW_JOINT_ACC_L2 || kd == REW_JOINT_VEL_L2 || kd == REW_JOINT_POS_LIMITS ||
                                  kd == REW_JOINT_POWER || kd == REW_JOINT_DEVIATION_L1 || kd == REW_STAND_STILL || kd == REW_JOINT_POS_PENALTY ||
                                  kd == REW_ACTION_RATE_L2;
      if constexpr (joint_kind) {
        constexpr uint32_t m0 = spec_lmask<SP>(t, 0), m1 = spec_lmask<SP>(t, 1), m2 = spec_lmask<SP>(t, 2), m3 = spec_lmask<SP>(t, 3);
        constexpr uint32_t m_any = m0 | m1 | m2 | m3, m_all = m0 & m1 & m2 & m3;
        const uint32_t lm = my_k == 0 ? m0 : (my_k == 1 ? m1 : (my_k == 2 ? m2 : m3));
        float part = 0.f;
        static_for<0, JX>([&](auto jc) __attribute__((always_inline)) {
          constexpr int j = decltype(jc)::value;
          if constexpr (((m_any >> j) & 1u) != 0u) {
            float st;  // (this->: see above)
            if constexpr (kd == REW_JOINT_TORQUES_L2) st = this->tau_app[j] * this->tau_app[j];
            else if constexpr (kd == REW_JOINT_ACC_L2) st = this->qacc[j] * this->qacc[j];
            else if constexpr (kd == REW_JOINT_VEL_L2) st = this->qd[j] * this->qd[j];
            else if constexpr (kd == REW_JOINT_POS_LIMITS) st = fmaxf(this->L.soft_lo[j] - this->q[j], 0.f) + fmaxf(this->q[j] - this->L.soft_hi[j], 0.f);
            else if constexpr (kd == REW_JOINT_POWER) st = fabsf(this->qd[j] * this->tau_app[j]);
            else if constexpr (kd == REW_JOINT_DEVIATION_L1 || kd == REW_STAND_STILL) st = fabsf(this->q[j] - this->L.q0[j]);
            else if constexpr (kd == REW_JOINT_POS_PENALTY) { const float dq = this->q[j] - this->L.q0[j]; st = dq * dq; }
            else { const float da = this->act[j] - this->prev_act[j]; st = da * da; }
            if constexpr (((m_all >> j) & 1u) != 0u) part += st;
            else part += ((lm >> j) & 1u) != 0u ? st : 0.f;
          }
        });
        a0[t] = part;
      } else if constexpr (kd == REW_WHEEL_VEL_PENALTY) {  // rewards.py:132-153: pairs (wheel body, wheel joint) - here the joints' |qd|, by the limb that owns
        static_assert(SP::REW[t].n_idx <= 4, "wheel pairs");  // the joint, into wq[pair] (summed over the limbs below); the bodies' first-air flags: body loop
        static_for<0, SP::REW[t].n_idx>([&](auto pc) __attribute__((always_inline)) {
          constexpr int pi = decltype(pc)::value;
          constexpr int jid = SP::REW[t].idx_b[pi], kj = SP::JOINT_K[jid], jj = SP::JOINT_J[jid];
          wq[pi] = my_k == kj ? fabsf(this->qd[jj]) : 0.f;  // (counted by limb kj alone - a trunk joint sits in every lane; gsum below)
        });
      } else if constexpr (kd == REW_JOINT_MIRROR) {  // rewards.py:259-278: a pair is counted by the limb of its first joint; the partner's angle comes over by DPP
        float part = 0.f;
        static_for<0, SP::REW[t].n_idx>([&](auto pc) __attribute__((always_inline)) {
          constexpr int pi = decltype(pc)::value;
          constexpr int ja_id = SP::REW[t].idx_a[pi], jb_id = SP::REW[t].idx_b[pi];
          constexpr int ka = SP::JOINT_K[ja_id], ja = SP::JOINT_J[ja_id], kb = SP::JOINT_K[jb_id], jb = SP::JOINT_J[jb_id];
          float other = this->q[jb];
          if constexpr (jb < CL && (ka ^ kb) != 0) other = cx.template limb_xor<(ka ^ kb)>(this->q[jb]);  // (a trunk joint sits in every lane)
          const float d = this->q[ja] - other;
          part += my_k == ka ? d * d : 0.f;
        });
        a0[t] = part;
      }
    });
    // ---- body kinds: the body slots this lane owns
    {
      constexpr uint64_t REL = spec_rel_mask<SP>();
      ChainTP C = this->new_chain();
      // (trunk + limbs instance: the chain words in LDS already hold the kinematics of the final joint positions, step() stage 3)
      if constexpr (REL != 0ull && NW == 0) chain_kinematics<TP, SP>(L, q, C);
#pragma unroll
      for (int i = 0; i < Base::MAXOWN; ++i) {
        const int so = this->own[i];
        const int s = so < 0 ? 0 : so;
        const int bo = L.slot_body[s];
        const bool valid = so >= 0 && bo >= 0 && !(s == 0 && !L.owns_base_body);
        const int b = valid ? bo : 0;
        const F4 hs = hist_n.ld(s), ts = tim.ld(s);
        const float hm = fmaxf(hs.x, fmaxf(hs.y, hs.z));
        const float ca = ts.x, cc = ts.y, la = ts.z, lc = ts.w;
        V3 relp{0.f, 0.f, 0.f}, relv{0.f, 0.f, 0.f};
        if constexpr (REL != 0ull) {
          if (valid && in_body_mask<REL>(b)) body_rel(C, s, relp, relv);
        }
        static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
          constexpr int t = decltype(tc)::value;
          constexpr int kd = SP::REW[t].kind;
          constexpr uint64_t BM = SP::REW[t].body_mask;
          constexpr float p0 = SP::REW[t].p[0], p1 = SP::REW[t].p[1];
          if constexpr (kd == REW_FEET_GAIT) {  // the four feet's timers, each into its own accumulator: the sum over the env below is a broadcast
            static_for<0, 4>([&](auto fc) __attribute__((always_inline)) {
              constexpr int f = decltype(fc)::value;
              const bool is = valid && b == SP::REW[t].idx_a[f];
              gait_air[f] += is ? ca : 0.f;
              gait_con[f] += is ? cc : 0.f;
            });
          } else if constexpr (kd == REW_WHEEL_VEL_PENALTY) {  // first-air flag of the pair's wheel body, by the lane that owns the body's slot
            static_for<0, SP::REW[t].n_idx>([&](auto fc) __attribute__((always_inline)) {
              constexpr int f = decltype(fc)::value;
              ax[t][f] += valid && b == SP::REW[t].idx_a[f] && ca > 0.f && ca < fc_hi ? 1.f : 0.f;
            });
          } else if constexpr (kd == REW_FEET_DISTANCE_Y_EXP || kd == REW_FEET_DISTANCE_XY_EXP) {  // rewards.py:439-461, 464-505
            static_for<0, SP::REW[t].n_idx>([&](auto fc) __attribute__((always_inline)) {
              constexpr int f = decltype(fc)::value;
              const float ey = ((f & 1) ? -0.5f : 0.5f) * p1 - relp.y;
              const float ex = kd == REW_FEET_DISTANCE_XY_EXP ? (f < 2 ? 0.5f : -0.5f) * SP::REW[t].p[2] - relp.x : 0.f;
              a0[t] += valid && b == SP::REW[t].idx_a[f] ? ex * ex + ey * ey : 0.f;
            });
          } else if constexpr (BM != 0ull) {
            const bool on = valid && in_body_mask<BM>(b);
            if constexpr (kd == REW_HANDSTAND_FEET_AIR_TIME) a0[t] += on && cc > 0.f && cc < fc_hi ? la - p0 : 0.f;  // .../unitree_a1_handstand/env/rewards.py:40-47
            else if constexpr (kd == REW_HANDSTAND_FEET_ON_AIR) a0[t] += on && !(ca > 0.f && ca < fc_hi) ? 1.f : 0.f;  // .../env/rewards.py:31-37
            else if constexpr (kd == REW_HANDSTAND_FEET_HEIGHT_EXP) {                                               // .../env/rewards.py:18-28
              const float dz = pos.z + dot(Rwb.r2, relp) - p1;
              a0[t] += on ? dz * dz : 0.f;
            }
            else if constexpr (kd == REW_UNDESIRED_CONTACTS) a0[t] += on && hm > p0 ? 1.f : 0.f;                      // rewards.py:665-675
            else if constexpr (kd == REW_CONTACT_FORCES) a0[t] += on ? fmaxf(hm - p0, 0.f) : 0.f;                 // [UPSTREAM] contact_forces
            else if constexpr (kd == REW_FEET_CONTACT_WITHOUT_CMD || kd == REW_FEET_CONTACT) a0[t] += on && cc > 0.f && cc < fc_hi ? 1.f : 0.f;  // rewards.py:416-425, 399-413
            else if constexpr (kd == REW_FEET_AIR_TIME) a0[t] += on && cc > 0.f && cc < fc_hi ? la - p0 : 0.f;    // rewards.py:340-360
            else if constexpr (kd == REW_FEET_AIR_TIME_POSITIVE_BIPED) {                                          // rewards.py:363-383
              const bool inc = cc > 0.f;
              a0[t] += on && inc ? 1.f : 0.f;
              ax[t][0] = on ? fminf(ax[t][0], inc ? cc : ca) : ax[t][0];
            } else if constexpr (kd == REW_FEET_AIR_TIME_VARIANCE) {                                              // rewards.py:386-397
              const float xa = fminf(la, 0.5f), xc = fminf(lc, 0.5f), o = on ? 1.f : 0.f;
              a0[t] += o; ax[t][0] += o * xa; ax[t][1] += o * xa * xa; ax[t][2] += o * xc; ax[t][3] += o * xc * xc;
            } else if constexpr (kd == REW_FEET_STUMBLE) {                                                        // rewards.py:428-436
              const F4 fs = cf.ld(s);
              a0[t] += on && fsqrt(fs.x * fs.x + fs.y * fs.y) > 4.f * fabsf(fs.z) ? 1.f : 0.f;
            } else if constexpr (kd == REW_FEET_HEIGHT_BODY) {                                                    // rewards.py:527-554
              const float er = relp.z - p0;
              a0[t] += on ? er * er * ftanh(p1 * fsqrt(relv.x * relv.x + relv.y * relv.y)) : 0.f;
            } else if constexpr (kd == REW_FEET_SLIDE) {                                                          // rewards.py:557-587
              a0[t] += on && hm > 1.0f ? fsqrt(relv.x * relv.x + relv.y * relv.y) : 0.f;
            } else if constexpr (kd == REW_FEET_HEIGHT) {                                                         // rewards.py:507-524
              const V3 vw = lin_w + mul(Rwb, relv);
              const float er = pos.z + dot(Rwb.r2, relp) - p0;
              a0[t] += on ? er * er * ftanh(p1 * fsqrt(vw.x * vw.x + vw.y * vw.y)) : 0.f;
            }
          }
        });
      }
    }
    // ---- the env's sums (every lane ends with the same bits: gsum / esum pair identical values) and the terms' own arithmetic
    RL_PHASE(17, "rewards.terms");
    const float step_dt = ctx.uniform(T.step_dt);
    float v[NT];
    float total = 0.f;
    static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = decltype(tc)::value;
      constexpr int kd = SP::REW[t].kind;
      constexpr float p0 = SP::REW[t].p[0], p1 = SP::REW[t].p[1], p2 = SP::REW[t].p[2], p3 = SP::REW[t].p[3];
      float f = 0.f;
      if constexpr (is_scalar_reward_kind(kd)) {
        RewTab R{};
        R.kind = kd; R.p[0] = p0; R.p[1] = p1; R.p[2] = p2; R.p[3] = p3;
        f = scalar_term_value(R, E);
      } else if constexpr (kd == REW_JOINT_TORQUES_L2 || kd == REW_JOINT_ACC_L2 || kd == REW_JOINT_VEL_L2 || kd == REW_JOINT_POS_LIMITS || kd == REW_JOINT_POWER ||
                           kd == REW_JOINT_DEVIATION_L1 || kd == REW_ACTION_RATE_L2) {
        f = cx.gsum(a0[t]);
      } else if constexpr (kd == REW_STAND_STILL) {
        f = cx.gsum(a0[t]) * (cmd_norm < p0 ? 1.f : 0.f) * gate;
      } else if constexpr (kd == REW_JOINT_POS_PENALTY) {
        const float run = fsqrt(cx.gsum(a0[t]));
        f = ((cmd_norm > p2 || bv > p1) ? run : p0 * run) * gate;
      } else if constexpr (kd == REW_JOINT_MIRROR) {
        f = cx.gsum(a0[t]) * p0 * gate;
      } else if constexpr (kd == REW_WHEEL_VEL_PENALTY) {
        const bool running = cmd_norm > p1 || bv > p0;
        float part = 0.f;
        static_for<0, SP::REW[t].n_idx>([&](auto pc) __attribute__((always_inline)) {
          constexpr int pi = decltype(pc)::value;
          const float fa = cx.esum(ax[t][pi]), aq = cx.gsum(wq[pi]);  // (collectives: every lane)
          part += (running ? fa : 1.f) * aq;
        });
        f = part;
      } else if constexpr (kd == REW_BASE_HEIGHT_L2) {  // rewards.py:616-644; the 3 x 3 base ray caster (velocity_env_cfg.py:78-85): ray r by lane r % LPE
        float tgt = p0;
        if constexpr (p1 > 0.5f) {
          float hs = 0.f;
#pragma unroll
          for (int r0 = 0; r0 < 9; r0 += LPE) {
            const int r9 = r0 + my_li < 9 ? r0 + my_li : 8;
            const int iy = r9 / 3, ix = r9 - 3 * iy;
            const float lx = (float)(ix - 1) * 0.05f, ly = (float)(iy - 1) * 0.05f;
            float hz;
            V3 nn;
            terrain_sample(this->u, this->S.terrain, this->pos.x, this->pos.y, this->yaw_c * lx - this->yaw_s * ly, this->yaw_s * lx + this->yaw_c * ly, hz, nn);
            hs += r0 + my_li < 9 ? hz : 0.f;
          }
          tgt += cx.esum(hs) * (1.0f / 9.0f);
        }
        f = (this->pos.z - tgt) * (this->pos.z - tgt) * gate;
      } else if constexpr (kd == REW_FEET_DISTANCE_Y_EXP || kd == REW_FEET_DISTANCE_XY_EXP) {
        f = fexp(-cx.esum(a0[t]) * frcp(p0)) * gate;
      } else if constexpr (kd == REW_FEET_GAIT) {  // GaitReward, rewards.py:156-256
        float air[4], con[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { air[i] = cx.esum(gait_air[i]); con[i] = cx.esum(gait_con[i]); }
        const float inv_std = frcp(p0), me2 = p1 * p1;
        auto se = [&](float a, float b) { float d = a - b; return fminf(d * d, me2); };
        float sacc = se(air[0], air[1]) + se(con[0], con[1]) + se(air[2], air[3]) + se(con[2], con[3]);
        sacc += se(air[0], con[2]) + se(con[0], air[2]) + se(air[1], con[3]) + se(con[1], air[3]);
        sacc += se(air[0], con[3]) + se(con[0], air[3]) + se(air[2], con[1]) + se(con[2], air[1]);
        f = ((cmd_norm > p3 || bv > p2) ? fexp(-sacc * inv_std) : 0.f) * gate;
      } else {
        const float s0 = cx.esum(a0[t]);
        if constexpr (kd == REW_UNDESIRED_CONTACTS) f = s0 * gate;
        else if constexpr (kd == REW_CONTACT_FORCES) f = s0;
        else if constexpr (kd == REW_FEET_CONTACT_WITHOUT_CMD) f = s0 * (cmd_norm < 0.1f ? 1.f : 0.f) * gate;
        else if constexpr (kd == REW_FEET_CONTACT) f = (s0 != p0 ? 1.f : 0.f) * moving * gate;
        else if constexpr (kd == REW_FEET_AIR_TIME) f = s0 * moving * gate;
        else if constexpr (kd == REW_HANDSTAND_FEET_AIR_TIME) f = s0;
        else if constexpr (kd == REW_HANDSTAND_FEET_HEIGHT_EXP) f = fexp(-s0 * frcp(p0));
        else if constexpr (kd == REW_HANDSTAND_FEET_ON_AIR) f = s0 == 0.f ? 1.f : 0.f;
        else if constexpr (kd == REW_FEET_STUMBLE) f = (s0 > 0.f ? 1.f : 0.f) * gate;
        else if constexpr (kd == REW_FEET_HEIGHT_BODY || kd == REW_FEET_HEIGHT) f = s0 * moving * gate;
        else if constexpr (kd == REW_FEET_SLIDE) f = s0 * gate;
        else if constexpr (kd == REW_FEET_AIR_TIME_POSITIVE_BIPED) {
          const float mn = cx.emin(ax[t][0]);  // (a collective: every lane, whatever s0 says)
          f = (s0 == 1.f ? fminf(mn, p0) : 0.f) * moving * gate;
        }
        else if constexpr (kd == REW_FEET_AIR_TIME_VARIANCE) {
          const float s1 = cx.esum(ax[t][0]), s2 = cx.esum(ax[t][1]), s3 = cx.esum(ax[t][2]), s4 = cx.esum(ax[t][3]);
          const float inv_n = frcp(s0), inv_den = frcp(fmaxf(s0 - 1.f, 1.f));
          f = ((s2 - s1 * s1 * inv_n) + (s4 - s3 * s3 * inv_n)) * inv_den * gate;
        }
      }
      v[t] = f * SP::REW[t].weight * step_dt;  // RewardManager [UPSTREAM B2]
      total += v[t];
    });
    // ---- per-term outputs + episode sums: term t by lane t % LPE
    RL_PHASE(18, "rewards.writeback");
    static_for<0, NACC>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      float mine = 0.f;
      static_for<0, LPE>([&](auto cc2) __attribute__((always_inline)) {
        constexpr int c = decltype(cc2)::value, tt = i * LPE + c;
        if constexpr (tt < NT) mine = my_li == c ? v[tt] : mine;
      });
      const int t = my_li + LPE * i;
      if (t < NT) {
        S.rew_terms[(uint32_t)t * (uint32_t)Np + (uint32_t)e] = mine;
        const float ns = acc[i] + mine;
        S.ep_sums[(uint32_t)t * (uint32_t)Np + (uint32_t)e] = fold_done ? 0.f : ns;
        if constexpr (DEFER_LOG) this->log_sum[i] = ns;
        else if (fold_done && e < S.N) cx.atomic_add(this->log_slot() + LOG_EP_SUM0 + t, ns);
      }
    });
    return total;
  }

  // ---------------------------------------------------------------- observations [UPSTREAM B2 / B6]
  // pose of the height scanner: the root link on the quadrupeds, the torso on G1 (rides on trunk link scan_depth).  `scan_p`: x, y
  // as offsets from the root position (the terrain lookup adds the root in fp64, env_step.h terrain_fetch), z the world height
  // (`chain_fresh`: the chain words in LDS hold the kinematics of the current joint positions - no env of the wavefront was reset
  // since step() refreshed them)
  RL_FN void scanner_pose(float& cy, float& sy, V3& scan_p, bool chain_fresh) {
    if (NW == 0) {
      cy = yaw_c; sy = yaw_s; scan_p = {0.f, 0.f, pos.z};
      return;
    }
    M3 Rf;
    V3 pf;
    if (chain_fresh) {  // the chain words of the final joint positions are in place (step_front)
      ChainTP C = this->new_chain();
      trunk_frame<TP>(C, T.scan_depth, Rf, pf);
    } else {  // some env of the wavefront has a new posture: the trunk joints alone, in registers
      RL_PHASE(31, "obs.kinematics");
      trunk_frame_of_pose<TP>(L, q, this->u.trunk_restart, T.scan_depth, Rf, pf);
      RL_PHASE(20, "observations");
    }
    const M3 Rs = mul(Rwb, Rf);
    const float hn = frsqrt(fmaxf(Rs.r0.x * Rs.r0.x + Rs.r1.x * Rs.r1.x, 1e-30f));
    cy = Rs.r0.x * hn; sy = Rs.r1.x * hn;
    const V3 so = mul(Rwb, pf + mul(Rf, V3{T.scan_pos[0], T.scan_pos[1], T.scan_pos[2]}));
    scan_p = {so.x, so.y, pos.z + so.z};
  }

  // One observation group -> its LDS-staged row.  No dispatch on terms: the host expanded the term list into per-column
  // descriptors (ObsGroupTabT); a column is a gather from the env's feature vector F, [+ noise], clip, scale.  Groups that
  // corrupt stage the raw value first and finish in a second pass, 4 consecutive columns (= one Philox block: noise index =
  // noise_base + column, block = index >> 2 as in uniform01) per lane - a Philox4x32-10 call is ~900 cycles (v_mul_hi/lo_u32 are
  // quarter rate), so it must not be paid per column.
  // Height-scan rays of this lane: yaw-aligned grid, x fastest [UPSTREAM B6]; SCAN_RB rays per lane per trip so that all their
  // 8-byte loads overlap (187 rays = ONE trip of 12 x 16 lanes).  The gather costs ~5 us of the A1 Rough step at 4096 envs
  // (tools/ablate.sh): two cache lines per ray through the CU's vector L1, 6 k line requests per CU and step.
  static constexpr int SCAN_RB = LPE >= 32 ? 6 : 12;  // (32 lanes per env: 6 x 32 = 192 rays in the one trip)
  struct ScanPatches {
    TerrainPatch tp[SCAN_RB];
    bool single_trip;
  };
  // Which ray does slot s (= lane + LPE * i: what a lane fetches in its i-th load) stand for?  The reference's order - ray s, local x
  // fastest [UPSTREAM B6].  The heightfield is contiguous along WORLD y (hf[ix * ny + iy]), so that order puts the env's consecutive lanes next
  // to each other in memory when the robot looks along world y and a grid ROW apart (16 KB) when it looks along world x, where every lane of a
  // load touches cache lines of its own.  -DRL_SCAN_YAW_LANES (round 6) makes the slot -> ray map follow the yaw - heading
  // nearer to world x (|cos| > |sin|): local y fastest - with the set of rays, the value of each ray and its column unchanged.  Measured and
  // NOT kept: A1 Rough 4096 41.65 -> 43.25 us, Go2 43.87 -> 45.42 us (profiles/r06d_scan_lanes_ab.txt, one call; interpreter kernels of the same sources) - the specialised A1 kernel 35.45 -> 36.97 us with FETCH_SIZE unchanged (9.72 vs 9.74 MiB counted per launch): the lines are L2 hits either way, the two integer
  // divisions per slot and the scattered LDS row writes cost more than the shared lines save.
  RL_FN int scan_ray_of_slot(int s, int scan_n, float cy, float sy) const {
    s = s < scan_n ? s : scan_n - 1;
#ifndef RL_SCAN_YAW_LANES
    return s;
#else
    const int snx = ctx.uniform_i(T.scan_nx);
    const float inv_sny = ctx.uniform(1.0f / (float)T.scan_ny);
    const int ixt = (int)(((float)s + 0.5f) * inv_sny), iyt = s - ixt * ctx.uniform_i(T.scan_ny);  // s = ixt * ny + iyt
    return fabsf(cy) > fabsf(sy) ? iyt * snx + ixt : s;
#endif
  }
  RL_FN void scan_fetch_trip(int r0, int scan_n, float cy, float sy, V3 scan_p, ScanPatches& sp) const {
    const TerrainBase tb = terrain_base(this->u, pos.x, pos.y);  // the root in grid coordinates, once for all the lane's rays
    const int snx = ctx.uniform_i(T.scan_nx);
    const float inv_snx = ctx.uniform(1.0f / (float)T.scan_nx);
    const float res = T.scan_res, cx0 = 0.5f * (float)(T.scan_nx - 1), cy0 = 0.5f * (float)(T.scan_ny - 1);
#pragma unroll
    for (int i = 0; i < SCAN_RB; ++i) {
      const int r = scan_ray_of_slot(r0 + i * LPE, scan_n, cy, sy);
      int iy = (int)(((float)r + 0.5f) * inv_snx), ix = r - iy * snx;  // exact for r < 2^20
      float lx = ((float)ix - cx0) * res, ly = ((float)iy - cy0) * res;
      sp.tp[i] = terrain_fetch(this->u, S.terrain, tb, scan_p.x + cy * lx - sy * ly, scan_p.y + sy * lx + cy * ly);
    }
  }
  RL_FN void scan_fetch(float cy, float sy, V3 scan_p, ScanPatches& sp) const {
    const int scan_n = ctx.uniform_i(T.scan_nx * T.scan_ny);
    const bool wanted = ctx.uniform_i(T.obs[0].scan_n + T.obs[1].scan_n) > 0;
    sp.single_trip = wanted && scan_n <= SCAN_RB * LPE;
#ifdef RL_ABL_NO_SCAN
    sp.single_trip = false;
#endif
    if (sp.single_trip) scan_fetch_trip(li, scan_n, cy, sy, scan_p, sp);
  }

  // DIRECT: the row goes straight to its place in HBM (`stage` = the env's row of the output buffer) instead of an LDS staging row
  // that flush_obs() bursts out - for groups without noise in the one-lane-per-limb mapping, whose 16 envs x 235 critic columns
  // would otherwise cost 15 KB of LDS per wavefront, i.e. the fourth wavefront of a CU (direct_group below).  The 4 lanes of an env
  // write 16 consecutive bytes per instruction; the L2 merges the partial lines.
  template <bool DIRECT, class GT>
  RL_FN void write_group(const GT& G, const float* F, float* stage, uint32_t noise_base, float cy, float sy, V3 scan_p, const ScanPatches& sp) {
    constexpr int NITC = (TP::OBS_NC + LPE - 1) / LPE;
#ifdef RL_ABL_NO_SCAN
    const int n_cols = ctx.uniform_i(G.n_cols), scan_off = ctx.uniform_i(G.scan_off), scan_n = 0;
#else
    const int n_cols = ctx.uniform_i(G.n_cols), scan_off = ctx.uniform_i(G.scan_off), scan_n = ctx.uniform_i(G.scan_n);
#endif
    const int dim = ctx.uniform_i(G.dim);
    const bool corrupt = !DIRECT && ctx.uniform_i(G.corrupt) != 0;
    {  // non-scan columns: ordinal n = li, li + LPE, ...; all descriptor reads, then all feature reads, then the arithmetic
      ObsColTab d[NITC];
      float f[NITC];
#pragma unroll
      for (int i = 0; i < NITC; ++i) {
        const int n = li + LPE * i;
        d[i] = G.col[n < n_cols ? n : 0];
      }
#pragma unroll
      for (int i = 0; i < NITC; ++i) f[i] = F[d[i].src];
#pragma unroll
      for (int i = 0; i < NITC; ++i) {
        const int n = li + LPE * i;
        if (n < n_cols) stage[n < scan_off ? n : n + scan_n] = corrupt ? f[i] : clampf(f[i], d[i].clip_lo, d[i].clip_hi) * d[i].scale;
      }
    }
    if (scan_n > 0) write_scan(stage, scan_off, scan_n, corrupt, ctx.uniform(G.scan.scale), ctx.uniform(G.scan.clip_lo), ctx.uniform(G.scan.clip_hi), cy, sy, scan_p, sp);
    if (corrupt) noise_pass(G, stage, noise_base, dim, scan_off, scan_n);
  }
  // the scan's columns of a row: z_sensor - hit_z - offset per ray; the heightfield patches were fetched by scan_fetch()
  RL_FN void write_scan(float* stage, int scan_off, int scan_n, bool corrupt, float s_scale, float s_lo, float s_hi, float cy, float sy, V3 scan_p,
                        const ScanPatches& sp) {
    const float soff = T.scan_offset;
    if (sp.single_trip) {
#pragma unroll
      for (int i = 0; i < SCAN_RB; ++i) {
        const int sl = li + i * LPE, r = scan_ray_of_slot(sl, scan_n, cy, sy);
        const float v = scan_p.z - terrain_height(sp.tp[i]) - soff;
        if (sl < scan_n) stage[scan_off + r] = corrupt ? v : clampf(v, s_lo, s_hi) * s_scale;
      }
    } else {
      for (int r0 = li; r0 < scan_n; r0 += SCAN_RB * LPE) {
        ScanPatches one;
        scan_fetch_trip(r0, scan_n, cy, sy, scan_p, one);
#pragma unroll
        for (int i = 0; i < SCAN_RB; ++i) {
          const int sl = r0 + i * LPE, r = scan_ray_of_slot(sl, scan_n, cy, sy);
          const float v = scan_p.z - terrain_height(one.tp[i]) - soff;
          if (sl < scan_n) stage[scan_off + r] = corrupt ? v : clampf(v, s_lo, s_hi) * s_scale;
        }
      }
    }
  }
  // second pass of a group that corrupts: noise -> clip -> scale on the staged raw row, four consecutive columns (one Philox block) per lane
  template <class GT>
  RL_FN void noise_pass(const GT& G, float* stage, uint32_t noise_base, int dim, int scan_off, int scan_n) {
    ctx.group_sync();
    const int nblk = (dim + 3) >> 2;
    // (tried in round 5, one call: a lane's 3 - 4 blocks with their Philox rounds side by side - spelled out per round, results pinned so that
    // the compiler neither re-serialises the chains nor sinks them into the column branches - is SLOWER: A1 36.32 -> 36.88 us, Go2W 43.39 ->
    // 43.98; profiles/r05q_noise_interleave_ab.txt.  One block at a time it stays.)
#pragma unroll 1
    for (int b = li; b < nblk; b += LPE) {
      float un[4];
      uniform01x4(S.seed, (uint32_t)e, S.step_counter, STREAM_NOISE, (noise_base >> 2) + (uint32_t)b, un);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int col = 4 * b + c;
        if (col >= dim) continue;
        const bool in_scan = col >= scan_off && col < scan_off + scan_n;
        const int n = col < scan_off ? col : col - scan_n;
        const ObsColTab& dc = in_scan ? G.scan : G.col[in_scan ? 0 : n];
        stage[col] = clampf(stage[col] + fmaf(dc.noise_rng, un[c], dc.noise_lo), dc.clip_lo, dc.clip_hi) * dc.scale;  // (explicit fma: rl_math.h lerp_draw)
      }
    }
  }

  // the height-scan loads of the final pose: issued as early as the pose is final (see step_back) and consumed by the group(s) that carry
  // the scan, after everything else of the observation stage
  struct ScanAhead {
    ScanPatches sp;
    float cy, sy;
    V3 scan_p;
  };
  RL_FN void scan_ahead(bool chain_fresh, ScanAhead& A) {
    derive();
    scanner_pose(A.cy, A.sy, A.scan_p, chain_fresh);
    scan_fetch(A.cy, A.sy, A.scan_p, A.sp);
  }
  RL_FN void observations(bool chain_fresh) {
    ScanAhead A;
    scan_ahead(chain_fresh, A);
    observations(A);
  }
  // The observation stage on a Spec (env_spec.h), lane mappings that stage both rows in LDS: every value goes from its owner's registers
  // straight to its column of the row - lane 0 of the env the base terms, the first sub-lane of a limb its joints, the scan's rays as
  // above - with the term's offset, clip and scale constant expressions: no feature vector in LDS, no per-column descriptors, one fence
  // less.  The noise pass of a group that corrupts is the interpreter's (a Philox block per four columns; its per-column constants still
  // come from the table image).  Same values, same bits.  (What this stage costs is the scan's gather and the noise, DESIGN.md section 3:
  // this saves two dependent LDS round trips and ~60 vector instructions of it.)
  RL_FN void observations_spec(const ScanAhead& A) {
    derive();
    Ctx& cx = ctx;  // (plain locals for the nested generic lambdas: see compute_rewards_spec)
    const int my_li = li, my_sub = sub;
    const uint32_t wheel = T.wheel_joint_mask;
    static_for<0, 2>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      float* stage = cx.obs_stage(g);
      // (a group the Spec's task corrupts may run uncorrupted: play.py:134 switches `enable_corruption` off and nothing else a Spec holds,
      // so the play variant of a task matches its Spec - env_spec.h spec_matches - and the flag is read from the table image)
      const bool corrupt = SP::OBS_CORRUPT[g] != 0 && cx.uniform_i(this->T.obs[g].corrupt) != 0;
      static_for<0, SP::N_OBS[g]>([&](auto tc) __attribute__((always_inline)) {
        constexpr int tm = decltype(tc)::value;
        constexpr ObsSpec O = SP::OBS[g][tm];
        auto fin = [&](float v) __attribute__((always_inline)) { return corrupt ? v : clampf(v, O.clip_lo, O.clip_hi) * O.scale; };
        if constexpr (O.kind == OBS_BASE_LIN_VEL || O.kind == OBS_BASE_ANG_VEL || O.kind == OBS_PROJECTED_GRAVITY || O.kind == OBS_VELOCITY_COMMANDS) {
          const V3 v = O.kind == OBS_BASE_LIN_VEL ? this->lin_b : (O.kind == OBS_BASE_ANG_VEL ? this->ang_b : (O.kind == OBS_PROJECTED_GRAVITY ? this->grav_b : this->cmd));
          if (my_li == 0) {
            stage[O.offset + 0] = fin(v.x);
            stage[O.offset + 1] = fin(v.y);
            stage[O.offset + 2] = fin(v.z);
          }
        } else if constexpr (O.kind == OBS_HEIGHT_SCAN) {
#ifndef RL_ABL_NO_SCAN
          this->write_scan(stage, O.offset, cx.uniform_i(this->T.scan_nx * this->T.scan_ny), corrupt, O.scale, O.clip_lo, O.clip_hi, A.cy, A.sy, A.scan_p, A.sp);
#endif
        } else {
          if (my_sub == 0) {
            static_for<0, JX>([&](auto jc) __attribute__((always_inline)) {
              constexpr int j = decltype(jc)::value;
              const int jid = (!TP::PAD || this->L.joint_own[j]) ? this->L.joint_id[j] : -1;
              if (jid >= 0) {
                const float qr = this->q[j] - this->L.q0[j];
                float v;
                if constexpr (O.kind == OBS_JOINT_POS_REL) v = qr;
                else if constexpr (O.kind == OBS_JOINT_VEL_REL) v = this->qd[j] - this->L.qd0[j];
                else if constexpr (O.kind == OBS_LAST_ACTION) v = this->act[j];
                else v = ((wheel >> (jid & 31)) & 1u) ? 0.f : qr;  // OBS_JOINT_POS_REL_NO_WHEEL (observations.py:17-27)
                stage[O.offset + jid] = fin(v);
              }
            });
          }
        }
      });
      if (corrupt) {
        const auto& G = this->T.obs[g];
        this->noise_pass(G, stage, g == 0 ? 0u : 1024u, cx.uniform_i(G.dim), cx.uniform_i(G.scan_off), cx.uniform_i(G.scan_n));
      }
      if constexpr (g == 0) RL_PHASE(21, "obs.policy_done");
    });
    RL_PHASE(22, "obs.flush");
    ctx.flush_obs(S.obs_policy, T.policy_dim, 0);
    ctx.flush_obs(S.obs_critic, T.critic_dim, 1);
  }
  RL_FN void observations(const ScanAhead& A) {
#ifndef RL_SPEC_OBS_OFF  // (A/B switch)
    // quadruped instances with several sub-lanes per limb.  (One lane per limb: a group without noise goes straight to HBM - the interpreter's
    // write_group<DIRECT>.  Trunk + limbs instances: the owner-writes are 10 joints x 3 terms x 2 groups by the limb's first sub-lane - more
    // LDS instructions per wavefront than the column tables spread over 32 lanes: G1 102.3 -> 104.3 us, profiles/r05d_spec_obs_ab.txt.)
    if constexpr (SP::ON && SUB > 1 && NW == 0) {
      observations_spec(A);
      return;
    }
#endif
    derive();
    // the height-scan loads went out first and are consumed by the group(s) that carry the scan, after the feature vector and the
    // non-scan columns.  (Issuing them before the reward stage was tried: the 72 patch registers do not survive it - the compiler
    // parks them in AGPRs, which needs the data, i.e. waits for the loads on the spot: 52.6 us either way.)
    const ScanPatches& sp = A.sp;
    const float cy = A.cy, sy = A.sy;
    const V3 scan_p = A.scan_p;
    // the env's feature vector -> LDS (env_tables.h FEAT_*): lane 0 the base block, the first sub-lane of a limb its joints
    float* F = ctx.feat_stage();
    const int D = ctx.uniform_i(T.D);
    if (li == 0) {
      F[FEAT_LIN + 0] = lin_b.x; F[FEAT_LIN + 1] = lin_b.y; F[FEAT_LIN + 2] = lin_b.z;
      F[FEAT_ANG + 0] = ang_b.x; F[FEAT_ANG + 1] = ang_b.y; F[FEAT_ANG + 2] = ang_b.z;
      F[FEAT_GRAV + 0] = grav_b.x; F[FEAT_GRAV + 1] = grav_b.y; F[FEAT_GRAV + 2] = grav_b.z;
      F[FEAT_CMD + 0] = cmd.x; F[FEAT_CMD + 1] = cmd.y; F[FEAT_CMD + 2] = cmd.z;
    }
    if (sub == 0) {
      const uint32_t wheel = T.wheel_joint_mask;
#pragma unroll
      for (int j = 0; j < JX; ++j) {
        const int jid = (!TP::PAD || L.joint_own[j]) ? L.joint_id[j] : -1;
        if (jid < 0) continue;  // padding / trunk joints accounted for by limb 0
        const float qr = q[j] - L.q0[j];
        F[FEAT_JOINT + jid] = qr;
        F[FEAT_JOINT + D + jid] = qd[j] - L.qd0[j];
        F[FEAT_JOINT + 2 * D + jid] = act[j];
        F[FEAT_JOINT + 3 * D + jid] = ((wheel >> (jid & 31)) & 1u) ? 0.f : qr;
      }
    }
    ctx.group_sync();
    const bool d0 = direct_group(T, 0), d1 = direct_group(T, 1);
    if (SUB == 1 && d0) write_group<true>(T.obs[0], F, S.obs_policy + (size_t)e * (size_t)T.policy_dim, 0u, cy, sy, scan_p, sp);
    else write_group<false>(T.obs[0], F, ctx.obs_stage(0), 0u, cy, sy, scan_p, sp);
    RL_PHASE(21, "obs.policy_done");
    if (SUB == 1 && d1) write_group<true>(T.obs[1], F, S.obs_critic + (size_t)e * (size_t)T.critic_dim, 1024u, cy, sy, scan_p, sp);
    else write_group<false>(T.obs[1], F, ctx.obs_stage(1), 1024u, cy, sy, scan_p, sp);
    RL_PHASE(22, "obs.flush");
    if (!(SUB == 1 && d0)) ctx.flush_obs(S.obs_policy, T.policy_dim, 0);
    if (!(SUB == 1 && d1)) ctx.flush_obs(S.obs_critic, T.critic_dim, 1);
  }

  // episode log of THIS step: slot step_counter % LOG_RING.  Every step starts from a slot the previous step zeroed, so a
  // slot is what the reference rebuilds as extras["log"] on every call - no snapshot + memset between steps on the host
  // (the wavefront's partial row of the slot: env_tables.h LOG_PARTS)
  RL_FN float* log_slot() const {
    return S.log + ((size_t)(S.step_counter & (uint32_t)(LOG_RING - 1)) * LOG_PARTS + (size_t)(ctx.tile() & (LOG_PARTS - 1))) * LOG_SIZE;
  }

  // command_levels_lin_vel / _ang_vel (VEL/mdp/curriculums.py:21-94) decide on the mean episode sum of a driving reward term over
  // the envs that the deciding step (count % max_episode_length == 0) resets - a reduction over the whole launch that the reference
  // takes FIRST inside _reset_idx, so the commands those very resets draw, and the heading clip of that step, already see the
  // widened range.  Envs with these curricula therefore step in TWO launches (rl_env_host.h step()): step_head() - everything up
  // to the rewards, this collection, state written back -, the one-thread decision (apply_cmd_levels), then step_tail() from the
  // re-loaded state: resets, commands, push, observations.  Tasks without the curricula (every shipped cfg deletes the terms,
  // unitree_a1/rough_env_cfg.py:158-159) run step(): one launch.
  RL_FN void collect_cmd_levels(bool done) {
    if (!(T.cur_lin || T.cur_ang) || !done || e >= S.N || S.step_counter % (uint32_t)T.max_episode_length != 0u) return;
    for (int t = li; t < T.n_rewards; t += LPE) {
      const float v = S.ep_sums[(size_t)t * Np + e];
      if (T.cur_lin && t == T.cur_lin_term) { ctx.atomic_add(S.cmd_levels + CL_SUM_LIN, v); ctx.atomic_add(S.cmd_levels + CL_CNT_LIN, 1.f); }
      if (T.cur_ang && t == T.cur_ang_term) { ctx.atomic_add(S.cmd_levels + CL_SUM_ANG, v); ctx.atomic_add(S.cmd_levels + CL_CNT_ANG, 1.f); }
    }
  }
  // inspection views as a reader sees them after step(): zeroed by the reset of a done env (`live` = 0)
  RL_FN void write_dbg_views(float live) {
    if (S.dbg_torque == nullptr) return;
    if (sub == 0)
#pragma unroll
      for (int j = 0; j < JX; ++j) {
        if (TP::PAD && !L.joint_own[j]) continue;
        S.dbg_torque[(size_t)e * T.D + L.joint_id[j]] = live * tau_app[j];
        S.dbg_acc[(size_t)e * T.D + L.joint_id[j]] = live * qacc[j];
      }
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
      int b = L.slot_body[s];
      if (b >= 0 && (s != 0 || L.owns_base_body) && this->owns_slot(s)) {
        float* o = S.dbg_cforce + ((size_t)e * T.n_bodies + b) * 3;
        const F4 fs = cf.ld(s);
        o[0] = live * fs.x; o[1] = live * fs.y; o[2] = live * fs.z;
      }
    }
  }
  RL_FN bool out_of_bounds() const {  // terrain_out_of_bounds (velocity_env_cfg.py:655-659)
    if (!T.term_oob || T.is_plane) return false;
    const float mw = (float)T.num_rows * T.tile_size + 2.f * T.border, mh = (float)T.num_cols * T.tile_size + 2.f * T.border;
    return fabsf(pos.x) > 0.5f * mw - T.oob_buffer || fabsf(pos.y) > 0.5f * mh - T.oob_buffer;
  }

  // ---------------------------------------------------------------- step()
  RL_FN void step() { step_front<false>(); }
  RL_FN void step_head() { step_front<true>(); }
  // second launch of a split step: the state step_head() wrote back, its termination flags, then stages 6 - 9
  RL_FN void step_tail() {
    this->load();
    load_task();
    derive();
    const bool terminated = S.terminated[e] != 0, time_out = S.time_out[e] != 0;
    const bool t_oob = out_of_bounds();
    const bool t_timeout = T.term_time_out && ep_len >= (long long)T.max_episode_length;  // (step_head counted this step already)
    step_back<true>(terminated, time_out, t_timeout, t_oob, terminated);
  }

  template <bool HEAD>
  RL_FN void step_front() {
    // episode-log ring upkeep, by the first LOG_PARTS wavefronts of the launch: wavefront p keeps partial row p, lane l its word l
    // (a) the previous step's slot is final now: if that step reset nobody, it inherits its predecessor, so that every
    //     slot reads as "the log of the most recent step that reset an env" - what a caller of the reference sees, which
    //     rebuilds extras["log"] only inside _reset_idx [UPSTREAM B1];  (b) clear the next step's slot.
    // (whether the step reset anybody is the sum of its rows' LOG_FRESH words, which no copy touches: every wavefront reads the same
    // answer whatever the others have written by then.)  Its reads go out HERE, in one batch with the state's; the writes follow the state loads.
    constexpr size_t LOG_SLOT_WORDS = (size_t)LOG_PARTS * LOG_SIZE;
    const int log_tl = ctx.tile(), log_wl = ctx.env_in_tile() * LPE + li;  // wavefront of the launch, lane of the wavefront
    float* log_pv = S.log + (size_t)((S.step_counter - 1u) & (uint32_t)(LOG_RING - 1)) * LOG_SLOT_WORDS;
    const float* log_pp = S.log + (size_t)((S.step_counter - 2u) & (uint32_t)(LOG_RING - 1)) * LOG_SLOT_WORDS;
    float log_fresh = 0.f, log_word = 0.f;
    if (log_tl < LOG_PARTS) {
      for (int p = 0; p < LOG_PARTS; ++p) log_fresh += log_pv[p * LOG_SIZE + LOG_FRESH];
      log_word = log_pp[log_tl * LOG_SIZE + log_wl];
    }
    // the actions: their loads go out with the state's, unconditionally (a padding env reads env 0's row, a padding joint column 0; the
    // values are selected below).  Under `e < N && joint_id >= 0 ? load : 0` each load sat in a branch of its own behind the LDS read of
    // its joint id, and the clamp that follows waited for it: ten HBM round trips one after the other on the trunk + limbs instances
    // (G1: 5.5 k ticks of the step, profiles/r05k_phase_clock_g1_reset_every64.txt)
    float a_in[JX];
    {
      const float* arow = S.action_in + (size_t)(e < S.N ? e : 0) * T.D;
#pragma unroll
      for (int j = 0; j < JX; ++j) {
        const int jid = L.joint_id[j];
        a_in[j] = arow[TP::PAD && jid < 0 ? 0 : jid];
      }
    }
    RL_PHASE_START();
    RL_PHASE(0, "load");
    this->load();
    load_task();  // same batch of HBM loads as the state: one round trip instead of a second one after the substeps
    rl_pin(a_in);
    if (log_tl < LOG_PARTS) {
      float* nx = S.log + (size_t)((S.step_counter + 1u) & (uint32_t)(LOG_RING - 1)) * LOG_SLOT_WORDS;
      const bool inherit = log_fresh == 0.f;
      if (inherit && log_wl != LOG_FRESH) log_pv[log_tl * LOG_SIZE + log_wl] = log_word;
      nx[log_tl * LOG_SIZE + log_wl] = 0.f;
      for (int p = log_tl + S.Npad / (64 / LPE); p < LOG_PARTS; p += S.Npad / (64 / LPE)) {  // (a launch of fewer than LOG_PARTS wavefronts)
        if (inherit && log_wl != LOG_FRESH) log_pv[p * LOG_SIZE + log_wl] = log_pp[p * LOG_SIZE + log_wl];
        nx[p * LOG_SIZE + log_wl] = 0.f;
      }
    }
    RL_PHASE(1, "action");
    // 1 ActionManager.process_action [UPSTREAM B2]; JointPosition/VelocityAction (velocity_env_cfg.py:124-126)
    float q_tgt[JX], qd_tgt[JX];
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      prev_act[j] = act[j];
      const float a = (e < S.N && (!TP::PAD || L.joint_id[j] >= 0)) ? a_in[j] : 0.f;
      act[j] = a;
      float pr = clampf(a * L.a_scale[j] + L.a_off[j], L.a_lo[j], L.a_hi[j]);
      q_tgt[j] = L.action_is_vel[j] ? 0.f : pr;
      qd_tgt[j] = L.action_is_vel[j] ? pr : 0.f;
    }
    // 2 decimation loop: actuators -> physics -> contact sensor
#ifdef RL_ABL_SUBSTEPS
    if constexpr (Base::ABA) {
      this->substeps_aba(q_tgt, qd_tgt, RL_ABL_SUBSTEPS);
    } else
#endif
    if constexpr (Base::ABA) {
      this->substeps_aba(q_tgt, qd_tgt, T.decimation);
    } else {
      this->self_load();
      for (int s = 0; s < T.decimation; ++s) this->substep_aba_trunk(q_tgt, qd_tgt);
    }
    RL_PHASE(15, "terminations");
    // 3 counters
    ep_len += 1;
    derive();
    if constexpr (NW > 0) {  // kinematics of the final joint positions, once, for the reward stage (feet) and the scanner pose
      ChainTP Cf = this->new_chain();
      this->kinematics(Cf);
      ctx.group_sync();
    }

    // 4 terminations (velocity_env_cfg.py:648-664)
    bool t_timeout = T.term_time_out && ep_len >= (long long)T.max_episode_length;
    const bool t_oob = out_of_bounds();
    bool t_illegal = false;
    if (T.term_illegal) {
      // over the slots this lane keeps (own[]: all of them when a lane is a whole limb), every read issued whether or not the slot counts:
      // `bit && owns && hist > thr` over all NBS slots was up to three dependent LDS round trips per slot, nine slots in a row
      const uint64_t ill_mask = T.illegal_body_mask;
      const float ill_thr = T.illegal_threshold;
      float c = 0.f;
#pragma unroll
      for (int i = 0; i < Base::MAXOWN; ++i) {
        const int so = this->own[i], s = so < 0 ? 0 : so;
        const float h = hist_max(s);
        const bool bit = body_bit(ill_mask, s);
        c += (so >= 0 && bit && h > ill_thr) ? 1.f : 0.f;
      }
      t_illegal = ctx.esum(c) > 0.f;
    }
    bool terminated = t_illegal, time_out = t_timeout || t_oob;
    // 5 rewards
    RL_PHASE(16, "rewards");
#ifdef RL_ABL_NO_REWARDS  // analysis builds: what the kernel costs without this stage (tools/ablate.sh)
    float rew = 0.f;
    const bool fold_done = false;
#else
    float rew;
    const bool fold_done = !HEAD && (terminated || time_out);  // the episode sums' log + zero ride in the write-back (compute_rewards)
    if constexpr (SP::ON) rew = compute_rewards_spec(terminated, fold_done);
    else rew = compute_rewards(terminated, fold_done);
#endif
    RL_PHASE(19, "resets+commands+push");
    if (li == 0) {
      S.reward[e] = rew;
      S.terminated[e] = terminated ? 1 : 0;
      S.time_out[e] = time_out ? 1 : 0;
      if (S.ro_rewards != nullptr && e < S.N) {  // rl_env_step_record: the transition's second half goes straight into the rollout storage
        if (S.ro_values != nullptr) {
          S.ro_rewards[e] = time_out ? fmaf(S.ro_gamma, S.ro_values[e], rew) : rew;  // bootstrapping on time outs (rsl_rl PPO.process_env_step); one explicit fma, as record_kernel / gae_kernel
          S.ro_dones[e] = (terminated || time_out) ? 1 : 0;
        } else {  // deferred bootstrap (values_dev == NULL: the critic of this step may still be running on another stream): the raw
          S.ro_rewards[e] = rew;  // reward, and the time-out flag in bit 1 of the done byte - rl_rollout_compute_returns applies gamma * V there
          S.ro_dones[e] = ((terminated || time_out) ? 1 : 0) | (time_out ? 2 : 0);
        }
      }
    }
    if constexpr (HEAD) {  // first launch of a split step: the decision's inputs, the views, the state; step_tail() goes on from here
      collect_cmd_levels(terminated || time_out);
      write_dbg_views((terminated || time_out) ? 0.f : 1.f);
      this->store();
      store_task();
    } else {
      step_back<false>(terminated, time_out, t_timeout, t_oob, t_illegal, fold_done);
    }
  }

  template <bool TAIL>
  RL_FN void step_back(const bool terminated, const bool time_out, const bool t_timeout, const bool t_oob, const bool t_illegal, const bool sums_folded = false) {
    // 6 reset done envs
    if (terminated || time_out) {
      if (!(DEFER_LOG && sums_folded) && li == 0 && e < S.N) {
        if (t_timeout) ctx.atomic_add(log_slot() + LOG_TERM_TIMEOUT, 1.f);
        if (t_oob) ctx.atomic_add(log_slot() + LOG_TERM_OOB, 1.f);
        if (t_illegal) ctx.atomic_add(log_slot() + LOG_TERM_ILLEGAL, 1.f);
      }
      reset_env(true, sums_folded);
      derive();
      RL_PHASE(19, "resets+commands+push");
    }
    if constexpr (!TAIL) write_dbg_views((terminated || time_out) ? 0.f : 1.f);  // (a split step wrote them in its first launch)
#ifdef RL_SCAN_EARLY  // (A/B switch) the scan's loads go out HERE: the pose is final (commands and the push below touch velocities only), and the
    // command update, the push and the write-back of the state then run under their latency
    ScanAhead scanA;
    scan_ahead(!TAIL && !ctx.any(terminated || time_out), scanA);
#endif
    // 7 CommandManager.compute [UPSTREAM B7]
    RL_PHASE(29, "commands");
    {
      const float inv_max_step = T.step_dt * frcp(T.cmd_resample[1]);
      float ex = cmd.x - lin_b.x, ey = cmd.y - lin_b.y;
      metric_xy += fsqrt(ex * ex + ey * ey) * inv_max_step;
      metric_yaw += fabsf(cmd.z - ang_b.z) * inv_max_step;
      cmd_time_left -= T.step_dt;
      if (cmd_time_left <= 0.f) {
        cmd_time_left = U(STREAM_COMMAND, 6, T.cmd_resample[0], T.cmd_resample[1]);
        resample_command(STREAM_COMMAND, 0);
      }
      if (T.cmd_heading && is_heading) {  // the only consumer of the heading angle itself (everything else uses its cos / sin)
        const float heading_w = atan2f(yaw_s, yaw_c);
        cmd.z = clampf(T.cmd_heading_stiffness * wrap_to_pi(heading_target - heading_w), T.cur_ang ? S.cmd_levels[CL_ANG_Z] : T.cmd_range[2][0],
                       T.cur_ang ? S.cmd_levels[CL_ANG_Z + 1] : T.cmd_range[2][1]);  // the LIVE range: the curriculum edits cfg.ranges in place
      }
      if (is_standing) cmd = {0.f, 0.f, 0.f};
      // the "pits" branch of commands.py:61-85 never fires: ROUGH_TERRAINS_CFG has no sub-terrain of that name (utils.py:27-28)
    }
    // 8 interval event: push_by_setting_velocity (velocity_env_cfg.py:366-371) [UPSTREAM B2/B8]
    RL_PHASE(30, "push");
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
    // The state is final here: write it back BEFORE the observation stage, so that the stores drain behind that stage's
    // arithmetic instead of behind the end of the kernel (a wavefront has 63 memory operations in flight at most; the trunk +
    // limbs instance issues ~100 stores, i.e. it used to sit out a full HBM write round trip with nothing else to do: 11 us of
    // G1's 150)
    RL_PHASE(23, "store");
#ifndef RL_ABL_NO_STORE  // analysis builds: what the write-back of the state costs (wrong results)
    this->store();
    store_task();
#endif
    // 9 observations
    RL_PHASE(20, "observations");
#ifndef RL_ABL_NO_OBS
#ifdef RL_SCAN_EARLY
    observations(scanA);
#else
    observations(!TAIL && !ctx.any(terminated || time_out));  // (a tail launch starts without the chain words of the trunk + limbs instance)
#endif
#endif
    // 10 the episode log of a done env (see DEFER_LOG)
#ifndef RL_ABL_NO_LOG  // analysis builds: what the log's same-address atomic adds cost a launch in which many envs reset (wrong logs)
    if constexpr (DEFER_LOG) {
      if (sums_folded && e < S.N) {
        float* slot = log_slot();
        int n_rew;
        if constexpr (SP::ON) n_rew = SP::N_REW;
        else n_rew = T.n_rewards;
#pragma unroll
        for (int i = 0; i < NACC_M; ++i) {
          const int t = li + LPE * i;
          if (t < n_rew) ctx.atomic_add(slot + LOG_EP_SUM0 + t, log_sum[i]);
        }
        if (li == 0) {
          ctx.atomic_add(slot + LOG_RESET_COUNT, 1.0f);
          ctx.atomic_add(slot + LOG_FRESH, 1.0f);
          ctx.atomic_add(slot + LOG_METRIC_XY, log_mxy);
          ctx.atomic_add(slot + LOG_METRIC_YAW, log_myaw);
          if (t_timeout) ctx.atomic_add(slot + LOG_TERM_TIMEOUT, 1.f);
          if (t_oob) ctx.atomic_add(slot + LOG_TERM_OOB, 1.f);
          if (t_illegal) ctx.atomic_add(slot + LOG_TERM_ILLEGAL, 1.f);
        }
      }
    }
#endif
    RL_PHASE(24, "end");
  }

  // ---------------------------------------------------------------- reset() entry: reset masked envs, recompute obs
  RL_FN void reset_entry() {
    if (S.mode == KMODE_STEP_TAIL) {  // (rides in the reset kernels: neither is on the hot path, and the step kernels stay as they are)
      step_tail();
      return;
    }
    this->load();
    load_task();
#pragma unroll
    for (int j = 0; j < JX; ++j) prev_act[j] = act[j];
    if (S.reset_mask == nullptr || S.reset_mask[e]) reset_env(false);
    observations(false);
    this->store();
    store_task();
  }
};

}  // namespace rl
