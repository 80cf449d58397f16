// rl_env_host.h - host side of the C-ABI (include/rl_env.h): descriptor -> device tables, HBM
// allocation in the SoA layout of env_tables.h, "startup" events, launches.  Backend-agnostic: the
// HIP build (rl_env.hip) and the CPU lane-emulator build (tests/emu) instantiate it with their own
// alloc/copy/launch primitives, so both expose the identical C-ABI.
#pragma once
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/rl_env.h"
#include "env_aos.h"
#include "env_spec.h"
#include "env_tables.h"

namespace rl {

static_assert((int)TASK_NF == (int)RL_TASK_STATE_NF && (int)TS_EXT_F == (int)RL_TS_EXT_FORCE && (int)TS_PUSH == (int)RL_TS_PUSH_TIME_LEFT, "RL_BUF_TASK_STATE row layout");

inline std::string& last_error() {
  static thread_local std::string e;
  return e;
}
inline int fail(const std::string& msg) {
  last_error() = msg;
  return -1;
}

inline int obs_term_dim(const rl_env_desc& d, int kind) {
  switch (kind) {
    case RL_OBS_JOINT_POS_REL: case RL_OBS_JOINT_VEL_REL: case RL_OBS_LAST_ACTION: case RL_OBS_JOINT_POS_REL_NO_WHEEL: return d.model.num_dof;
    case RL_OBS_HEIGHT_SCAN: return d.task.scan_nx * d.task.scan_ny;
    default: return 3;
  }
}

// ------------------------------------------------------------------------------------------------
// descriptor -> Tables.  Requires the trunk + 4 limb chains topology the lane program is written for.
// ------------------------------------------------------------------------------------------------
inline void quat_to_rows(const float q[4], float R[9]) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  double n = sqrt(w * w + x * x + y * y + z * z);
  if (n < 1e-12) { w = 1; x = y = z = 0; n = 1; }
  w /= n; x /= n; y /= n; z /= n;
  R[0] = (float)(1 - 2 * (y * y + z * z)); R[1] = (float)(2 * (x * y - w * z)); R[2] = (float)(2 * (x * z + w * y));
  R[3] = (float)(2 * (x * y + w * z)); R[4] = (float)(1 - 2 * (x * x + z * z)); R[5] = (float)(2 * (y * z - w * x));
  R[6] = (float)(2 * (x * z - w * y)); R[7] = (float)(2 * (y * z + w * x)); R[8] = (float)(1 - 2 * (x * x + y * y));
}

// shape of the lane-program instance that simulates this model (env_tables.h Topo<>)
inline int topo_shape(const rl_model_desc& m, int& CL, int& NW, int& SPL, int& NBS, int& rotpad) {
  if (m.num_chains != NLANE || m.chain_len < 1 || m.chain_len > MAX_CL || m.num_trunk < 0 || m.num_trunk > MAX_NW)
    return fail("lane program needs a trunk of <= 6 serial joints carrying 4 limb chains of <= 7 joints (got " + std::to_string(m.num_chains) + " chains x " +
                std::to_string(m.chain_len) + ", trunk " + std::to_string(m.num_trunk) + ")");
  bool equal = true;
  for (int k = 1; k < NLANE; ++k) equal = equal && m.chain_nj[k] == m.chain_nj[0];
  // joint frames rotated against the parent link (URDF joint rpy; DDT Tita): only the trunk + limbs instance carries the rotations
  bool rotated = false;
  for (int l = 1; l < m.num_links; ++l) {
    const float* q = m.link_quat[l];
    const float n = fabsf(q[0]) + fabsf(q[1]) + fabsf(q[2]) + fabsf(q[3]);
    rotated = rotated || (n > 0.f && (fabsf(q[1]) > 1e-6f || fabsf(q[2]) > 1e-6f || fabsf(q[3]) > 1e-6f));  // all-zero: descriptor without rotations
  }
  rotpad = 0;
  if (m.num_trunk == 0 && m.chain_len <= 4 && equal && m.chain_len >= 3 && !rotated) { CL = m.chain_len; NW = 0; SPL = 3; NBS = 6; }
  else if (m.num_trunk == 0 && m.chain_len <= 4 && !(std::getenv("RL_ENV_ROTPAD") && atoi(std::getenv("RL_ENV_ROTPAD")) == 0)) {
    // quadruped-shaped, but with rotated joint frames or limbs of unequal length (DDT Tita: two wheeled 4-joint legs, two empty limbs): the
    // 4-joint quadruped instance with rot0 and padding joints, Topo<4,0,3,6,0,1>.  (RL_ENV_ROTPAD=0: the trunk + limbs instance, as until round 4)
    CL = 4; NW = 0; SPL = 3; NBS = 6; rotpad = 1;
  } else { CL = 7; NW = m.num_trunk > 3 ? 6 : 3; SPL = 4; NBS = 9; }
  // a shorter trunk (ATOM01: one waist joint) or none runs on the NW = 3 instance with inert padding trunk joints; a longer one
  // (GR1: waist + head) on the NW = 6 instance
  return 0;
}

inline int build_tables(const rl_env_desc& d, Tables& T, std::vector<int>& body_lane, std::vector<int>& body_slot, std::vector<int>& link_lane_out,
                        std::vector<int>& link_pos_out, bool merge = false) {
  memset(&T, 0, sizeof(T));
  const rl_model_desc& m = d.model;
  int CL, NW, SPL, NBS, rotpad;
  if (topo_shape(m, CL, NW, SPL, NBS, rotpad)) return -1;
  const int NGRP = CL + 1;
  if (merge && !(NW == 0 && CL == 4 && !rotpad)) return fail("merged base share: 4-joint quadruped limbs (same length, unrotated joint frames) only");
  T.CL = CL; T.NW = NW; T.SPL = SPL; T.NBS = NBS; T.nw_used = m.num_trunk; T.merged = merge ? 1 : 0; T.rotpad = rotpad;
  T.D = m.num_dof;
  T.n_bodies = m.num_bodies;
  body_lane.assign(m.num_bodies, -1);
  body_slot.assign(m.num_bodies, -1);
  std::vector<int>& link_k = link_lane_out;   // limb links: lane; trunk links: -1
  std::vector<int>& link_j = link_pos_out;    // limb links: position in the chain; trunk links: trunk depth (0 = base)
  link_k.assign(m.num_links, -2);
  link_j.assign(m.num_links, -1);
  link_k[0] = -1; link_j[0] = 0;
  T.trunk_restart = 1u;
  for (int d = 0; d <= MAX_NW; ++d) T.trunk_anc[d] = 0u;
  for (int i = 0; i < m.num_trunk; ++i) {
    int link = m.trunk_link[i];
    if (m.trunk_parent[i] != 0 && m.trunk_parent[i] != -1) return fail("trunk_parent: 0 (hangs off its predecessor in trunk_link) or -1 (hangs off the base)");
    const int pd = (i == 0 || m.trunk_parent[i] == -1) ? 0 : i;  // trunk depth of the parent link
    if (link < 1 || link >= m.num_links || m.link_parent[link] != (pd == 0 ? 0 : m.trunk_link[pd - 1])) return fail("trunk_link does not describe serial chains off the base");
    link_k[link] = -1; link_j[link] = i + 1;
    if (pd == 0) T.trunk_restart |= 1u << i;
    T.trunk_anc[i + 1] = (pd == 0 ? 0u : T.trunk_anc[pd]) | (1u << i);
  }
  auto fill_joint = [&](LaneTab& L, int jx, int link) {
    const int jt = link - 1;
    // (the axis as a UNIT vector: Rodrigues' formula and the motion subspace assume one; fp32 components of a tilted axis are off by ~2e-9 -
    // FFTAI GR1 -, which the fp64 lane program sees against the oracle; in fp32 the normalised components round to the same numbers)
    const double an = sqrt((double)m.link_axis[link][0] * m.link_axis[link][0] + (double)m.link_axis[link][1] * m.link_axis[link][1] + (double)m.link_axis[link][2] * m.link_axis[link][2]);
    for (int c = 0; c < 3; ++c) { L.origin[jx][c] = m.link_origin[link][c]; L.axis[jx][c] = an > 0.0 ? (float)((double)m.link_axis[link][c] / an) : 0.f; }
    quat_to_rows(m.link_quat[link], L.rot0[jx]);
    L.lower[jx] = m.joint_lower[jt]; L.upper[jx] = m.joint_upper[jt]; L.vel_limit[jx] = m.joint_vel_limit[jt];
    L.armature[jx] = m.joint_armature[jt]; L.q0[jx] = m.default_joint_pos[jt]; L.qd0[jx] = m.default_joint_vel[jt];
    L.soft_lo[jx] = m.soft_lower[jt]; L.soft_hi[jx] = m.soft_upper[jt];
    L.act_implicit[jx] = m.act_implicit[jt]; L.kp0[jx] = m.act_kp[jt]; L.kd0[jx] = m.act_kd[jt];
    L.eff[jx] = m.act_effort_limit[jt]; L.sat[jx] = m.act_saturation[jt]; L.act_vlim[jx] = m.act_vel_limit[jt];
    L.action_is_vel[jx] = m.action_is_vel[jt]; L.a_scale[jx] = m.action_scale[jt]; L.a_off[jx] = m.action_offset[jt];
    L.a_lo[jx] = m.action_clip_lo[jt]; L.a_hi[jx] = m.action_clip_hi[jt];
    L.joint_id[jx] = jt;
  };
  for (int k = 0; k < NLANE; ++k) {
    LaneTab& L = T.lane[k];
    for (int s = 0; s < MAX_NBS; ++s) { L.slot_body[s] = -1; L.slot_grp[s] = 0; }
    for (int g = 0; g < MAX_NGRP; ++g)
      for (int s = 0; s < MAX_SPL; ++s) L.sph_r[g][s] = -1.f;
    L.base_body_local = -1;
    L.nj = m.chain_nj[k];
    L.attach = m.chain_attach[k];
    // a lane's link group 0 is a share of the trunk link its limb hangs off - or of the spine link the model compiler gave it
    // (rl_model_desc.chain_grp0: GR1's head, which no limb hangs off)
    L.grp0_depth = m.chain_grp0[k] > 0 ? m.chain_grp0[k] - 1 : L.attach;
    if (L.grp0_depth < 0 || L.grp0_depth > m.num_trunk) return fail("chain_grp0 names a trunk link the model does not have");
    if (L.nj > CL || L.attach < 0 || L.attach > m.num_trunk) return fail("bad chain description");
    for (int jx = 0; jx < MAX_JX; ++jx) {  // padding joints: inert (axis 0, no gains), velocity limit > 0 so clamps are no-ops
      L.joint_id[jx] = -1; L.joint_own[jx] = 0; L.vel_limit[jx] = 1e9f; L.act_vlim[jx] = 1e9f; L.lower[jx] = -1e9f; L.upper[jx] = 1e9f;
      L.a_lo[jx] = -1e30f; L.a_hi[jx] = 1e30f; L.act_implicit[jx] = 1;
      L.rot0[jx][0] = L.rot0[jx][4] = L.rot0[jx][8] = 1.f;
    }
    for (int j = 0; j < L.nj; ++j) {
      int link = m.chain_link[k][j];
      int parent = j == 0 ? (L.attach == 0 ? 0 : m.trunk_link[L.attach - 1]) : m.chain_link[k][j - 1];
      if (link < 1 || link >= m.num_links || m.link_parent[link] != parent) return fail("chain_link does not describe serial chains off the trunk");
      link_k[link] = k; link_j[link] = j;
      fill_joint(L, j, link);
      L.joint_own[j] = 1;
    }
    for (int i = 0; i < m.num_trunk; ++i) {
      fill_joint(L, CL + i, m.trunk_link[i]);
      L.joint_own[CL + i] = k == 0 ? 1 : 0;
    }
  }
  for (int l = 0; l < m.num_links; ++l)
    if (link_k[l] == -2) return fail("link " + std::to_string(l) + " belongs to neither the trunk nor a limb chain");
  // self-collision: capsules and capsule pairs dealt to the env's 16 virtual lanes (env_tables.h SelfLaneTab)
  for (int v = 0; v < SELF_CAPS; ++v) {
    SelfLaneTab& sl = T.self_lane[v];
    memset(&sl, 0, sizeof(sl));
    sl.cap = sl.frame = -1;
    for (int i = 0; i < SELF_PPL; ++i) sl.pair[i] = -1;
  }
  if (m.num_self_pairs > 0) {
    if (NW == 0) return fail("self-collision pairs need the trunk + limbs instance");
    if (m.num_capsules < 2 || m.num_capsules > SELF_CAPS || m.num_capsules > RL_MAX_CAPSULES || m.num_self_pairs > SELF_CAPS * SELF_PPL || m.num_self_pairs > RL_MAX_SELF_PAIRS)
      return fail("self-collision: capsule / pair count out of range");
    int cap_k[SELF_CAPS], cap_g[SELF_CAPS];
    for (int pass = 0; pass < 2; ++pass)  // capsules of limb links first (they must sit with a lane of their limb), trunk links wherever room is left
      for (int c = 0; c < m.num_capsules; ++c) {
        const int l = m.capsule_link[c];
        if (l < 0 || l >= m.num_links) return fail("self-collision: capsule link out of range");
        const bool limb = link_k[l] >= 0;
        if (limb != (pass == 0)) continue;
        int v = -1;
        if (limb) {
          for (int q = 0; q < 4 && v < 0; ++q)
            if (T.self_lane[4 * link_k[l] + q].cap < 0) v = 4 * link_k[l] + q;
        } else {
          for (int q = SELF_CAPS - 1; q >= 0 && v < 0; --q)
            if (T.self_lane[q].cap < 0) v = q;
        }
        if (v < 0) return fail("self-collision: more than four capsules on one limb");
        SelfLaneTab& sl = T.self_lane[v];
        sl.cap = c;
        sl.frame = limb ? link_j[l] : (link_j[l] == 0 ? -1 : CL + link_j[l] - 1);
        for (int q = 0; q < 3; ++q) { sl.p0[q] = m.capsule_p0[c][q]; sl.p1[q] = m.capsule_p1[c][q]; }
        sl.r = m.capsule_radius[c];
        cap_k[c] = limb ? link_k[l] : 7;
        cap_g[c] = limb ? link_j[l] + 1 : link_j[l];
      }
    for (int p = 0; p < m.num_self_pairs; ++p) {
      const int a = m.self_pair[p][0], b = m.self_pair[p][1];
      if (a < 0 || b < 0 || a >= m.num_capsules || b >= m.num_capsules || a == b) return fail("self-collision: pair index out of range");
      T.self_lane[p % SELF_CAPS].pair[p / SELF_CAPS] = a | b << 4 | cap_k[a] << 8 | cap_g[a] << 11 | cap_k[b] << 15 | cap_g[b] << 18;
    }
  }
  // bodies -> lane slots.  Trunk-link bodies with collision spheres: slot 0 of a lane whose group 0 rides
  // on that trunk link; limb bodies: slots 1.. of their lane; sphere-less trunk bodies: any free slot.
  std::vector<int> nsph(m.num_bodies, 0);
  for (int g = 0; g < m.num_spheres; ++g) nsph[m.sphere_body[g]]++;
  int n_base_bodies = 0;
  int next_slot[NLANE] = {1, 1, 1, 1};
  std::vector<int> deferred;
  for (int b = 0; b < m.num_bodies; ++b) {
    int link = m.body_link[b];
    if (link_k[link] == -1) {
      const int depth = link_j[link];
      // the first trunk body keeps the pre-trunk behaviour (A1: `base` always takes lane 0, slot 0)
      if (nsph[b] == 0 && !(NW == 0 && n_base_bodies < NLANE)) { deferred.push_back(b); continue; }
      int k = -1;
      for (int kk = 0; kk < NLANE && k < 0; ++kk)
        if (T.lane[kk].slot_body[0] < 0 && T.lane[kk].grp0_depth == depth) k = kk;
      if (k < 0) {
        if (nsph[b] == 0) { deferred.push_back(b); continue; }
        return fail("no lane left for trunk body " + std::to_string(b));
      }
      LaneTab& L = T.lane[k];
      L.slot_body[0] = b; L.slot_grp[0] = 0; L.base_body_local = n_base_bodies++; L.owns_base_body = 1;
      for (int c = 0; c < 3; ++c) L.slot_pos[0][c] = m.body_pos[b][c];
      body_lane[b] = k; body_slot[b] = 0;
    } else {
      int k = link_k[link], j = link_j[link];
      LaneTab& L = T.lane[k];
      int s = next_slot[k]++;
      if (s >= NBS) return fail("too many bodies on one chain");
      L.slot_body[s] = b; L.slot_grp[s] = j + 1;
      for (int c = 0; c < 3; ++c) L.slot_pos[s][c] = m.body_pos[b][c];
      body_lane[b] = k; body_slot[b] = s;
    }
  }
  for (int b : deferred) {  // sensor timers only (no spheres report to these slots)
    int k = -1;
    for (int kk = 0; kk < NLANE && k < 0; ++kk)
      if (next_slot[kk] < NBS) k = kk;
    if (k < 0) return fail("no free body slot for trunk body " + std::to_string(b));
    LaneTab& L = T.lane[k];
    int s = next_slot[k]++;
    L.slot_body[s] = b; L.slot_grp[s] = 0;
    for (int c = 0; c < 3; ++c) L.slot_pos[s][c] = m.body_pos[b][c];
    body_lane[b] = k; body_slot[b] = s;
  }
  T.n_base_bodies = n_base_bodies;
  // spheres -> lane / group / slot.  A trunk body with more spheres than its lane has slots also uses the lanes
  // riding on the same trunk link that own no trunk body: first every such body reserves the lanes it needs
  // (MagicLab Dog-W: 6-sphere base + 4-sphere head), then the lanes still free go to the first of them in body
  // order (A1 trunk: 8 corner spheres -> 2 per lane); the spheres are dealt round-robin over a body's lanes.
  // Limb spheres first: a merged instance (Topo<..., M0 = 1>) hosts the lane's share of the trunk link's spheres in the sphere
  // slots its limb link groups leave free, so a lane's room for trunk spheres is known only then.  When the trunk spheres do not
  // fit, the caller builds the tables again unmerged.
  int fill[NLANE][MAX_NGRP];
  memset(fill, 0, sizeof(fill));
  auto put_sphere = [&](int k, int grp, int g, int slot) {
    LaneTab& L = T.lane[k];
    const int s = fill[k][grp]++;
    for (int c = 0; c < 3; ++c) L.sph_c[grp][s][c] = m.sphere_center[g][c];
    L.sph_r[grp][s] = m.sphere_radius[g];
    L.sph_slot[grp][s] = slot;
    return s;
  };
  for (int g = 0; g < m.num_spheres; ++g) {
    const int b = m.sphere_body[g], link = m.body_link[b];
    if (link_k[link] == -1) continue;
    const int k = link_k[link], grp = link_j[link] + 1;
    if (fill[k][grp] >= SPL) return fail("too many collision spheres on one link group (body " + std::to_string(b) + ")");
    put_sphere(k, grp, g, body_slot[b]);
  }
  int base_cap[NLANE], base_fill[NLANE] = {0, 0, 0, 0};
  for (int k = 0; k < NLANE; ++k) {
    base_cap[k] = 0;
    for (int g = merge ? 1 : 0; g <= (merge ? CL : 0); ++g) base_cap[k] += SPL - fill[k][g];
  }
  std::vector<uint32_t> lanes_of(m.num_bodies, 0u);
  for (int pass = 0; pass < 2; ++pass)
    for (int b = 0; b < m.num_bodies; ++b) {
      const int link = m.body_link[b];
      if (link_k[link] != -1 || nsph[b] == 0 || body_lane[b] < 0 || body_slot[b] != 0) continue;
      if (pass == 0 && nsph[b] <= base_cap[body_lane[b]]) continue;
      if (pass == 1 && lanes_of[b] == 0u) continue;
      if (pass == 0) lanes_of[b] = 1u << body_lane[b];
      int room = 0;
      for (int kk = 0; kk < NLANE; ++kk) room += ((lanes_of[b] >> kk) & 1u) ? base_cap[kk] : 0;
      for (int kk = 0; kk < NLANE && (pass == 1 || room < nsph[b]); ++kk)
        if (T.lane[kk].base_body_local == -1 && T.lane[kk].grp0_depth == link_j[link]) {
          T.lane[kk].base_body_local = T.lane[body_lane[b]].base_body_local;
          lanes_of[b] |= 1u << kk;
          room += base_cap[kk];
        }
    }
  int rr = 0;
  for (int g = 0; g < m.num_spheres; ++g) {
    const int b = m.sphere_body[g], link = m.body_link[b];
    if (link_k[link] != -1) continue;
    int k = body_lane[b];
    if (lanes_of[b] != 0u)
      for (int tries = 0; tries < NLANE; ++tries, ++rr) {
        int kk = rr % NLANE;
        if (((lanes_of[b] >> kk) & 1u) && base_fill[kk] < base_cap[kk]) { k = kk; ++rr; break; }
      }
    if (base_fill[k] >= base_cap[k]) return fail("too many collision spheres on one link group (body " + std::to_string(b) + ")");
    ++base_fill[k];
    int grp = 0;
    if (merge) {  // a free slot of a limb link group: one that has spheres of its own first (its contact code runs anyway)
      for (int gg = 1; gg <= CL && grp == 0; ++gg)
        if (fill[k][gg] > 0 && fill[k][gg] < SPL) grp = gg;
      for (int gg = 1; gg <= CL && grp == 0; ++gg)
        if (fill[k][gg] < SPL) grp = gg;
    }
    const int s = put_sphere(k, grp, g, 0);
    if (merge) T.lane[k].sph_base_mask |= 1u << (grp * SPL + s);
  }
  // 16-lanes-per-env mapping: body slots each sub-lane updates (the slots of the link groups it evaluates)
  for (int k = 0; k < NLANE; ++k) {
    LaneTab& L = T.lane[k];
    const int maxown = NBS > 6 ? 4 : 2;
    for (int q = 0; q < 4; ++q) {
      int n = 0;
      for (int i = 0; i < LaneTab::MAXOWN; ++i) L.own_slot[q][i] = -1;
      for (int sl = 0; sl < NBS; ++sl) {
        const bool used = L.slot_body[sl] >= 0 || (sl == 0 && L.base_body_local >= 0);
        // the sub-lane that evaluates the slot's link group: g % 4, merged instances (g - 1) % 4 with the trunk slot on sub-lane 0
        const int gq = merge ? (L.slot_grp[sl] == 0 ? 0 : (L.slot_grp[sl] - 1) % 4) : L.slot_grp[sl] % 4;
        if (!used || gq != q) continue;
        if (n >= maxown) return fail("too many body slots on one sub-lane");
        L.own_slot[q][n++] = sl;
      }
    }
    // ... and with two sub-lanes per limb (quadrupeds; the trunk + limbs instance has the 16-lane mapping only)
    for (int q = 0; q < 2; ++q) {
      int n = 0;
      for (int i = 0; i < LaneTab::MAXOWN2; ++i) L.own_slot2[q][i] = -1;
      for (int sl = 0; sl < NBS && NW == 0; ++sl) {
        const bool used = L.slot_body[sl] >= 0 || (sl == 0 && L.base_body_local >= 0);
        const int gq = merge ? (L.slot_grp[sl] == 0 ? 0 : (L.slot_grp[sl] - 1) % 2) : L.slot_grp[sl] % 2;
        if (!used || gq != q) continue;
        if (n >= LaneTab::MAXOWN2) return fail("too many body slots on one sub-lane (two sub-lanes per limb)");
        L.own_slot2[q][n++] = sl;
      }
    }
  }
  // ... and with eight (trunk + limbs instances: a wavefront then holds two envs instead of four, half the limb-shared LDS, and a
  // 2048-env launch puts a wavefront on every SIMD).  A model that needs more rows than MAXOWN8 keeps the 16-lane mapping (sub8_ok).
  T.sub8_ok = NW > 0 && !merge ? 1 : 0;
  for (int k = 0; k < NLANE; ++k) {
    LaneTab& L = T.lane[k];
    for (int q = 0; q < 8; ++q) {  // (the lanes find their slots themselves, by the same rule: env_step.h EnvLane constructor)
      int n = 0;
      for (int sl = 0; sl < NBS && NW > 0; ++sl) {
        const bool used = L.slot_body[sl] >= 0 || (sl == 0 && L.base_body_local >= 0);
        if (used && L.slot_grp[sl] % 8 == q && ++n > LaneTab::MAXOWN8) T.sub8_ok = 0;
      }
    }
  }
  // bodies the events / the scanner address
  {
    const int wl = m.body_link[d.task.base_body], sl = m.body_link[d.task.scan_body];
    if (link_k[wl] != -1 || link_k[sl] != -1) return fail("the base body and the scanner body must sit on trunk links");
    if (m.body_link[0] != 0) return fail("body 0 must be the root body");
    T.wrench_depth = link_j[wl];
    T.scan_depth = link_j[sl];
    for (int c = 0; c < 3; ++c) T.scan_pos[c] = m.body_pos[d.task.scan_body][c];
    if (NW == 0 && d.task.base_body != 0) return fail("quadruped instances need the base body to be the root body");
  }
  T.slot_valid = 0;
  for (int k = 0; k < NLANE; ++k)
    for (int g = 0; g < NGRP; ++g)
      for (int s2 = 0; s2 < SPL; ++s2)
        if (T.lane[k].sph_r[g][s2] > 0.f) T.slot_valid |= 1u << (g * SPL + s2);
  // sim / terrain / task scalars
  const rl_sim_desc& s = d.sim;
  T.dt = s.dt; T.decimation = s.decimation; T.gravity = s.gravity; T.contact_k = s.contact_k; T.contact_c = s.contact_c;
  T.contact_phi_ref = s.contact_phi_ref; T.contact_ct = s.contact_ct; T.contact_vdep = s.contact_vdep; T.contact_vstick = s.contact_vstick;
  T.limit_k = s.limit_k; T.limit_c = s.limit_c; T.force_threshold = s.force_threshold;
  const rl_terrain_desc& tr = d.terrain;
  T.is_plane = tr.is_plane; T.nx = tr.nx; T.ny = tr.ny; T.hscale = tr.hscale; T.x0 = tr.x0; T.y0 = tr.y0;
  T.num_rows = tr.num_rows; T.num_cols = tr.num_cols; T.tile_size = tr.tile_size; T.border = tr.border; T.curriculum = tr.curriculum;
  const rl_task_desc& t = d.task;
  T.step_dt = s.dt * (float)s.decimation;
  T.max_episode_length_s = t.episode_length_s;
  T.max_episode_length = (int)ceil((double)t.episode_length_s / ((double)s.dt * s.decimation) - 1e-4);  // dt is fp32: 20 / (4 * 0.005f) = 1000.00002
  memcpy(T.cmd_range, t.cmd_range, sizeof(T.cmd_range));
  memcpy(T.cmd_resample, t.cmd_resample, sizeof(T.cmd_resample));
  T.cmd_rel_standing = t.cmd_rel_standing; T.cmd_rel_heading = t.cmd_rel_heading; T.cmd_heading_stiffness = t.cmd_heading_stiffness;
  T.cmd_small_threshold = t.cmd_small_threshold; T.cmd_heading = t.cmd_heading;
  T.cur_lin = t.cur_cmd_lin; T.cur_ang = t.cur_cmd_ang; T.cur_lin_term = t.cur_cmd_lin_term; T.cur_ang_term = t.cur_cmd_ang_term;
  if ((t.cur_cmd_lin && (t.cur_cmd_lin_term < 0 || t.cur_cmd_lin_term >= t.n_rewards)) || (t.cur_cmd_ang && (t.cur_cmd_ang_term < 0 || t.cur_cmd_ang_term >= t.n_rewards)))
    return fail("command_levels curriculum names a reward term that does not exist");
  if (t.n_policy > MAX_OBS || t.n_critic > MAX_OBS || t.n_rewards > MAX_T) return fail("too many terms");
  T.n_policy = t.n_policy; T.n_critic = t.n_critic; T.policy_corrupt = t.policy_corrupt; T.critic_corrupt = t.critic_corrupt;
  for (int grp = 0; grp < 2; ++grp) {
    const rl_obs_term* src = grp == 0 ? t.policy : t.critic;
    ObsTab* dst = grp == 0 ? T.policy : T.critic;
    int n = grp == 0 ? t.n_policy : t.n_critic, off = 0;
    for (int i = 0; i < n; ++i) {
      dst[i].kind = src[i].kind; dst[i].scale = src[i].scale; dst[i].clip_lo = src[i].clip_lo; dst[i].clip_hi = src[i].clip_hi;
      dst[i].noise_lo = src[i].noise_lo; dst[i].noise_hi = src[i].noise_hi; dst[i].has_noise = src[i].has_noise; dst[i].offset = off;
      off += obs_term_dim(d, src[i].kind);
    }
    (grp == 0 ? T.policy_dim : T.critic_dim) = off;
    // per-column descriptors (env_tables.h ObsGroupTabT): what the lane program reads
    auto& G = T.obs[grp];
    const int corrupt = grp == 0 ? t.policy_corrupt : t.critic_corrupt;
    const int NCmax = 12 + 3 * (NLANE * CL + NW);
    G.n_cols = 0; G.scan_off = 0; G.scan_n = 0; G.dim = off; G.corrupt = corrupt ? 1 : 0;
    for (int i = 0; i < n; ++i) {
      const bool noisy = corrupt && src[i].has_noise;
      ObsColTab c{src[i].scale, src[i].clip_lo, src[i].clip_hi, noisy ? src[i].noise_lo : 0.f, noisy ? src[i].noise_hi - src[i].noise_lo : 0.f, 0};
      const int w = obs_term_dim(d, src[i].kind);
      if (src[i].kind == RL_OBS_HEIGHT_SCAN) {
        if (G.scan_n > 0) return fail("an observation group can hold one height scan");
        G.scan = c; G.scan_off = dst[i].offset; G.scan_n = w;
        continue;
      }
      if (G.n_cols + w > NCmax) return fail("observation group has more columns than the lane program's column table (" + std::to_string(NCmax) + ")");
      for (int q = 0; q < w; ++q) {
        switch (src[i].kind) {
          case RL_OBS_BASE_LIN_VEL: c.src = FEAT_LIN + q; break;
          case RL_OBS_BASE_ANG_VEL: c.src = FEAT_ANG + q; break;
          case RL_OBS_PROJECTED_GRAVITY: c.src = FEAT_GRAV + q; break;
          case RL_OBS_VELOCITY_COMMANDS: c.src = FEAT_CMD + q; break;
          case RL_OBS_JOINT_POS_REL: c.src = FEAT_JOINT + q; break;
          case RL_OBS_JOINT_VEL_REL: c.src = FEAT_JOINT + m.num_dof + q; break;
          case RL_OBS_LAST_ACTION: c.src = FEAT_JOINT + 2 * m.num_dof + q; break;
          case RL_OBS_JOINT_POS_REL_NO_WHEEL: c.src = FEAT_JOINT + 3 * m.num_dof + q; break;
          default: return fail("unknown observation kind");
        }
        G.col[G.n_cols++] = c;
      }
    }
    // the non-scan columns are stored densely: column of ordinal n is n (before the scan) or n + scan_n (after it)
  }
  T.scan_nx = t.scan_nx; T.scan_ny = t.scan_ny; T.scan_res = t.scan_res; T.scan_offset = t.scan_offset; T.wheel_joint_mask = t.wheel_joint_mask;
  T.n_rewards = t.n_rewards;
  int pool_used = 0;
  for (int i = 0; i < t.n_rewards; ++i) {
    const rl_reward_term& r = t.rewards[i];
    if (r.kind < 0 || r.kind >= RL_REW_NUM_KINDS) return fail("unknown reward kind");
    RewTab& R = T.rew[i];
    R.kind = r.kind; R.weight = r.weight; memcpy(R.p, r.p, sizeof(R.p)); R.joint_mask = r.joint_mask; R.body_mask = r.body_mask;
    R.n_idx = r.n_idx; R.idx_off = pool_used;
    if (r.n_idx < 0 || r.n_idx > 16 || pool_used + (r.kind == RL_REW_FEET_GAIT ? 4 : r.n_idx) > IDX_POOL) return fail("index lists of the reward terms exceed the pool");
    const int nidx = r.kind == RL_REW_FEET_GAIT ? 4 : r.n_idx;
    for (int q = 0; q < nidx; ++q) { T.idx_pool_a[pool_used + q] = r.idx_a[q]; T.idx_pool_b[pool_used + q] = r.idx_b[q]; }
    pool_used += nidx;
    // how the lane that evaluates the term finds its inputs (env_terms.h term_value): joint-sum kinds name a row of the
    // joint-statistics table (same order as the JS_* enum there); kinds that read a body's position / velocity relative
    // to the root add their bodies to rew_rel_mask
    switch (r.kind) {
      case RL_REW_JOINT_TORQUES_L2: R.row = 0; break;
      case RL_REW_JOINT_ACC_L2: R.row = 1; break;
      case RL_REW_JOINT_VEL_L2: R.row = 2; break;
      case RL_REW_JOINT_POS_LIMITS: R.row = 3; break;
      case RL_REW_JOINT_POWER: R.row = 4; break;
      case RL_REW_JOINT_DEVIATION_L1: case RL_REW_STAND_STILL: R.row = 5; break;
      case RL_REW_JOINT_POS_PENALTY: R.row = 6; break;
      case RL_REW_ACTION_RATE_L2: R.row = 7; R.joint_mask = 0xffffffffu; break;  // every action dimension (cut to D below)
      default: R.row = -1; break;
    }
    R.joint_mask &= m.num_dof >= 32 ? 0xffffffffu : ((1u << m.num_dof) - 1u);  // the lane that sums a row reads 8 columns per trip
    for (int q = 0; q < nidx; ++q)  // index lists address the tables directly: validate them here, not in the kernel (and before any shift by them)
      if (r.idx_a[q] < 0 || r.idx_a[q] >= RL_MAX_BODIES || r.idx_a[q] >= 64 || r.idx_b[q] < 0 || r.idx_b[q] >= RL_MAX_BODIES) return fail("reward term index list out of range");
    if (r.kind == RL_REW_ACTION_MIRROR || r.kind == RL_REW_ACTION_SYNC)  // the lists index the action buffer (and, action_sync: name one of 8 groups)
      for (int q = 0; q < nidx; ++q)
        if (r.idx_a[q] >= m.num_dof || (r.kind == RL_REW_ACTION_MIRROR ? r.idx_b[q] >= m.num_dof : r.idx_b[q] >= 8)) return fail("action_mirror / action_sync index list out of range");
    if (r.kind == RL_REW_FEET_HEIGHT_BODY || r.kind == RL_REW_FEET_SLIDE || r.kind == RL_REW_FEET_HEIGHT || r.kind == RL_REW_HANDSTAND_FEET_HEIGHT_EXP)
      T.rew_rel_mask |= r.body_mask;
    if (r.kind == RL_REW_FEET_DISTANCE_Y_EXP || r.kind == RL_REW_FEET_DISTANCE_XY_EXP)
      for (int q = 0; q < r.n_idx; ++q) T.rew_rel_mask |= 1ull << r.idx_a[q];
    if (r.kind == RL_REW_FEET_STUMBLE) T.rew_ext_mask |= r.body_mask;  // reads the net force
  }
  T.rew_ext_mask |= T.rew_rel_mask;
  {  // evaluation schedule (TaskTab::rew_slot): scalar kinds last; whatever does not fit the first 16 slots and is scalar goes to the mini-trip
    int ns = 0;
    for (int pass = 0; pass < 2; ++pass)
      for (int i = 0; i < t.n_rewards; ++i)
        if (is_scalar_reward_kind(T.rew[i].kind) == (pass == 1)) T.rew_slot[ns++] = i;
    for (int i = ns; i < MAX_T; ++i) T.rew_slot[i] = 0;
    int n_scalar = 0;
    for (int i = 0; i < t.n_rewards; ++i) n_scalar += is_scalar_reward_kind(T.rew[i].kind) ? 1 : 0;
    const int LPE16 = 16;
    T.n_main = t.n_rewards <= LPE16 ? t.n_rewards : std::max(LPE16, t.n_rewards - n_scalar);
  }
  T.term_time_out = t.term_time_out; T.term_oob = t.term_out_of_bounds; T.term_illegal = t.term_illegal_contact;
  if (const char* tv = std::getenv("RL_ENV_TERMS"))  // RL_ENV_TERMS=0: no termination term, so no env ever resets inside step() (timing A/Bs: what the reset path costs a launch)
    if (atoi(tv) == 0) {
      T.term_time_out = T.term_oob = T.term_illegal = 0;
      fprintf(stderr, "[rl_env] WARNING: RL_ENV_TERMS=0 - every termination term of the task is DISABLED (timing ablation): no env resets on a fall or a time-out\n");
    }
  T.oob_buffer = t.oob_buffer; T.illegal_threshold = t.illegal_threshold; T.illegal_body_mask = t.illegal_body_mask;
  T.ev_wrench = t.ev_wrench; T.ev_reset_joints = t.ev_reset_joints; T.ev_gains = t.ev_gains; T.ev_reset_base = t.ev_reset_base; T.ev_push = t.ev_push;
  memcpy(T.wrench_force, t.wrench_force, sizeof(T.wrench_force)); memcpy(T.wrench_torque, t.wrench_torque, sizeof(T.wrench_torque));
  memcpy(T.reset_jpos, t.reset_joint_pos_scale, sizeof(T.reset_jpos)); memcpy(T.reset_jvel, t.reset_joint_vel_scale, sizeof(T.reset_jvel));
  memcpy(T.gain_kp, t.gain_kp_scale, sizeof(T.gain_kp)); memcpy(T.gain_kd, t.gain_kd_scale, sizeof(T.gain_kd));
  memcpy(T.reset_pose, t.reset_pose, sizeof(T.reset_pose)); memcpy(T.reset_vel, t.reset_vel, sizeof(T.reset_vel));
  memcpy(T.push_interval, t.push_interval, sizeof(T.push_interval)); memcpy(T.push_vel, t.push_vel, sizeof(T.push_vel));
  memcpy(T.default_root_pos, m.default_root_pos, sizeof(T.default_root_pos)); memcpy(T.default_root_quat, m.default_root_quat, sizeof(T.default_root_quat));
  if (const char* iv = std::getenv("RL_ENV_INTERVALS"))  // RL_ENV_INTERVALS=0: no push event, no command resampling between resets (timing A/Bs: what the interval events cost a launch)
    if (atoi(iv) == 0) {
      T.ev_push = 0; T.cmd_resample[0] = T.cmd_resample[1] = 1e9f;
      fprintf(stderr, "[rl_env] WARNING: RL_ENV_INTERVALS=0 - the push event and the command resampling between resets are DISABLED (timing ablation)\n");
    }
  return 0;
}

// descriptor -> Tables as rl_env_create does it (tools/gen_specs.py compiles the specialised term stacks from the same call)
inline int compile_tables(const rl_env_desc& d, Tables& tables, std::vector<int>& body_lane, std::vector<int>& body_slot, std::vector<int>& link_lane,
                          std::vector<int>& link_pos) {
  // wheeled quadrupeds (4-joint limbs): the merged instance when the trunk's spheres fit the free sphere slots of the limbs
  // (Go2W, ZSL1W, M20, Dog-W), else the instance with a link group for the trunk share.  RL_ENV_MERGE=0: never merged, 2: whenever it fits.
  const char* mv = std::getenv("RL_ENV_MERGE");
  int cl_, nw_, spl_, nbs_, rotpad_ = 0;
  if (topo_shape(d.model, cl_, nw_, spl_, nbs_, rotpad_)) return -1;
  const bool want = d.model.num_trunk == 0 && d.model.chain_len == 4 && !rotpad_ && !(mv && atoi(mv) == 0);
  bool merged = want && build_tables(d, tables, body_lane, body_slot, link_lane, link_pos, true) == 0;
  // it pays when the last link group (the wheels) has spheres - the group that costs the unmerged instance a second contact
  // pass; without (B2W) the unmerged instance skips that pass anyway and is 1 % faster (profiles/r02_merged_wheeled.txt)
  if (merged && ((tables.slot_valid >> (tables.CL * tables.SPL)) & ((1u << tables.SPL) - 1u)) == 0u && !(mv && atoi(mv) == 2)) merged = false;
  if (!merged && build_tables(d, tables, body_lane, body_slot, link_lane, link_pos, false)) return -1;
  return 0;
}

// packed (per-instance) table image that the env kernels stage into LDS
inline size_t packed_size(const TaskTab& T) {
  return T.NW > 3 ? sizeof(TablesT<TopoGR>) : T.NW > 0 ? sizeof(TablesT<TopoG1>) : (T.rotpad ? sizeof(TablesT<TopoQuad4R>) : T.CL == 4 ? sizeof(TablesT<TopoQuad4>) : sizeof(TablesT<TopoQuad3>));
}
template <class TP>
inline size_t staged_bytes_t(const TaskTab& T) {  // `rew` is the last member: everything up to its first n_rewards entries
  return (sizeof(TablesBody<TP>) - (size_t)(MAX_T - T.n_rewards) * sizeof(RewTab) + 15) / 16 * 16;
}
inline size_t staged_bytes(const TaskTab& T) {
  return T.NW > 3 ? staged_bytes_t<TopoGR>(T) : T.NW > 0 ? staged_bytes_t<TopoG1>(T) : (T.rotpad ? staged_bytes_t<TopoQuad4R>(T) : T.CL == 4 ? staged_bytes_t<TopoQuad4>(T) : staged_bytes_t<TopoQuad3>(T));
}
inline std::vector<uint8_t> pack_image(const Tables& T) {
  std::vector<uint8_t> img(packed_size(T), 0);
  if (T.NW > 3) pack_tables<TopoGR>(T, *reinterpret_cast<TablesT<TopoGR>*>(img.data()));
  else if (T.NW > 0) pack_tables<TopoG1>(T, *reinterpret_cast<TablesT<TopoG1>*>(img.data()));
  else if (T.rotpad) pack_tables<TopoQuad4R>(T, *reinterpret_cast<TablesT<TopoQuad4R>*>(img.data()));
  else if (T.CL == 4) pack_tables<TopoQuad4>(T, *reinterpret_cast<TablesT<TopoQuad4>*>(img.data()));
  else pack_tables<TopoQuad3>(T, *reinterpret_cast<TablesT<TopoQuad3>*>(img.data()));
  return img;
}

// ------------------------------------------------------------------------------------------------
// The environment object behind the opaque rl_env*.
// ------------------------------------------------------------------------------------------------
template <class Backend>
struct EnvImpl {
  Backend be;
  rl_env_desc desc;
  Tables tables;
  Tables* tables_dev = nullptr;   // unpacked (export / import kernels)
  void* packed_dev = nullptr;     // TablesT<Topo> image the env kernels stage into LDS
  KState S;
  CmdLevelParams cmd_level_params{};
  int spec_id = 0;  // env_spec.h: the specialised step kernel this env runs (0: the interpreter)
  int N = 0, Npad = 0, D = 0, B = 0, CL = 0, inst = 0, ept = ENVS_PER_WAVE;  // inst: lane-program instance key (CL, + 100 merged, + 200 six-joint trunk, + 400 rot / pad quadruped)
  uint64_t seed = 0;
  uint32_t step_counter = 0;
  // the kernels take the step count as *step_base + launch literal (rl_env_graph_*): `anchor` mirrors the device word
  uint32_t* step_base = nullptr;
  uint32_t anchor = 0;
  bool capturing = false;
  uint32_t snap_step = 0, graph_n = 0;
  int snap_slot = 0, graph_slot = -1;
  std::vector<int> body_lane, body_slot, link_lane, link_pos;
  std::vector<void*> allocs;
  // AoS inspection buffers
  float *root_state = nullptr, *joint_pos = nullptr, *joint_vel = nullptr, *cforce = nullptr, *ctimers = nullptr, *action_aos = nullptr;
  float *env_origin_aos = nullptr, *task_state = nullptr, *gains = nullptr;
  // the two observation groups alternate between two HBM buffers (include/rl_env.h "Ownership"): obs_slot = the one last written
  float* obs_ring[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [group][slot]
  int obs_slot = 0;
  bool alloc_failed = false;
  uint8_t* reset_mask = nullptr;
  float* terrain_dev = nullptr;
  float* terrain_origins_dev = nullptr;

  template <class Tp>
  Tp* alloc(size_t n) {
    void* p = be.alloc(n * sizeof(Tp));
    if (p) {
      be.zero(p, n * sizeof(Tp));
      allocs.push_back(p);
    } else {
      alloc_failed = true;  // checked once at the end of create(): a kernel must never see a null buffer
    }
    return (Tp*)p;
  }

  int create(const rl_env_desc* d, const float* terrain_heights, const float* terrain_origins, const float* env_origins,
             int32_t num_envs, uint64_t seed_, int32_t device) {
    desc = *d;
    seed = seed_;
    N = num_envs;
    Npad = (N + ENVS_PER_WAVE - 1) / ENVS_PER_WAVE * ENVS_PER_WAVE;  // multiple of 16 suits both lane mappings
    D = d->model.num_dof;
    B = d->model.num_bodies;
    if (N <= 0) return fail("num_envs must be positive");
    if (compile_tables(*d, tables, body_lane, body_slot, link_lane, link_pos)) return -1;
    // a step kernel specialised on exactly this task (env_spec.h)?  RL_ENV_SPEC=0: the interpreter (A/B runs, parity of the two paths)
    spec_id = 0;
    if (!(std::getenv("RL_ENV_SPEC") && atoi(std::getenv("RL_ENV_SPEC")) == 0)) {
#define RL_SPEC_MATCH(NAME, ID) if (spec_id == 0 && spec_matches<NAME>(tables)) spec_id = ID;
      RL_SPEC_LIST(RL_SPEC_MATCH)
#undef RL_SPEC_MATCH
      // ... or one compiled for this task at run time and registered with the library (rl_env_register_spec_plugin: robot_lab_amd/jit.py)
      if (spec_id == 0) spec_id = be.match_plugin(&tables);
    }
    be.spec_id = spec_id;
    CL = tables.CL;
    inst = tables.CL + (tables.merged ? 100 : 0) + (tables.NW > 3 ? 200 : 0) + (tables.rotpad ? 400 : 0);
    if (be.init(device)) return fail("device init failed: " + be.error());
    ept = be.envs_per_wave(tables, Npad);  // the lane mapping (16 or 4 lanes per env) decides the layout of the state tiles
    if (std::getenv("RL_ENV_DEBUG")) fprintf(stderr, "rl_env: lane program CL %d NW %d merged %d rot/pad %d, %d envs per wavefront\n", tables.CL, tables.NW, tables.merged, tables.rotpad, ept);
    if (be.configure(tables)) return fail("kernel configuration failed: " + be.error());
    const size_t Np = Npad, ntile = Npad / ept;
    const Layout ly(tables.CL, tables.NW, tables.NBS);
    memset(&S, 0, sizeof(S));
    S.N = N; S.Npad = Npad; S.seed = seed; S.ept = ept; S.table_bytes = (uint32_t)staged_bytes(tables);
    S.self_k = d->model.num_self_pairs > 0 ? d->sim.self_k : 0.f;  // (build_tables has checked that the instance can run the pass)
    S.self_trips = (d->model.num_self_pairs + SELF_CAPS - 1) / SELF_CAPS;
    if (const char* sv = std::getenv("RL_ENV_SELF"))  // RL_ENV_SELF=0: the pass off although the model lists pairs (timing A/Bs, diagnostics)
      if (atoi(sv) == 0) S.self_k = 0.f;
    S.lane_state = alloc<float>(ntile * (size_t)ly.NF_LANE * NLANE * ept);
    S.env_state = alloc<float>(ntile * (size_t)ly.NF_ENV * ept);
    S.flags = alloc<int32_t>(Np); S.level = alloc<int32_t>(Np); S.ttype = alloc<int32_t>(Np);
    S.ep_len = alloc<int64_t>(Np); S.ep_sums = alloc<float>(MAX_T * Np);
    for (int g = 0; g < 2; ++g) {  // one allocation per group: [2][Npad][dim] (RL_BUF_OBS_*_RING)
      const size_t row = Np * (size_t)std::max(1, g == 0 ? tables.policy_dim : tables.critic_dim);
      obs_ring[g][0] = alloc<float>(2 * row);
      obs_ring[g][1] = obs_ring[g][0] ? obs_ring[g][0] + row : nullptr;
    }
    obs_slot = 0;
    S.obs_policy = obs_ring[0][0];
    S.obs_critic = obs_ring[1][0];
    S.reward = alloc<float>(Np); S.terminated = alloc<uint8_t>(Np); S.time_out = alloc<uint8_t>(Np);
    S.rew_terms = alloc<float>(MAX_T * Np); S.command_out = alloc<float>(3 * Np); S.log = alloc<float>((size_t)LOG_RING * LOG_PARTS * LOG_SIZE);
    root_state = alloc<float>(Np * 13); joint_pos = alloc<float>(Np * D); joint_vel = alloc<float>(Np * D);
    ctimers = alloc<float>(Np * B * 4); action_aos = alloc<float>(Np * D); env_origin_aos = alloc<float>(Np * 3);
    task_state = alloc<float>(Np * TASK_NF); gains = alloc<float>(Np * 2 * D);
    reset_mask = alloc<uint8_t>(Np);
    S.cmd_levels = alloc<float>(CL_WORDS);
    step_base = alloc<uint32_t>(4);
    S.step_base = step_base;
    tables_dev = alloc<Tables>(1);
    if (alloc_failed) return fail("device allocation failed: " + be.error());
    if (!desc.terrain.is_plane) {
      if (!terrain_heights || !terrain_origins) return fail("heightfield terrain needs heights and sub-terrain origins");
      size_t nh = (size_t)desc.terrain.nx * desc.terrain.ny;
      if (desc.terrain.nx < 2 || desc.terrain.ny < 2) return fail("heightfield terrain needs at least 2 x 2 samples");
      if (TERRAIN_PAIRS) nh = (size_t)(desc.terrain.nx - 1) * desc.terrain.ny * 2;  // rows ix, ix + 1 interleaved (env_step.h TerrainPatch)
      terrain_dev = alloc<float>(nh);
      terrain_origins_dev = alloc<float>((size_t)desc.terrain.num_rows * desc.terrain.num_cols * 3);
      if (alloc_failed) return fail("device allocation failed (terrain): " + be.error());
      if (TERRAIN_PAIRS) {
        std::vector<float> h2(nh);
        const size_t ny = (size_t)desc.terrain.ny;
        for (size_t ix = 0; ix + 1 < (size_t)desc.terrain.nx; ++ix)
          for (size_t iy = 0; iy < ny; ++iy) {
            h2[(ix * ny + iy) * 2] = terrain_heights[ix * ny + iy];
            h2[(ix * ny + iy) * 2 + 1] = terrain_heights[(ix + 1) * ny + iy];
          }
        be.h2d(terrain_dev, h2.data(), nh * sizeof(float));
      } else be.h2d(terrain_dev, terrain_heights, nh * sizeof(float));
      be.h2d(terrain_origins_dev, terrain_origins, (size_t)desc.terrain.num_rows * desc.terrain.num_cols * 3 * sizeof(float));
    } else {
      if (!env_origins) return fail("plane terrain needs env_origins");
      // a one-tile origin table of zeros: reset_env (env_terms.h) reads its tile's origin without a branch on the terrain type, and selects
      terrain_origins_dev = alloc<float>(4);
      if (alloc_failed) return fail("device allocation failed (terrain): " + be.error());
      be.zero(terrain_origins_dev, 16);
    }
    S.terrain = terrain_dev;
    S.terrain_origins = terrain_origins_dev;
    be.h2d(tables_dev, &tables, sizeof(Tables));
    {
      std::vector<uint8_t> img = pack_image(tables);
      packed_dev = alloc<uint8_t>(img.size());
      if (alloc_failed) return fail("device allocation failed: " + be.error());
      be.h2d(packed_dev, img.data(), img.size());
    }
    startup(terrain_origins, env_origins);
    {  // command_levels_* curricula: ranges start at range x range_multiplier[0] (curriculums.py:30-41), end at x [1]
      const rl_task_desc& t = desc.task;
      float lv[CL_WORDS] = {};
      const float ml = t.cur_cmd_lin ? t.cur_cmd_lin_mult[0] : 1.f, ma = t.cur_cmd_ang ? t.cur_cmd_ang_mult[0] : 1.f;
      for (int i = 0; i < 2; ++i) {
        lv[CL_LIN_X + i] = t.cmd_range[0][i] * ml; lv[CL_LIN_Y + i] = t.cmd_range[1][i] * ml; lv[CL_ANG_Z + i] = t.cmd_range[2][i] * ma;
        cmd_level_params.final_x[i] = t.cmd_range[0][i] * t.cur_cmd_lin_mult[1];
        cmd_level_params.final_y[i] = t.cmd_range[1][i] * t.cur_cmd_lin_mult[1];
        cmd_level_params.final_z[i] = t.cmd_range[2][i] * t.cur_cmd_ang_mult[1];
      }
      cmd_level_params.lin = t.cur_cmd_lin; cmd_level_params.ang = t.cur_cmd_ang;
      cmd_level_params.lin_weight = t.cur_cmd_lin ? t.rewards[t.cur_cmd_lin_term].weight : 0.f;
      cmd_level_params.ang_weight = t.cur_cmd_ang ? t.rewards[t.cur_cmd_ang_term].weight : 0.f;
      cmd_level_params.max_episode_length_s = t.episode_length_s;
      be.h2d(S.cmd_levels, lv, sizeof(lv));
    }
    return 0;
  }

  // "startup" events (velocity_env_cfg.py:262-314) [UPSTREAM B8] + initial terrain levels [UPSTREAM B9].
  // Host-side, once; same draws as oracle/env.py:startup_randomisation.
  void startup(const float* terrain_origins, const float* env_origins) {
    const rl_model_desc& m = desc.model;
    const rl_task_desc& t = desc.task;
    const size_t Np = Npad, ntile = Npad / ept;
    const Layout ly(tables.CL, tables.NW, tables.NBS);
    std::vector<float> lane(ntile * (size_t)ly.NF_LANE * NLANE * ept, 0.f), env(ntile * (size_t)ly.NF_ENV * ept, 0.f);
    std::vector<int32_t> level(Np, 0), ttype(Np, 0);
    std::vector<float> bs(64, 1.f), bd(64, 1.f), br(64, 0.f);
    int nb = t.friction_buckets > 0 ? (t.friction_buckets > 64 ? 64 : t.friction_buckets) : 1;
    if (t.ev_material)
      for (int j = 0; j < nb; ++j) {
        bs[j] = uniform_range(seed, GLOBAL_ENV, 0, STREAM_STARTUP, 3 * j, t.friction_static[0], t.friction_static[1]);
        bd[j] = uniform_range(seed, GLOBAL_ENV, 0, STREAM_STARTUP, 3 * j + 1, t.friction_dynamic[0], t.friction_dynamic[1]);
        br[j] = uniform_range(seed, GLOBAL_ENV, 0, STREAM_STARTUP, 3 * j + 2, t.restitution[0], t.restitution[1]);
        bd[j] = fminf(bd[j], bs[j]);
      }
    std::vector<double> lm(m.num_links), lh(m.num_links * 3), lI(m.num_links * 9);
    for (int e = 0; e < Npad; ++e) {
      // per-body mass / com / material
      std::fill(lm.begin(), lm.end(), 0.0); std::fill(lh.begin(), lh.end(), 0.0); std::fill(lI.begin(), lI.end(), 0.0);
      for (int b = 0; b < m.num_bodies; ++b) {
        double mass = m.body_mass[b];
        if (t.ev_mass_base && ((t.mass_base_mask >> b) & 1ull))
          mass += uniform_range(seed, e, 0, STREAM_STARTUP, IDX_MASS_ADD + b, t.mass_base_add[0], t.mass_base_add[1]);
        if (t.ev_mass_others && ((t.mass_scale_mask >> b) & 1ull))
          mass *= uniform_range(seed, e, 0, STREAM_STARTUP, IDX_MASS_SCALE + b, t.mass_scale[0], t.mass_scale[1]);
        if (m.body_mass[b] > 0.f && mass < 1e-6) mass = 1e-6;
        double c[3] = {m.body_com[b][0], m.body_com[b][1], m.body_com[b][2]};
        if (t.ev_com && b == t.base_body)
          for (int a = 0; a < 3; ++a) c[a] += uniform_range(seed, e, 0, STREAM_STARTUP, IDX_COM + 3 * b + a, t.com_range[a][0], t.com_range[a][1]);
        if (b == 0)  // root body: its COM defines root_com_lin_vel [UPSTREAM B3]
          for (int a = 0; a < 3; ++a) env[env_index(ly, e, ly.EF_BASE_COM + a, ept)] = (float)c[a];
        if (b == t.base_body)
          for (int a = 0; a < 3; ++a) env[env_index(ly, e, ly.EF_WR_COM + a, ept)] = (float)c[a];
        double sc = m.body_mass[b] > 0.f ? mass / m.body_mass[b] : 0.0;
        const float* I6 = m.body_inertia[b];
        double Ic[9] = {sc * I6[0], sc * I6[3], sc * I6[4], sc * I6[3], sc * I6[1], sc * I6[5], sc * I6[4], sc * I6[5], sc * I6[2]};
        int l = m.body_link[b];
        double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
        lm[l] += mass;
        for (int a = 0; a < 3; ++a) lh[l * 3 + a] += mass * c[a];
        for (int a = 0; a < 3; ++a)
          for (int bb = 0; bb < 3; ++bb) lI[l * 9 + a * 3 + bb] += Ic[a * 3 + bb] + mass * ((a == bb ? cc : 0.0) - c[a] * c[bb]);  // about the link origin
        // material
        int bucket = 0;
        if (t.ev_material) {
          bucket = (int)floorf(uniform01(seed, e, 0, STREAM_STARTUP, IDX_BUCKET + b) * (float)nb);
          if (bucket > nb - 1) bucket = nb - 1;
        }
        int k = body_lane[b], s = body_slot[b];
        float mu[3] = {t.ev_material ? bs[bucket] : 1.f, t.ev_material ? bd[bucket] : 1.f, t.ev_material ? br[bucket] : 0.f};
        for (int kk = 0; kk < NLANE; ++kk) {
          // base-link bodies may have spheres on other lanes too: replicate their material into slot 0 there
          bool here = kk == k || (s == 0 && tables.lane[kk].base_body_local == tables.lane[k].base_body_local && !tables.lane[kk].owns_base_body);
          if (!here) continue;
          for (int a = 0; a < 3; ++a) lane[lane_index(ly, e, kk, ly.LF_FRICTION + s * 3 + a, ept)] = mu[a];
        }
      }
      // composite per link -> (mass, com, inertia about com)
      for (int l = 0; l < m.num_links; ++l) {
        double mass = lm[l], c[3] = {0, 0, 0};
        if (mass > 0) for (int a = 0; a < 3; ++a) c[a] = lh[l * 3 + a] / mass;
        double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
        double Ic[9];
        for (int a = 0; a < 3; ++a)
          for (int bb = 0; bb < 3; ++bb) Ic[a * 3 + bb] = lI[l * 9 + a * 3 + bb] - mass * ((a == bb ? cc : 0.0) - c[a] * c[bb]);
        float rec[10] = {(float)mass, (float)c[0], (float)c[1], (float)c[2], (float)Ic[0], (float)Ic[4], (float)Ic[8], (float)Ic[1], (float)Ic[2], (float)Ic[5]};
        if (link_lane[l] < 0) {  // trunk link: link_pos = trunk depth
          for (int f = 0; f < 10; ++f) env[env_index(ly, e, ly.EF_BASE_INERTIA + link_pos[l] * INERTIA_NF + f, ept)] = rec[f];
        } else {
          int k = link_lane[l], j = link_pos[l];
          for (int f = 0; f < 10; ++f) lane[lane_index(ly, e, k, ly.LF_INERTIA + j * INERTIA_NF + f, ept)] = rec[f];
        }
      }
      for (int k = 0; k < NLANE; ++k)
        for (int j = 0; j < CL; ++j) {
          lane[lane_index(ly, e, k, ly.LF_KP + j, ept)] = tables.lane[k].kp0[j];
          lane[lane_index(ly, e, k, ly.LF_KD + j, ept)] = tables.lane[k].kd0[j];
          lane[lane_index(ly, e, k, ly.LF_Q + j, ept)] = tables.lane[k].q0[j];
        }
      for (int i = 0; i < tables.NW; ++i) {
        env[env_index(ly, e, ly.EF_TKP + i, ept)] = tables.lane[0].kp0[CL + i];
        env[env_index(ly, e, ly.EF_TKD + i, ept)] = tables.lane[0].kd0[CL + i];
        env[env_index(ly, e, ly.EF_TQ + i, ept)] = tables.lane[0].q0[CL + i];
      }
      // terrain level / type / env origin
      if (desc.terrain.is_plane) {
        int ee = e < N ? e : N - 1;
        for (int a = 0; a < 3; ++a) env[env_index(ly, e, ly.EF_ORIGIN + a, ept)] = env_origins[ee * 3 + a];
      } else {
        int ee = e < N ? e : N - 1;  // padding envs mirror the last real env's cell
        ttype[e] = (int)floor((double)ee / ((double)N / desc.terrain.num_cols));
        int lv = (int)floorf(uniform01(seed, ee, 0, STREAM_STARTUP, IDX_INIT_LEVEL) * (float)(desc.terrain.max_init_level + 1));
        level[e] = lv > desc.terrain.max_init_level ? desc.terrain.max_init_level : lv;
        for (int a = 0; a < 3; ++a) env[env_index(ly, e, ly.EF_ORIGIN + a, ept)] = terrain_origins[((size_t)level[e] * desc.terrain.num_cols + ttype[e]) * 3 + a];
      }
      env[env_index(ly, e, ly.EF_ROOT + 3, ept)] = 1.f;  // identity quaternion until the first reset
      for (int a = 0; a < 3; ++a) env[env_index(ly, e, ly.EF_ROOT + a, ept)] = env[env_index(ly, e, ly.EF_ORIGIN + a, ept)] + m.default_root_pos[a];
    }
    be.h2d(S.lane_state, lane.data(), lane.size() * sizeof(lane[0]));
    be.h2d(S.env_state, env.data(), env.size() * sizeof(env[0]));
    be.h2d(S.level, level.data(), level.size() * sizeof(level[0])); be.h2d(S.ttype, ttype.data(), ttype.size() * sizeof(ttype[0]));
  }

  // The inspection views (applied torque, joint acceleration, net contact force per body) cost 75 extra
  // words per env-step of HBM writes on A1; they are allocated - and from then on written by every step -
  // when a caller first asks for one of them (rl_env_get_buffer).
  int enable_inspection() {
    if (S.dbg_torque) return 0;
    const size_t Np = Npad;
    if (be.activate()) return fail("device activation failed: " + be.error());
    float *t = alloc<float>(Np * D), *a = alloc<float>(Np * D), *c = alloc<float>(Np * B * 3);
    if (!(t && a && c)) return fail("device allocation failed: " + be.error());
    S.dbg_torque = t; S.dbg_acc = a; S.dbg_cforce = c;
    return 0;
  }

  AosPtrs aos() const { return AosPtrs{root_state, joint_pos, joint_vel, ctimers, action_aos, env_origin_aos, task_state, gains}; }

  // next observation buffers: the ones the previous call wrote stay untouched during this one
  void flip_obs(KState& s) {
    obs_slot ^= 1;
    S.obs_policy = s.obs_policy = obs_ring[0][obs_slot];
    S.obs_critic = s.obs_critic = obs_ring[1][obs_slot];
  }

  int reset(const int32_t* env_ids, int32_t n, void* stream) {
    if (be.activate()) return fail("device activation failed: " + be.error());
    KState s = S;
    s.step_counter = step_counter - anchor;
    if (env_ids == nullptr) {
      s.reset_mask = nullptr;
    } else {
      std::vector<uint8_t> mask(Npad, 0);
      for (int i = 0; i < n; ++i) {
        if (env_ids[i] < 0 || env_ids[i] >= N) return fail("env id out of range");
        mask[env_ids[i]] = 1;
      }
      be.h2d_stream(reset_mask, mask.data(), Npad, stream);
      s.reset_mask = reset_mask;
    }
    flip_obs(s);
    s.mode = KMODE_RESET;
    return be.launch(s, packed_dev, inst, stream) ? fail("launch failed: " + be.error()) : 0;
  }

  int step(const float* action_dev, void* stream, const float* ro_values = nullptr, float* ro_rewards = nullptr, uint8_t* ro_dones = nullptr,
           float ro_gamma = 0.f) {
    if (!action_dev) return fail("action pointer is null");
    if ((ro_values || ro_rewards || ro_dones) && !(ro_rewards && ro_dones)) return fail("rollout sink needs rewards and dones (values may be NULL: deferred bootstrap)");
    if (be.activate()) return fail("device activation failed: " + be.error());
    KState s = S;
    s.step_counter = ++step_counter - anchor;
    flip_obs(s);
    s.action_in = action_dev;
    s.ro_values = ro_values; s.ro_rewards = ro_rewards; s.ro_dones = ro_dones; s.ro_gamma = ro_gamma;
    if (!(tables.cur_lin || tables.cur_ang)) {
      s.mode = KMODE_STEP;
      return be.launch(s, packed_dev, inst, stream) ? fail("launch failed: " + be.error()) : 0;
    }
    // command_levels_* curricula (VEL/mdp/curriculums.py:21-94): the decision of a step whose counter is a multiple of the episode
    // length is a reduction over every env that step resets, and the reference takes it FIRST inside _reset_idx - the commands those
    // resets draw and that step's heading clip already see the widened range.  So such an env steps in three launches: the step up
    // to the rewards (collects the decision's inputs, writes the state back), the one-thread decision (it tests the step count
    // itself: it sits in captured graphs, too), and the rest of the step - resets, commands, push, observations - from the
    // re-loaded state (env_terms.h step_head / step_tail).  Every shipped cfg deletes these terms: one launch, above.
    s.mode = KMODE_STEP_HEAD;
    if (be.launch(s, packed_dev, inst, stream)) return fail("launch failed: " + be.error());
    if (be.launch_cmd_levels(S.cmd_levels, cmd_level_params, S.step_base, step_counter - anchor, (uint32_t)tables.max_episode_length, stream))
      return fail("launch failed: " + be.error());
    s.mode = KMODE_STEP_TAIL;
    return be.launch(s, packed_dev, inst, stream) ? fail("launch failed: " + be.error()) : 0;
  }

  // include/rl_env.h "hipGraph capture of a loop around rl_env_step"
  int set_step_count(uint32_t count) {
    if (capturing) return fail("rl_env_set_step_count inside a capture");
    step_counter = count;  // the launch literal is step_counter - anchor (mod 2^32): nothing to do on the device
    return 0;
  }
  int graph_begin(void* stream) {
    if (capturing) return fail("rl_env_graph_begin: a capture is already open");
    if (be.activate()) return fail("device activation failed: " + be.error());
    if (be.launch_u32(step_base, step_counter, /*add=*/0, stream)) return fail("launch failed: " + be.error());
    anchor = step_counter;
    snap_step = step_counter; snap_slot = obs_slot;
    capturing = true;
    return 0;
  }
  int graph_end(void* stream) {
    if (!capturing) return fail("rl_env_graph_end without rl_env_graph_begin");
    capturing = false;
    const uint32_t n = step_counter - snap_step;
    // capturing executed nothing: back to where the capture started
    step_counter = snap_step;
    if (obs_slot != snap_slot) flip_obs(S);
    if (n == 0 || (n & 1u)) return fail("a captured loop needs an even, positive number of rl_env_step launches (observation buffers alternate), got " + std::to_string(n));
    if (be.launch_u32(step_base, n, /*add=*/1, stream)) return fail("launch failed: " + be.error());
    graph_n = n; graph_slot = snap_slot;
    return (int)n;
  }
  int graph_launching(void* stream) {
    if (capturing || graph_n == 0) return fail("rl_env_graph_launching: no captured loop");
    if (obs_slot != graph_slot) return fail("the env is not where the capture found it: an odd number of steps / resets since (observation slot parity)");
    if (be.activate()) return fail("device activation failed: " + be.error());
    if (step_counter != anchor) {  // direct steps since the last replay: re-anchor (the graph's literals count from the anchor)
      if (be.launch_u32(step_base, step_counter, 0, stream)) return fail("launch failed: " + be.error());
      anchor = step_counter;
    }
    step_counter += graph_n; anchor += graph_n;  // what the replay's last node does to the device word
    return 0;
  }

  int export_state(void* stream) {
    if (be.activate()) return fail("device activation failed: " + be.error());
    return be.launch_export(S, tables_dev, aos(), stream) ? fail("export launch failed: " + be.error()) : 0;
  }
  int commit_state(void* stream) {
    if (be.activate()) return fail("device activation failed: " + be.error());
    return be.launch_commit(S, tables_dev, aos(), stream) ? fail("commit launch failed: " + be.error()) : 0;
  }
  // host arrays -> export, overwrite the given parts, commit (include/rl_env.h rl_env_import_state)
  int import_state(const float* r, const float* q, const float* qd, void* stream) {
    if (export_state(stream)) return -1;
    if (r) be.h2d_stream(root_state, r, (size_t)N * 13 * sizeof(float), stream);
    if (q) be.h2d_stream(joint_pos, q, (size_t)N * D * sizeof(float), stream);
    if (qd) be.h2d_stream(joint_vel, qd, (size_t)N * D * sizeof(float), stream);
    return commit_state(stream);
  }

  void destroy() {
    (void)be.activate();
    for (void* p : allocs) be.free(p);
    allocs.clear();
  }
};

}  // namespace rl
