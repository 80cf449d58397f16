// env_step.h - the env-step lane program: ONE function that advances an environment by one
// `ManagerBasedRLEnv.step()` (stages 1-9), written for one lane of a 4-lane group.
//
// It is compiled twice from this single source:
//   * by hipcc for gfx950 (rl_env.hip): Ctx = wavefront context, group ops are DPP/ds_swizzle
//     shuffles, tables live in LDS;
//   * by g++ for the CPU lane emulator (tests/emu): Ctx = 4 host threads + a barrier.  That build is
//     test infrastructure for `-m "not gpu"` CI only and is never loaded by the product path.
//
// Physics (DESIGN.md "Simulator"): floating-base star articulation; composite-rigid-body inertia
// and RNEA bias in BASE coordinates; linearly-implicit contact / joint-limit / PD terms; the
// (6+CL) x (6+CL) per-lane system is reduced by a Schur complement onto the 6 base DoF, the four
// lanes' 6x6 contributions are summed with wavefront shuffles, every lane solves the 6x6 system
// redundantly and back-substitutes its own chain.  Same equations as oracle/physics.py, different
// formulation (that one is generic-tree, dense, fp64, link coordinates).
#pragma once
#include "env_tables.h"

namespace rl {

template <int N>
struct SymIdx {  // upper-triangular packed index of an N x N symmetric matrix
  static constexpr int size = N * (N + 1) / 2;
  static constexpr int at(int i, int j) { return i <= j ? i * N - i * (i - 1) / 2 + (j - i) : j * N - j * (j - 1) / 2 + (i - j); }
};

RL_FN float comp(V3 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; }
RL_FN V3 unit(int i) { return {i == 0 ? 1.f : 0.f, i == 1 ? 1.f : 0.f, i == 2 ? 1.f : 0.f}; }
RL_FN V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }

// bilinear heightfield: height + unit normal at world (x, y)     (oracle/physics.py TerrainSampler)
RL_FN void terrain_sample(const Tables& T, const float* __restrict__ hf, float x, float y, float& h, V3& n) {
  if (T.is_plane) {
    h = 0.f;
    n = {0.f, 0.f, 1.f};
    return;
  }
  float gx = (x - T.x0) / T.hscale, gy = (y - T.y0) / T.hscale;
  float fxf = floorf(gx), fyf = floorf(gy);
  int ix = (int)fminf(fmaxf(fxf, 0.f), (float)(T.nx - 2));
  int iy = (int)fminf(fmaxf(fyf, 0.f), (float)(T.ny - 2));
  float fx = clampf(gx - (float)ix, 0.f, 1.f), fy = clampf(gy - (float)iy, 0.f, 1.f);
  const float* b = hf + (size_t)ix * T.ny + iy;
  float h00 = b[0], h01 = b[1], h10 = b[T.ny], h11 = b[T.ny + 1];
  float hx0 = h00 + fx * (h10 - h00), hx1 = h01 + fx * (h11 - h01);
  h = hx0 + fy * (hx1 - hx0);
  float dzdx = ((1.f - fy) * (h10 - h00) + fy * (h11 - h01)) / T.hscale;
  float dzdy = ((1.f - fx) * (h01 - h00) + fx * (h11 - h10)) / T.hscale;
  float inv = 1.0f / sqrtf(dzdx * dzdx + dzdy * dzdy + 1.0f);
  n = {-dzdx * inv, -dzdy * inv, inv};
}

template <int CL>
struct Chain {  // kinematics of the lane's chain in base coordinates
  M3 R[CL];
  V3 p[CL], ax[CL];
};

template <int CL>
RL_FN void chain_kinematics(const LaneTab& L, const float (&q)[CL], Chain<CL>& C) {
  M3 Rp = identity3();
  V3 pp{0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < CL; ++j) {
    V3 al = ld3(L.axis[j]);
    C.p[j] = pp + mul(Rp, ld3(L.origin[j]));
    C.ax[j] = mul(Rp, al);
    C.R[j] = mul(Rp, rodrigues(al, q[j]));
    Rp = C.R[j];
    pp = C.p[j];
  }
}

// velocity (base coords) of the point x rigidly attached to link group g (0 = base), relative terms selectable
template <int CL>
RL_FN V3 point_velocity(const Chain<CL>& C, int g, V3 x, SV V0, const float (&qd)[CL]) {
  V3 u = V0.l + cross(V0.a, x);
#pragma unroll
  for (int i = 0; i < CL; ++i)
    if (i < g) u += qd[i] * cross(C.ax[i], x - C.p[i]);
  return u;
}

template <class Ctx, int CL>
struct EnvLane {
  static constexpr int NV = 6 + CL;
  using UI = SymIdx<NV>;
  static constexpr int NSPH = (CL + 1) * SPL;

  Ctx& ctx;
  const KState& S;
  const Tables& T;
  const LaneTab& L;
  int e, k, gl, NL, Np;
  // persistent state in registers
  V3 pos, vlin, vang;
  Q4 quat;
  float q[CL], qd[CL], kp[CL], kd[CL], act[CL], prev_act[CL];
  float Im[CL];
  V3 Icom_c[CL];
  S3 Icom_I[CL];
  SI I0;       // base link inertia (base coords)
  V3 base_com; // COM of the base body
  V3 extF, extT;
  float tim[NBS][4], fric[NBS][3];
  // per-step scratch
  float tau_app[CL], qacc[CL];
  V3 cf[NBS];
  float hist_n[NBS][3];  // |F| of the last three substeps, newest first

  RL_FN EnvLane(Ctx& c, const KState& s) : ctx(c), S(s), T(c.tables()), L(c.tables().lane[c.k()]) {
    e = ctx.env();
    k = ctx.k();
    Np = S.Npad;
    NL = NLANE * Np;
    gl = e * NLANE + k;
  }

  // ------------------------------------------------------------------ load / store
  RL_FN void load() {
    const float* r = S.root + e;
    pos = {r[0 * Np], r[1 * Np], r[2 * Np]};
    quat = {r[3 * Np], r[4 * Np], r[5 * Np], r[6 * Np]};
    vlin = {r[7 * Np], r[8 * Np], r[9 * Np]};
    vang = {r[10 * Np], r[11 * Np], r[12 * Np]};
    const float* w = S.wrench + e;
    extF = {w[0], w[Np], w[2 * Np]};
    extT = {w[3 * Np], w[4 * Np], w[5 * Np]};
    const float* bi = S.base_inertia + e;
    I0 = make_si(bi[0], V3{bi[Np], bi[2 * Np], bi[3 * Np]},
                 S3{bi[4 * Np], bi[5 * Np], bi[6 * Np], bi[7 * Np], bi[8 * Np], bi[9 * Np]});
    base_com = {S.base_com[e], S.base_com[Np + e], S.base_com[2 * Np + e]};
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      q[j] = S.q[j * NL + gl];
      qd[j] = S.qd[j * NL + gl];
      kp[j] = S.kp[j * NL + gl];
      kd[j] = S.kd[j * NL + gl];
      act[j] = S.act[j * NL + gl];
      const float* li = S.link_inertia + (size_t)j * INERTIA_NF * NL + gl;
      Im[j] = li[0];
      Icom_c[j] = {li[NL], li[2 * NL], li[3 * NL]};
      Icom_I[j] = {li[4 * NL], li[5 * NL], li[6 * NL], li[7 * NL], li[8 * NL], li[9 * NL]};
      tau_app[j] = 0.f;
      qacc[j] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < NBS; ++s) {
#pragma unroll
      for (int t = 0; t < 4; ++t) tim[s][t] = S.timers[(s * 4 + t) * NL + gl];
#pragma unroll
      for (int t = 0; t < 3; ++t) fric[s][t] = S.friction[(s * 3 + t) * NL + gl];
      cf[s] = {0.f, 0.f, 0.f};
      hist_n[s][0] = hist_n[s][1] = hist_n[s][2] = 0.f;
    }
  }

  RL_FN void store() {
    if (k == 0) {
      float* r = S.root + e;
      r[0 * Np] = pos.x; r[1 * Np] = pos.y; r[2 * Np] = pos.z;
      r[3 * Np] = quat.w; r[4 * Np] = quat.x; r[5 * Np] = quat.y; r[6 * Np] = quat.z;
      r[7 * Np] = vlin.x; r[8 * Np] = vlin.y; r[9 * Np] = vlin.z;
      r[10 * Np] = vang.x; r[11 * Np] = vang.y; r[12 * Np] = vang.z;
      float* w = S.wrench + e;
      w[0] = extF.x; w[Np] = extF.y; w[2 * Np] = extF.z;
      w[3 * Np] = extT.x; w[4 * Np] = extT.y; w[5 * Np] = extT.z;
    }
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      S.q[j * NL + gl] = q[j];
      S.qd[j * NL + gl] = qd[j];
      S.kp[j * NL + gl] = kp[j];
      S.kd[j * NL + gl] = kd[j];
      S.act[j * NL + gl] = act[j];
    }
#pragma unroll
    for (int s = 0; s < NBS; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) S.timers[(s * 4 + t) * NL + gl] = tim[s][t];
  }

  // ------------------------------------------------------------------ actuators [UPSTREAM B4]
  // returns explicit torque; fills tau_app (applied torque estimate) and the implicit-PD diagonal terms
  RL_FN void actuators(const float (&q_tgt)[CL], const float (&qd_tgt)[CL], float (&tau_e)[CL], float (&pd_diag)[CL], float (&pd_rhs)[CL]) {
    const float dt = T.dt;
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      float qt = L.action_is_vel[j] ? q[j] : q_tgt[j];
      float er = qt - q[j], ed = qd_tgt[j] - qd[j];
      float tc = kp[j] * er + kd[j] * ed;
      float eff = L.eff[j];
      if (L.act_implicit[j]) {
        float est = clampf(tc, -eff, eff);
        bool sat = fabsf(tc) > eff;
        tau_app[j] = est;
        tau_e[j] = sat ? est : 0.f;
        pd_diag[j] = sat ? 0.f : dt * (kd[j] + kp[j] * dt);
        pd_rhs[j] = sat ? 0.f : dt * (kp[j] * er + kd[j] * qd_tgt[j]);
      } else {  // DCMotor torque-speed clip (unitree.py:55-63)
        float tmax = clampf(L.sat[j] * (1.0f - qd[j] / L.act_vlim[j]), 0.f, eff);
        float tmin = clampf(L.sat[j] * (-1.0f - qd[j] / L.act_vlim[j]), -eff, 0.f);
        float t = clampf(tc, tmin, tmax);
        tau_app[j] = t;
        tau_e[j] = t;
        pd_diag[j] = 0.f;
        pd_rhs[j] = 0.f;
      }
    }
  }

  // ------------------------------------------------------------------ one physics substep
  RL_FN void substep(const float (&q_tgt)[CL], const float (&qd_tgt)[CL]) {
    const float dt = T.dt;
    float tau_e[CL], pd_diag[CL], pd_rhs[CL];
    actuators(q_tgt, qd_tgt, tau_e, pd_diag, pd_rhs);

    const M3 Rwb = quat_to_mat(quat);
    SV V0{mulT(Rwb, vang), mulT(Rwb, vlin)};
    SV a0{{0.f, 0.f, 0.f}, mulT(Rwb, V3{0.f, 0.f, T.gravity})};
    Chain<CL> C;
    chain_kinematics<CL>(L, q, C);

    float U[UI::size];
    float rv[NV];
#pragma unroll
    for (int i = 0; i < UI::size; ++i) U[i] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) rv[i] = 0.f;

    // ---- CRBA + RNEA in base coordinates
    SV Sj[CL];
    SI Ic[CL];
    SV Fs[CL], Hs[CL];
    {
      SV Vp = V0, ap = a0;
#pragma unroll
      for (int j = 0; j < CL; ++j) {
        Sj[j] = SV{C.ax[j], cross(C.p[j], C.ax[j])};
        SV vj = Sj[j] * qd[j];
        SV Vj = Vp + vj;
        SV aj = ap + crm(Vj, vj);
        V3 cb = C.p[j] + mul(C.R[j], Icom_c[j]);
        Ic[j] = make_si(Im[j], cb, rotate(C.R[j], Icom_I[j]));
        Hs[j] = apply(Ic[j], Vj);
        Fs[j] = apply(Ic[j], aj) + crf(Vj, Hs[j]);
        Vp = Vj;
        ap = aj;
      }
#pragma unroll
      for (int j = CL - 2; j >= 0; --j) {  // suffix sums: composite inertia / force / momentum
        Ic[j] = Ic[j] + Ic[j + 1];
        Fs[j] = Fs[j] + Fs[j + 1];
        Hs[j] = Hs[j] + Hs[j + 1];
      }
    }
    SI Itop = Ic[0];
    SV ftop = Fs[0], htop = Hs[0];
    if (k == 0) {  // the base link itself, its bias force and the persistent external wrench [UPSTREAM B8]
      SV h0 = apply(I0, V0);
      SV f0 = apply(I0, a0) + crf(V0, h0);
      f0.a -= extT + cross(base_com, extF);
      f0.l -= extF;
      Itop = Itop + I0;
      ftop = ftop + f0;
      htop = htop + h0;
    }
    {  // 6x6 block from the spatial inertia: [[I, hx],[hx^T, m 1]]
      U[UI::at(0, 0)] = Itop.I.xx; U[UI::at(1, 1)] = Itop.I.yy; U[UI::at(2, 2)] = Itop.I.zz;
      U[UI::at(0, 1)] = Itop.I.xy; U[UI::at(0, 2)] = Itop.I.xz; U[UI::at(1, 2)] = Itop.I.yz;
      U[UI::at(3, 3)] = Itop.m; U[UI::at(4, 4)] = Itop.m; U[UI::at(5, 5)] = Itop.m;
      U[UI::at(0, 4)] = -Itop.h.z; U[UI::at(0, 5)] = Itop.h.y;
      U[UI::at(1, 3)] = Itop.h.z;  U[UI::at(1, 5)] = -Itop.h.x;
      U[UI::at(2, 3)] = -Itop.h.y; U[UI::at(2, 4)] = Itop.h.x;
      rv[0] = htop.a.x - dt * ftop.a.x; rv[1] = htop.a.y - dt * ftop.a.y; rv[2] = htop.a.z - dt * ftop.a.z;
      rv[3] = htop.l.x - dt * ftop.l.x; rv[4] = htop.l.y - dt * ftop.l.y; rv[5] = htop.l.z - dt * ftop.l.z;
    }
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      SV B = apply(Ic[j], Sj[j]);
      U[UI::at(0, 6 + j)] = B.a.x; U[UI::at(1, 6 + j)] = B.a.y; U[UI::at(2, 6 + j)] = B.a.z;
      U[UI::at(3, 6 + j)] = B.l.x; U[UI::at(4, 6 + j)] = B.l.y; U[UI::at(5, 6 + j)] = B.l.z;
#pragma unroll
      for (int i = 0; i <= j; ++i) U[UI::at(6 + i, 6 + j)] = dot(Sj[i], B);
      float arm = L.armature[j];
      U[UI::at(6 + j, 6 + j)] += arm;
      rv[6 + j] = dot(Sj[j], Hs[j]) + arm * qd[j] + dt * (tau_e[j] - dot(Sj[j], Fs[j])) + pd_rhs[j];
      // joint limits: implicit spring-damper (hard limits in the reference, a1.urdf:369,411,439)
      float below = L.lower[j] - q[j], above = q[j] - L.upper[j];
      float viol = below > 0.f ? below : (above > 0.f ? -above : 0.f);
      bool lim = (below > 0.f) || (above > 0.f);
      U[UI::at(6 + j, 6 + j)] += pd_diag[j] + (lim ? dt * (T.limit_k * dt + T.limit_c) : 0.f);
      rv[6 + j] += dt * T.limit_k * viol;
    }

    // ---- contacts: collision spheres vs heightfield, linearly-implicit (oracle/physics.py header)
    V3 cx[NSPH], cn[NSPH];
    float cbias[NSPH], cdn[NSPH], cdt[NSPH];
    bool cact[NSPH];
#pragma unroll
    for (int g = 0; g <= CL; ++g) {
#pragma unroll
      for (int s = 0; s < SPL; ++s) {
        const int ci = g * SPL + s;
        cact[ci] = false;
        float rad = L.sph_r[g][s];
        if (rad > 0.f) {
          V3 cl = ld3(L.sph_c[g][s]);
          V3 cb = g == 0 ? cl : C.p[g > 0 ? g - 1 : 0] + mul(C.R[g > 0 ? g - 1 : 0], cl);
          V3 cw = pos + mul(Rwb, cb);
          float hz;
          V3 nw;
          terrain_sample(T, S.terrain, cw.x, cw.y, hz, nw);
          float phi = rad - (cw.z - hz) * nw.z;
          if (phi > 0.f) {
            V3 nb = mulT(Rwb, nw);
            V3 x = cb - rad * nb;
            V3 u = point_velocity<CL>(C, g, x, V0, qd);
            float un = dot(nb, u);
            V3 ut = u - un * nb;
            float utn = norm(ut);
            int slot = L.sph_slot[g][s];
            float mus = 0.f, mud = 0.f, rest = 0.f;
#pragma unroll
            for (int b = 0; b < NBS; ++b)
              if (b == slot) { mus = fric[b][0]; mud = fric[b][1]; rest = fric[b][2]; }
            float cnrm = T.contact_c * fminf(1.0f, phi / T.contact_phi_ref) * (1.0f - rest);
            float dn = cnrm + T.contact_k * dt;
            float bias = fminf(T.contact_k * phi, T.contact_vdep * dn);
            float fn0 = bias - dn * un;
            if (fn0 > 0.f) {
              float mu = utn < T.contact_vstick ? mus : mud;
              float dtan = fminf(T.contact_ct, mu * fn0 / fmaxf(utn, 1e-6f));
              cact[ci] = true;
              cx[ci] = x; cn[ci] = nb; cbias[ci] = bias; cdn[ci] = dn; cdt[ci] = dtan;
              // Jacobian columns of the point velocity wrt [omega_b, v_b, qd]
              V3 col[NV];
              col[0] = cross(unit(0), x); col[1] = cross(unit(1), x); col[2] = cross(unit(2), x);
              col[3] = unit(0); col[4] = unit(1); col[5] = unit(2);
#pragma unroll
              for (int i = 0; i < CL; ++i) col[6 + i] = i < g ? cross(C.ax[i], x - C.p[i]) : V3{0.f, 0.f, 0.f};
              float gn[NV];
#pragma unroll
              for (int i = 0; i < NV; ++i) gn[i] = dot(col[i], nb);
              const float kt = dt * dtan, kn = dt * (dn - dtan);
#pragma unroll
              for (int i = 0; i < NV; ++i) {
                rv[i] += dt * bias * gn[i];
#pragma unroll
                for (int jj = i; jj < NV; ++jj) U[UI::at(i, jj)] += kt * dot(col[i], col[jj]) + kn * gn[i] * gn[jj];
              }
            }
          }
        }
      }
    }

    // ---- Schur complement of the chain block, 4-lane reduction, 6x6 solve, back substitution
    float Lc[CL][CL];
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      float s = U[UI::at(6 + j, 6 + j)];
#pragma unroll
      for (int m = 0; m < j; ++m) s -= Lc[j][m] * Lc[j][m];
      float d = sqrtf(s);
      Lc[j][j] = d;
      float inv = 1.0f / d;
#pragma unroll
      for (int i = j + 1; i < CL; ++i) {
        float t = U[UI::at(6 + j, 6 + i)];
#pragma unroll
        for (int m = 0; m < j; ++m) t -= Lc[i][m] * Lc[j][m];
        Lc[i][j] = t * inv;
      }
    }
    float Y[6][CL], z[CL];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int j = 0; j < CL; ++j) {
        float t = U[UI::at(r, 6 + j)];
#pragma unroll
        for (int m = 0; m < j; ++m) t -= Lc[j][m] * Y[r][m];
        Y[r][j] = t / Lc[j][j];
      }
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      float t = rv[6 + j];
#pragma unroll
      for (int m = 0; m < j; ++m) t -= Lc[j][m] * z[m];
      z[j] = t / Lc[j][j];
    }
    using BI = SymIdx<6>;
    float Cb[BI::size], db[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float t = rv[r];
#pragma unroll
      for (int j = 0; j < CL; ++j) t -= Y[r][j] * z[j];
      db[r] = ctx.gsum(t);
#pragma unroll
      for (int c = r; c < 6; ++c) {
        float v = U[UI::at(r, c)];
#pragma unroll
        for (int j = 0; j < CL; ++j) v -= Y[r][j] * Y[c][j];
        Cb[BI::at(r, c)] = ctx.gsum(v);
      }
    }
    float nu0[6];
    {  // 6x6 Cholesky solve
      float G[6][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float s = Cb[BI::at(j, j)];
#pragma unroll
        for (int m = 0; m < j; ++m) s -= G[j][m] * G[j][m];
        float d = sqrtf(s);
        G[j][j] = d;
        float inv = 1.0f / d;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          float t = Cb[BI::at(j, i)];
#pragma unroll
          for (int m = 0; m < j; ++m) t -= G[i][m] * G[j][m];
          G[i][j] = t * inv;
        }
      }
      float y6[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float t = db[j];
#pragma unroll
        for (int m = 0; m < j; ++m) t -= G[j][m] * y6[m];
        y6[j] = t / G[j][j];
      }
#pragma unroll
      for (int j = 5; j >= 0; --j) {
        float t = y6[j];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) t -= G[i][j] * nu0[i];
        nu0[j] = t / G[j][j];
      }
    }
    float qdn[CL];
#pragma unroll
    for (int j = CL - 1; j >= 0; --j) {
      float t = z[j];
#pragma unroll
      for (int r = 0; r < 6; ++r) t -= Y[r][j] * nu0[r];
#pragma unroll
      for (int i = j + 1; i < CL; ++i) t -= Lc[i][j] * qdn[i];
      qdn[j] = t / Lc[j][j];
    }
#pragma unroll
    for (int j = 0; j < CL; ++j) qdn[j] = clampf(qdn[j], -L.vel_limit[j], L.vel_limit[j]);

    // ---- contact sensor: net contact force per body with the NEW velocities (world frame)
    SV V0n{{nu0[0], nu0[1], nu0[2]}, {nu0[3], nu0[4], nu0[5]}};
    V3 fslot[NBS];
#pragma unroll
    for (int b = 0; b < NBS; ++b) fslot[b] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g <= CL; ++g)
#pragma unroll
      for (int s = 0; s < SPL; ++s) {
        const int ci = g * SPL + s;
        if (cact[ci]) {
          V3 u = point_velocity<CL>(C, g, cx[ci], V0n, qdn);
          float un = dot(cn[ci], u);
          V3 Fb = (cbias[ci] - (cdn[ci] - cdt[ci]) * un) * cn[ci] - cdt[ci] * u;
          V3 Fw = mul(Rwb, Fb);
          int slot = L.sph_slot[g][s];
#pragma unroll
          for (int b = 0; b < NBS; ++b)
            if (b == slot) fslot[b] += Fw;
        }
      }
    // base-link bodies can be fed by several lanes (A1: the trunk box corners are spread over 4 lanes)
    for (int bi = 0; bi < T.n_base_bodies; ++bi) {
      bool mine = L.base_body_local == bi;
      V3 f{ctx.gsum(mine ? fslot[0].x : 0.f), ctx.gsum(mine ? fslot[0].y : 0.f), ctx.gsum(mine ? fslot[0].z : 0.f)};
      if (mine) fslot[0] = f;
    }
    // [UPSTREAM B5] ContactSensor: history roll + air/contact timers, every physics step
#pragma unroll
    for (int b = 0; b < NBS; ++b) {
      cf[b] = fslot[b];
      float fn = norm(fslot[b]);
      hist_n[b][2] = hist_n[b][1];
      hist_n[b][1] = hist_n[b][0];
      hist_n[b][0] = fn;
      bool contact = fn > T.force_threshold;
      float ca = tim[b][0], cc = tim[b][1];
      bool first_contact = (ca > 0.f) && contact, first_detach = (cc > 0.f) && !contact;
      tim[b][2] = first_contact ? ca + dt : tim[b][2];
      tim[b][0] = contact ? 0.f : ca + dt;
      tim[b][3] = first_detach ? cc + dt : tim[b][3];
      tim[b][1] = contact ? cc + dt : 0.f;
    }
    // ---- integrate (semi-implicit Euler: new velocities move the positions)
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      qacc[j] = (qdn[j] - qd[j]) / dt;
      q[j] += dt * qdn[j];
      qd[j] = qdn[j];
    }
    Q4 dq = quat_normalize(Q4{1.0f, 0.5f * dt * nu0[0], 0.5f * dt * nu0[1], 0.5f * dt * nu0[2]});
    quat = quat_normalize(quat_mul(quat, dq));
    M3 Rn = quat_to_mat(quat);
    vang = mul(Rn, V0n.a);
    vlin = mul(Rn, V0n.l);
    pos = pos + dt * vlin;
  }
};

}  // namespace rl
