// env_step.h - the env-step lane program: ONE function that advances an environment by one
// `ManagerBasedRLEnv.step()` (stages 1-9), written for one lane of the 4 x SUB lanes that
// simulate an environment (SUB sub-lanes per limb: 4 on the GPU by default, 1 in the older mapping).
//
// It is compiled twice from this single source:
//   * by hipcc for gfx950 (rl_env.hip): Ctx = wavefront context, group ops are DPP/ds_swizzle
//     shuffles, tables live in LDS;
//   * by g++ for the CPU lane emulator (tests/emu): Ctx = 4 x SUB host threads + a barrier.  That build is
//     test infrastructure for `-m "not gpu"` CI only and is never loaded by the product path.
//
// Physics (DESIGN.md "Simulator"): floating-base articulation = a trunk (base + NW serial joints, G1: the waist) carrying 4
// limb chains; link inertias, velocities and bias forces in BASE coordinates; linearly-implicit contact / joint-limit / PD terms.
// The velocity-level system (H + A) nu+ = rhs is solved by an articulated-body recursion: every limb is eliminated joint by joint
// from the tip (6 x 6 symmetric link records, a contact is a 6 x 6 block on its link), the limbs' articulated inertias meet at the
// base (DPP sums / ds_add_f32), every lane solves the 6 x 6 base system redundantly and walks its own chain outwards.  Same
// equations as oracle/physics.py, different formulation (that one is generic-tree, dense, fp64, link coordinates).
#pragma once
#include <type_traits>

#include "env_spec.h"
#include "env_tables.h"

// RL_PHASE(id, "name") marks a phase boundary of the lane program.  Product builds: nothing.  Analysis builds of the device code:
//   -DRL_PHASE_MARKS  plants a `; PHASE name` comment in the assembly: tools/isa_profile.py attributes the static instruction mix
//                     to the phases (an asm volatile statement is also a scheduling barrier, so such a build is for reading only);
//   -DRL_PHASE_CLOCK  lane 0 of every wavefront accumulates the shader-clock ticks it spends in each phase into float row
//                     [wavefront][id] behind the reward-term rows (tools/phase_clock.py reads them): s_waitcnt 0, read the clock,
//                     add the interval to the phase that ends, read the clock again - the bookkeeping itself is not counted.
// The ids index tools/phase_clock.py PHASES.
// RL_PK: the joint elimination and the contact blocks on packed fp32 pairs (rl_math.h F2p; eliminate_pk below).  On by default - one call,
// specialised kernels: A1 33.95 -> 33.61 us, G1 86.19 -> 84.65 (profiles/r06m_a1_pk_ab.txt, r06m_g1_pk_ab.txt); -DRL_NO_PK: the scalar form.
#if !defined(RL_NO_PK) && !defined(RL_PK)
#define RL_PK 1
#endif
#if defined(RL_PHASE_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define RL_PHASE(id, name) asm volatile("; PHASE " name)
#elif defined(RL_PHASE_CLOCK) && defined(__HIP_DEVICE_COMPILE__)
#define RL_PHASE(id, name) this->phase_stamp(id)
#define RL_PHASE_START() (this->ph_t0 = 0, this->ph_cur = 0)  // (explicitly: round 4's tables carried a raw time stamp in the first row)
#define RL_PHASE_CLOCK_ON 1
#else
#define RL_PHASE(id, name) ((void)0)
#endif
#ifndef RL_PHASE_START
#define RL_PHASE_START() ((void)0)
#endif
constexpr int RL_PHASE_ROW0 = 24;   // first reward-term row the clock build borrows (A1 .. G1 tasks have <= 22 reward terms)
constexpr int RL_PHASE_SLOTS = 32;

namespace rl {

template <int N>
struct SymIdx {  // upper-triangular packed index of an N x N symmetric matrix
  static constexpr int size = N * (N + 1) / 2;
  static constexpr int at(int i, int j) { return i <= j ? i * N - i * (i - 1) / 2 + (j - i) : j * N - j * (j - 1) / 2 + (i - j); }
};

// compile-time loops whose index is needed as a constant expression (DPP controls, array slots that must stay registers)
template <int I, int N, class F>
RL_FN void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
template <int I, class F>
RL_FN void static_for_down(F&& f) {  // I, I - 1, ..., 0
  if constexpr (I >= 0) {
    f(std::integral_constant<int, I>{});
    static_for_down<I - 1>(f);
  }
}

RL_FN float comp(V3 v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; }
RL_FN V3 unit(int i) { return {i == 0 ? 1.f : 0.f, i == 1 ? 1.f : 0.f, i == 2 ? 1.f : 0.f}; }
RL_FN V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }

// Wave-uniform hot scalars.  On the GPU they are pinned into SGPRs with readfirstlane once per kernel:
// read as LDS table entries each one cost a ~64-cycle ds_read round trip at every use, which a lone
// wavefront per SIMD cannot hide (profiles/r01: 62 % of wave time was s_waitcnt).
struct Uni {
  float dt, inv_dt, gravity, contact_k, contact_c, inv_phi_ref, contact_ct, contact_vdep, contact_vstick, limit_k, limit_c, force_threshold;
  float inv_hscale, x0, y0;
  int is_plane, nx, ny;
  double x0d, y0d, inv_hd;  // the grid transform of the heightfield in fp64 (terrain_fetch)
  uint32_t trunk_restart;   // bit i: trunk joint i hangs off the BASE, not off trunk joint i - 1 (TaskTab::trunk_restart)
};
template <class Ctx>
RL_FN Uni make_uni(const Ctx& ctx, const TaskTab& T) {
  Uni u;
  u.dt = ctx.uniform(T.dt); u.inv_dt = ctx.uniform(1.0f / T.dt); u.gravity = ctx.uniform(T.gravity);
  u.contact_k = ctx.uniform(T.contact_k); u.contact_c = ctx.uniform(T.contact_c); u.inv_phi_ref = ctx.uniform(1.0f / T.contact_phi_ref);
  u.contact_ct = ctx.uniform(T.contact_ct); u.contact_vdep = ctx.uniform(T.contact_vdep); u.contact_vstick = ctx.uniform(T.contact_vstick);
  u.limit_k = ctx.uniform(T.limit_k); u.limit_c = ctx.uniform(T.limit_c); u.force_threshold = ctx.uniform(T.force_threshold);
  u.inv_hscale = ctx.uniform(T.is_plane ? 1.0f : 1.0f / T.hscale); u.x0 = ctx.uniform(T.x0); u.y0 = ctx.uniform(T.y0);
  u.is_plane = ctx.uniform_i(T.is_plane); u.nx = ctx.uniform_i(T.nx); u.ny = ctx.uniform_i(T.ny);
  auto uni_d = [&](double v) {  // a wave-uniform double: both halves pinned
    union { double d; int i[2]; } w;
    w.d = v;
    w.i[0] = ctx.uniform_i(w.i[0]); w.i[1] = ctx.uniform_i(w.i[1]);
    return w.d;
  };
  u.x0d = uni_d((double)T.x0); u.y0d = uni_d((double)T.y0);
  u.inv_hd = uni_d(T.is_plane ? 1.0 : 1.0 / (double)T.hscale);
  u.trunk_restart = (uint32_t)ctx.uniform_i((int)T.trunk_restart);
  return u;
}

// bilinear heightfield (oracle/physics.py TerrainSampler), split so that the 4 corner loads of several
// query points can be in flight together before any of them is consumed
struct F2 {
  float x, y;
};
RL_FN F2 ld2(const float* p) {  // 4-byte aligned 8-byte load
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f2v __attribute__((ext_vector_type(2), aligned(4)));
  f2v v = *reinterpret_cast<const f2v*>(p);
  return {v.x, v.y};
#else
  return {p[0], p[1]};
#endif
}
// The heightfield in HBM: the caller's row-major grid.  -DRL_TERRAIN_PAIRS (analysis switch; rl_env_host.h builds the array): rows ix and
// ix + 1 interleaved, hf2[(ix * ny + iy) * 2 + {0, 1}] = {h(ix, iy), h(ix + 1, iy)}, so that the four corners of cell (ix, iy) are FOUR
// CONSECUTIVE WORDS - one 16-byte load and 1.25 cache lines per query instead of two 8-byte loads in two rows 16 KB apart (2.1 lines), at
// twice the grid's bytes.  Measured in one call and NOT kept: A1 32.57 / 32.62 us, Go2W 38.82 / 38.62, G1 82.27 / 82.28
// (profiles/r06s_*_hf2_ab.txt) - the lookups are batched loads whose latency is covered either way; their line count is not what the step waits for.
#ifdef RL_TERRAIN_PAIRS
constexpr bool TERRAIN_PAIRS = true;
#else
constexpr bool TERRAIN_PAIRS = false;
#endif
struct F4h {
  float x, y, z, w;
};
RL_FN F4h ld4_a8(const float* p) {  // 8-byte aligned 16-byte load
#if defined(__HIP_DEVICE_COMPILE__)
  typedef float f4v __attribute__((ext_vector_type(4), aligned(8)));
  f4v v = *reinterpret_cast<const f4v*>(p);
  return {v.x, v.y, v.z, v.w};
#else
  return {p[0], p[1], p[2], p[3]};
#endif
}
struct TerrainPatch {
  float h00, h01, h10, h11, fx, fy;
};
// The query point is (bx + dx, by + dy): a base that is a WORLD coordinate (root position: tens of metres on the 80 m x 40 m
// terrain, fp32 ulp 4e-6 m) plus a small offset (sphere centre / scan ray relative to the root).  The BASE is turned into grid
// coordinates in fp64, once per environment and (sub)step: integer cell + fraction (terrain_base); a query then adds its offset to
// the fraction in fp32 - numbers of at most a few dozen cells, so the cell index and the in-cell fractions carry no
// world-coordinate round-off (error <= 2e-6 cells = 1e-7 m; summed in fp32 at world scale the sum alone moved a sphere by up to
// 4e-6 m = 0.08 N of contact force at k = 2e4 N/m, which is what sized the switch margins of the teacher-forced parity tests,
// tests/helpers.py SWITCH_EPS).  Round 2 did the whole transform in fp64 per QUERY: ~25 fp64 / conversion instructions each, 24
// queries per lane and step on A1 (profiles/r03_a1_*).
struct TerrainBase {
  int cx, cy;            // cell of the base point (unclamped)
  float fx, fy;          // its position inside that cell, [0, 1)
  float lox, hix, loy, hiy;  // the grid's first / last cell relative to (cx, cy): a query's cell offset is clamped to them
};
RL_FN TerrainBase terrain_base(const Uni& u, float bx, float by) {
  TerrainBase b;
  if (u.is_plane) {
    b.cx = b.cy = 0;
    b.fx = b.fy = b.lox = b.hix = b.loy = b.hiy = 0.f;
    return b;
  }
  const double gx = ((double)bx - u.x0d) * u.inv_hd, gy = ((double)by - u.y0d) * u.inv_hd;
  const double cx = floor(gx), cy = floor(gy);
  b.cx = (int)cx; b.cy = (int)cy;
  b.fx = (float)(gx - cx); b.fy = (float)(gy - cy);
  b.lox = (float)(-b.cx); b.hix = (float)(u.nx - 2 - b.cx);
  b.loy = (float)(-b.cy); b.hiy = (float)(u.ny - 2 - b.cy);
  return b;
}
RL_FN TerrainPatch terrain_fetch(const Uni& u, const float* __restrict__ hf, const TerrainBase& tb, float dx, float dy) {
  TerrainPatch p;
  if (u.is_plane) {
    p.h00 = p.h01 = p.h10 = p.h11 = 0.f;
    p.fx = p.fy = 0.f;
    return p;
  }
  // grid coordinates relative to the base cell; cell = clamp(floor(g), first, last), fraction = clamp(g - cell, 0, 1): outside the
  // grid the border cell with the point pushed onto its edge (the same rule as oracle/physics.py TerrainSampler)
  const float gx = tb.fx + dx * u.inv_hscale, gy = tb.fy + dy * u.inv_hscale;
  const float ox = clampf(floorf(gx), tb.lox, tb.hix), oy = clampf(floorf(gy), tb.loy, tb.hiy);
  p.fx = clampf(gx - ox, 0.f, 1.f);
  p.fy = clampf(gy - oy, 0.f, 1.f);
  // the integer clamp is the memory-safety net: a non-finite or astronomically large root position (a diverged env) makes the float
  // arithmetic above meaningless - (int)NaN is 0, (float)(-cx) rounds - and an unclamped cell index then reads far outside the
  // heightfield (a GPU memory fault took the whole launch down on a model with a zero velocity limit, profiles/r03p_all_tasks.txt)
  const int ix = imin(imax(tb.cx + (int)ox, 0), u.nx - 2), iy = imin(imax(tb.cy + (int)oy, 0), u.ny - 2);
  if constexpr (TERRAIN_PAIRS) {  // {h(ix, iy), h(ix + 1, iy), h(ix, iy + 1), h(ix + 1, iy + 1)}: one load
    const F4h r = ld4_a8(hf + (((uint32_t)ix * (uint32_t)u.ny + (uint32_t)iy) << 1));
    p.h00 = r.x; p.h10 = r.y; p.h01 = r.z; p.h11 = r.w;
  } else {  // (iy, iy+1) are adjacent in memory: two 8-byte loads per query instead of four 4-byte ones
    const float* b = hf + (uint32_t)ix * (uint32_t)u.ny + (uint32_t)iy;
    F2 r0 = ld2(b), r1 = ld2(b + u.ny);
    p.h00 = r0.x; p.h01 = r0.y; p.h10 = r1.x; p.h11 = r1.y;
  }
  return p;
}
RL_FN TerrainPatch terrain_fetch(const Uni& u, const float* __restrict__ hf, float bx, float by, float dx, float dy) {
  return terrain_fetch(u, hf, terrain_base(u, bx, by), dx, dy);
}
RL_FN void terrain_eval(const Uni& u, const TerrainPatch& p, float& h, V3& n) {
  float hx0 = p.h00 + p.fx * (p.h10 - p.h00), hx1 = p.h01 + p.fx * (p.h11 - p.h01);
  h = hx0 + p.fy * (hx1 - hx0);
  float dzdx = ((1.f - p.fy) * (p.h10 - p.h00) + p.fy * (p.h11 - p.h01)) * u.inv_hscale;
  float dzdy = ((1.f - p.fx) * (p.h01 - p.h00) + p.fx * (p.h11 - p.h10)) * u.inv_hscale;
  float inv = frsqrt(dzdx * dzdx + dzdy * dzdy + 1.0f);
  n = {-dzdx * inv, -dzdy * inv, inv};
}
RL_FN float terrain_height(const TerrainPatch& p) {  // the height alone (ray casters)
  const float hx0 = p.h00 + p.fx * (p.h10 - p.h00), hx1 = p.h01 + p.fx * (p.h11 - p.h01);
  return hx0 + p.fy * (hx1 - hx0);
}
RL_FN void terrain_sample(const Uni& u, const float* __restrict__ hf, float bx, float by, float dx, float dy, float& h, V3& n) {
  terrain_eval(u, terrain_fetch(u, hf, bx, by, dx, dy), h, n);
}

RL_FN M3 ld_m3(const float* p) { return M3{{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}}; }

// Four consecutive, 16-byte aligned words of LDS (the limb-shared records of the trunk + limbs instance): ONE ds_read_b128 /
// ds_write_b128 on the GPU instead of four ds_read_b32 / ds_write_b32 - a quarter of the LDS instructions of a kernel whose four
// wavefronts per CU queue on the one LDS pipe (4.2 k LDS instructions per wavefront and step before: profiles/r04a_g1_pmc_sq.txt).
struct F4 {
  float x, y, z, w;
};
RL_FN F4 ld4(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float4 v = *reinterpret_cast<const float4*>(p);
  return {v.x, v.y, v.z, v.w};
#else
  return {p[0], p[1], p[2], p[3]};
#endif
}
RL_FN void st4(float* p, F4 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  *reinterpret_cast<float4*>(p) = make_float4(v.x, v.y, v.z, v.w);
#else
  p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
#endif
}

// a joint's packed constants (LaneTabT::jc): origin and axis in the parent link frame
template <class LT>
RL_FN void joint_origin_axis(const LT& L, int j, V3& origin, V3& axis) {
  const F4 c0 = ld4(L.jc[j]), c1 = ld4(L.jc[j] + 4);
  origin = {c0.x, c0.y, c0.z};
  axis = {c0.w, c1.x, c1.y};
}

// Kinematics in base coordinates: the lane's limb (R, p, ax per joint) and the trunk joints (Rw, pw, axw).
// Two storage modes behind one accessor interface:
//   * registers (quadrupeds: 3-4 joints, everything stays in VGPRs);
//   * LDS words shared by the sub-lanes of a limb (G1: 7 + 3 joints = 150 words that the register allocator
//     otherwise spills to scratch around the 16 x 16 system; the sub-lanes hold identical values, so they
//     read / write the same word - a broadcast, no bank conflict: word w of limb l sits at b[w * STRIDE]).
template <class TP, bool LDS, int STRIDE>
struct ChainT;

template <class TP, int STRIDE>
struct ChainT<TP, false, STRIDE> {
  static constexpr int CL = TP::CL, NW = TP::NW, NWA = TP::NW > 0 ? TP::NW : 1;
  M3 R_[CL];
  V3 p_[CL], ax_[CL];
  M3 Rw_[NWA];
  V3 pw_[NWA], axw_[NWA];
  RL_FN explicit ChainT(float*) {}
  RL_FN M3 R(int j) const { return R_[j]; }
  RL_FN V3 p(int j) const { return p_[j]; }
  RL_FN V3 ax(int j) const { return ax_[j]; }
  RL_FN M3 Rw(int i) const { return Rw_[i]; }
  RL_FN V3 pw(int i) const { return pw_[i]; }
  RL_FN V3 axw(int i) const { return axw_[i]; }
  RL_FN void set(int j, const M3& R, V3 p, V3 ax) { R_[j] = R; p_[j] = p; ax_[j] = ax; }
  RL_FN void setw(int i, const M3& R, V3 p, V3 ax) { Rw_[i] = R; pw_[i] = p; axw_[i] = ax; }
};

template <class TP, int STRIDE>
struct ChainT<TP, true, STRIDE> {
  // LIMB-MAJOR: the limb's block is contiguous, a joint is 16 words - [ax.x ax.y ax.z p.x | p.y p.z R00 R01 | R02 R10 R11 R12 | R20 R21 R22 -]
  // - so that (axis, origin), what the velocity pass, the elimination and the outward pass read, is two 16-byte vectors and a whole
  // joint four.  (Until round 4 word w of limb l sat at b[w * limbs + l]: 15 ds_write_b32 + 6 .. 12 ds_read_b32 per joint and use.)
  static constexpr int CL = TP::CL, NW = TP::NW, JW = 16;
  float* b;
  RL_FN explicit ChainT(float* base) : b(base) {}
  RL_FN void axp(int slot, V3& ax, V3& p) const {
    const F4 a = ld4(b + slot * JW), c = ld4(b + slot * JW + 4);
    ax = {a.x, a.y, a.z};
    p = {a.w, c.x, c.y};
  }
  RL_FN void frame(int slot, M3& R, V3& p) const {
    const F4 a = ld4(b + slot * JW), c = ld4(b + slot * JW + 4), d = ld4(b + slot * JW + 8), e = ld4(b + slot * JW + 12);
    R = M3{{c.z, c.w, d.x}, {d.y, d.z, d.w}, {e.x, e.y, e.z}};
    p = {a.w, c.x, c.y};
  }
  RL_FN void put(int slot, const M3& R, V3 p, V3 ax) {
    st4(b + slot * JW, F4{ax.x, ax.y, ax.z, p.x});
    st4(b + slot * JW + 4, F4{p.y, p.z, R.r0.x, R.r0.y});
    st4(b + slot * JW + 8, F4{R.r0.z, R.r1.x, R.r1.y, R.r1.z});
    st4(b + slot * JW + 12, F4{R.r2.x, R.r2.y, R.r2.z, 0.f});
  }
  RL_FN M3 R(int j) const { M3 r; V3 q; frame(j, r, q); return r; }
  RL_FN V3 p(int j) const { V3 a, q; axp(j, a, q); return q; }
  RL_FN V3 ax(int j) const { V3 a, q; axp(j, a, q); return a; }
  RL_FN M3 Rw(int i) const { return R(CL + i); }
  RL_FN V3 pw(int i) const { return p(CL + i); }
  RL_FN V3 axw(int i) const { return ax(CL + i); }
  RL_FN void set(int j, const M3& R, V3 p, V3 ax) { put(j, R, p, ax); }
  RL_FN void setw(int i, const M3& R, V3 p, V3 ax) { put(CL + i, R, p, ax); }
};

// Closest points of the segments a0-a1 and b0-b1 (Ericson, Real-Time Collision Detection 5.1.9), branch-free: the same case order
// as oracle/physics.py `segment_closest`, so that both sides pick the same points where a segment degenerates to a point
// or the two are parallel.
RL_FN void segment_closest(V3 a0, V3 a1, V3 b0, V3 b1, V3& xa, V3& xb) {
  const V3 d1 = a1 - a0, d2 = b1 - b0, r = a0 - b0;
  const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), c = dot(d1, r), b = dot(d1, d2);
  const float eps = 1e-12f;
  const float den = a * e - b * b;
  const float ia = a > eps ? frcp(a) : 0.f, ie = e > eps ? frcp(e) : 0.f;
  float s = den > eps ? clampf((b * f - c * e) * frcp(den), 0.f, 1.f) : 0.f;
  s = a > eps ? s : 0.f;
  float t = (b * s + f) * ie;
  const float s_lo = clampf(-c * ia, 0.f, 1.f), s_hi = clampf((b - c) * ia, 0.f, 1.f);
  s = e > eps ? (t < 0.f ? s_lo : (t > 1.f ? s_hi : s)) : s_lo;
  t = clampf(t, 0.f, 1.f);
  xa = a0 + s * d1;
  xb = b0 + t * d2;
}

// frame of the trunk link reached after `depth` trunk joints (0 = the base itself)
template <class TP, class CT>
RL_FN void trunk_frame(const CT& C, int depth, M3& Rf, V3& pf) {
  Rf = identity3();
  pf = {0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < TP::NW; ++i)
    if (depth == i + 1) { Rf = C.Rw(i); pf = C.pw(i); }
}

// The same frame straight from the joint positions - the trunk loop of chain_kinematics() below alone, nothing stored: what the scanner
// pose of a freshly reset env needs of the new posture (the full kinematics through the limb-shared chain words was 5.5 k ticks = 2.9 us
// of a resetting G1 wavefront, profiles/r05g_phase_clock_clockspec_78_reset_env0.txt)
template <class TP, class LT>
RL_FN void trunk_frame_of_pose(const LT& L, const float (&q)[TP::JX], uint32_t restart, int depth, M3& Rf, V3& pf);

// `on_trunk(i, axis, origin)` / `on_limb(j, axis, origin)`: called with the joint's axis and origin (base coordinates) while they are in
// registers - the substep of the trunk + limbs instance builds the link velocities / bias accelerations there instead of reading
// the chain words back (two LDS round trips per joint less); NoJoint = nothing to do
struct NoJoint {
  RL_FN void operator()(int, V3, V3) const {}
};
// `restart`: bit i = trunk joint i hangs off the base (the trunk is one serial spine, or - Booster T1: a waist and a neck on the base -
// several serial pieces that each start at the base; a piece is a run of consecutive trunk joints)
// Rp * Rot(sg e_KIND, ang): a rotation about a basis vector of the joint frame touches two columns of Rp (env_spec.h spec_axis_kind)
template <int KIND>
RL_FN M3 mul_axis_rot(const M3& Rp, float sg, float ang) {
  float s, c;
  fsincos(ang, s, c);
  s *= sg;
  auto row = [&](V3 r) -> V3 {
    if constexpr (KIND == 0) return {r.x, c * r.y + s * r.z, c * r.z - s * r.y};
    else if constexpr (KIND == 1) return {c * r.x - s * r.z, r.y, s * r.x + c * r.z};
    else return {c * r.x + s * r.y, c * r.y - s * r.x, r.z};
  };
  return {row(Rp.r0), row(Rp.r1), row(Rp.r2)};
}
template <class TP, class SP = NoSpec, class CT, class FT = NoJoint, class FL = NoJoint>
RL_FN void chain_kinematics(const LaneTabT<TP>& L, const float (&q)[TP::JX], CT& C, uint32_t restart = 0u, FT&& on_trunk = FT{}, FL&& on_limb = FL{}) {
  constexpr int CL = TP::CL, NW = TP::NW;
  M3 Rp = identity3(), Ra = identity3();
  V3 pp{0.f, 0.f, 0.f}, pa{0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NW; ++i) {  // trunk joints (same in every lane)
    const int jx = CL + i;
    if (i > 0 && ((restart >> i) & 1u)) { Rp = identity3(); pp = {0.f, 0.f, 0.f}; }
    V3 al, oj;
    joint_origin_axis(L, jx, oj, al);
    M3 Rj0 = mul(Rp, ld_m3(L.rot0[TP::ROT ? jx : 0]));
    V3 pj = pp + mul(Rp, oj);
    M3 Rj = mul(Rj0, rodrigues(al, q[jx]));
    const V3 aw = mul(Rj0, al);
    C.setw(i, Rj, pj, aw);
    on_trunk(i, aw, pj);
    Rp = Rj;
    pp = pj;
    if (L.attach == i + 1) { Ra = Rj; pa = pj; }
  }
  Rp = Ra;
  pp = pa;
  static_for<0, CL>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    constexpr int KIND = spec_axis_kind<SP>(j);
    V3 al, oj;
    joint_origin_axis(L, j, oj, al);
    V3 pj = pp + mul(Rp, oj);
    if (TP::ROT) Rp = mul(Rp, ld_m3(L.rot0[TP::ROT ? j : 0]));
    M3 Rj;
    V3 aw;
    if constexpr (KIND == 3) {
      Rj = mul(Rp, rodrigues(al, q[j]));
      aw = mul(Rp, al);
    } else {  // the joint turns about +- a basis vector of its frame in every limb (the sign: this limb's table)
      const float sg = KIND == 0 ? al.x : (KIND == 1 ? al.y : al.z);
      Rj = mul_axis_rot<KIND>(Rp, sg, q[j]);
      aw = sg * (KIND == 0 ? col0(Rp) : (KIND == 1 ? col1(Rp) : col2(Rp)));
    }
    C.set(j, Rj, pj, aw);
    on_limb(j, aw, pj);
    Rp = Rj;
    pp = pj;
  });
}

template <class TP, class LT>
RL_FN void trunk_frame_of_pose(const LT& L, const float (&q)[TP::JX], uint32_t restart, int depth, M3& Rf, V3& pf) {
  constexpr int CL = TP::CL, NW = TP::NW;
  Rf = identity3();
  pf = {0.f, 0.f, 0.f};
  M3 Rp = identity3();
  V3 pp{0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int jx = CL + i;
    if (i > 0 && ((restart >> i) & 1u)) { Rp = identity3(); pp = {0.f, 0.f, 0.f}; }
    V3 al, oj;
    joint_origin_axis(L, jx, oj, al);
    const M3 Rj0 = mul(Rp, ld_m3(L.rot0[TP::ROT ? jx : 0]));
    const V3 pj = pp + mul(Rp, oj);
    const M3 Rj = mul(Rj0, rodrigues(al, q[jx]));
    Rp = Rj;
    pp = pj;
    if (depth == i + 1) { Rf = Rj; pf = pj; }
  }
}

// The same kinematics on the trunk + limbs instance with several sub-lanes per limb.  The local joint transforms
// T_j = rot0_j * Rodrigues(axis_j, q_j) (sin / cos and a 3 x 3 product per joint) come out the same in every sub-lane of a limb, so
// they are DEALT: joint jx is computed by sub-lane jx % SUB only, and when the chain product reaches joint jx every sub-lane takes
// T_jx from its owner with nine quad broadcasts.  What stays per lane and joint is R_j = R_parent * T_j, the joint origin and the
// world axis (axis_w = R_j * axis: Rodrigues(axis, .) leaves its own axis where it is, so this equals rot-frame * axis).  About
// 60 instead of 110 instructions per joint of the 10-joint chain, five times per step.
// (SUB here is the DEALING width: the sub-lanes of a DPP quad - with eight sub-lanes per limb each of the limb's two quads deals among
// its own four lanes, `sub` = the lane's index in its quad, and the broadcasts stay single quad_perm moves: Ctx::deal_bcast_m3.)
template <class TP, int SUB, class Ctx, class CT, class FT = NoJoint, class FL = NoJoint>
RL_FN void chain_kinematics_dealt(Ctx& ctx, int sub, const LaneTabT<TP>& L, const float (&q)[TP::JX], CT& C, uint32_t restart = 0u, FT&& on_trunk = FT{}, FL&& on_limb = FL{}) {
  static_assert(TP::ROT && SUB > 1 && SUB <= 4, "trunk + limbs instance, several sub-lanes per limb");
  constexpr int CL = TP::CL, NW = TP::NW, JX = TP::JX, NS = (JX + SUB - 1) / SUB;
  M3 Tl[NS];
  static_for<0, NS>([&](auto ic) __attribute__((always_inline)) {  // this sub-lane's joint of round i: SUB * i + sub (a partial last round is clamped, unused)
    constexpr int i = decltype(ic)::value;
    float qi = opaque(q[SUB * i]);  // (opaque: a select over plain loads of q becomes one load through a selected address - rl_math.h)
    static_for<1, SUB>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s2 = decltype(sc)::value, jq = SUB * i + s2 < JX ? SUB * i + s2 : JX - 1;
      const float cand = opaque(q[jq]);
      qi = sub == s2 ? cand : qi;
    });
    const int jx = imin(SUB * i + sub, JX - 1);
    const F4 r0 = ld4(L.rota[jx]), r1 = ld4(L.rota[jx] + 4), r2 = ld4(L.rota[jx] + 8);  // rot0 (row-major) and the axis: three vectors
    Tl[i] = mul(M3{{r0.x, r0.y, r0.z}, {r0.w, r1.x, r1.y}, {r1.z, r1.w, r2.x}}, rodrigues(V3{r2.y, r2.z, r2.w}, qi));
  });
  M3 Rp = identity3(), Ra = identity3();
  V3 pp{0.f, 0.f, 0.f}, pa{0.f, 0.f, 0.f};
  // (the bodies must be inlined at all three call sites: a lambda left as a function takes its captures - the lane object - by address)
  static_for<0, NW>([&](auto ic) __attribute__((always_inline)) {  // trunk joints (same in every lane)
    constexpr int i = decltype(ic)::value, jx = CL + i;
    if (i > 0 && ((restart >> i) & 1u)) { Rp = identity3(); pp = {0.f, 0.f, 0.f}; }
    const M3 Tj = ctx.template deal_bcast_m3<jx % SUB>(Tl[jx / SUB]);
    V3 oj, al;
    joint_origin_axis(L, jx, oj, al);
    const V3 pj = pp + mul(Rp, oj);
    const M3 Rj = mul(Rp, Tj);
    const V3 aw = mul(Rj, al);
    C.setw(i, Rj, pj, aw);
    on_trunk(i, aw, pj);
    Rp = Rj;
    pp = pj;
    if (L.attach == i + 1) { Ra = Rj; pa = pj; }
  });
  Rp = Ra;
  pp = pa;
  static_for<0, CL>([&](auto jc) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value;
    const M3 Tj = ctx.template deal_bcast_m3<j % SUB>(Tl[j / SUB]);
    V3 oj, al;
    joint_origin_axis(L, j, oj, al);
    const V3 pj = pp + mul(Rp, oj);
    const M3 Rj = mul(Rp, Tj);
    const V3 aw = mul(Rj, al);
    C.set(j, Rj, pj, aw);
    on_limb(j, aw, pj);
    Rp = Rj;
    pp = pj;
  });
}

// velocity (base coords) of the point x rigidly attached to a link that is moved by the trunk joints of `anc` (bit i: trunk joint i -
// TaskTab::trunk_anc[depth]: the joints between the base and that trunk link) and the first `lg` limb joints
template <class TP, class CT>
RL_FN V3 point_velocity(const CT& C, uint32_t anc, int lg, V3 x, SV V0, const float (&qd)[TP::JX]) {
  V3 u = V0.l + cross(V0.a, x);
#pragma unroll
  for (int i = 0; i < TP::NW; ++i)
    if ((anc >> i) & 1u) u += qd[TP::CL + i] * cross(C.axw(i), x - C.pw(i));
#pragma unroll
  for (int i = 0; i < TP::CL; ++i)
    if (i < lg) u += qd[i] * cross(C.ax(i), x - C.p(i));
  return u;
}

// Lane-private LDS scratchpad in 16-byte GRANULES: granule g of a lane lives at base[(g * STRIDE + lane) * 4] (STRIDE = 64 on the GPU:
// a ds_read_b128 / ds_write_b128 of the wavefront touches 64 consecutive 16-byte pieces).  It holds the per-body contact-sensor state -
// timers (4 words), force history (3), last force (3) - and friction (3), one granule each per body-slot row, and the contact stash
// (9 words in three granules per sphere slot): ~80 values that would otherwise sit in VGPRs all step.  (Until round 4 the layout was
// word-major - word f at base[f * 64 + lane], one ds_read_b32 / ds_write_b32 per word: 16 LDS instructions per owned slot and substep
// in the sensor pass alone, 18 per stashed contact.)
// Rows are indexed by BODY SLOT.  In the 16-lane mapping a lane only ever touches the <= NOWN slots it owns (the
// slots of the link groups it evaluates), so the scratchpad holds NOWN rows and `at` turns a slot into its place in
// the lane's ascending `own[]` list (rows of slots the lane does not own alias row 0: they are only read under a mask).
// NOWN == 0: one row per slot (a lane is a whole leg).
// (One lane per limb - GRAN false: that mapping keeps the word-major layout, word f of a lane at base[f * STRIDE + lane] and W words
// per row: its four-wavefront workgroup sits at the 160 KB of a CU's LDS, and 16 instead of 13 words per body-slot row do not fit.)
template <int STRIDE, int NOWN, int W, bool GRAN>
struct LsMat {
  float* p;  // row 0 (this lane's granule / first word)
  int own[NOWN > 0 ? NOWN : 1];  // copy of the lane's own[] (by value: a pointer into the lane object would pin it in memory)
  RL_FN void set_own(const int* o) {
#pragma unroll
    for (int j = 0; j < (NOWN > 0 ? NOWN : 1); ++j) own[j] = NOWN > 0 ? o[j] : 0;
  }
  RL_FN int at(int b) const {
    if (NOWN == 0) return b;
    int i = 0;
#pragma unroll
    for (int j = 1; j < NOWN; ++j) i += (own[j] == b) ? j : 0;
    return i;
  }
  RL_FN F4 ld(int b) const {
    if constexpr (GRAN) return ld4(p + at(b) * 4 * STRIDE);
    else {
      const float* r = p + at(b) * W * STRIDE;
      return F4{r[0], r[STRIDE], r[2 * STRIDE], W > 3 ? r[(W > 3 ? 3 : 0) * STRIDE] : 0.f};
    }
  }
  RL_FN void st(int b, F4 v) const {
    if constexpr (GRAN) st4(p + at(b) * 4 * STRIDE, v);
    else {
      float* r = p + at(b) * W * STRIDE;
      r[0] = v.x; r[STRIDE] = v.y; r[2 * STRIDE] = v.z;
      if (W > 3) r[(W > 3 ? 3 : 0) * STRIDE] = v.w;
    }
  }
};
// words of limb-shared LDS an instance needs (0 when it keeps everything in registers)
// Trunk + limbs instance (articulated-body form, substeps_aba_trunk): per limb the kinematics (15 words per joint), one 27-word
// record per link group (rigid inertia + contact damping of a link, as seen in base coordinates) and 12 words per limb joint
// that first hold the link velocity / bias acceleration and later the elimination's U / D and u / D; per ENV one 27-word
// accumulator per trunk link (base, waist links, torso) that the limbs and the trunk links' owners ds_add into, and 8 words per
// self-collision capsule (centre, half axis, radius, bounding radius in base coordinates: EnvLane::self_place).
constexpr int LINK_REC = 27;  // 21 (6 x 6 symmetric) + 6
// How a link record sits in LDS.  Default: the packed upper triangle + rho as seven 16-byte vectors, every sub-lane of a limb eliminating
// the whole record.  -DRL_ELIM_ROWS: six rows of 8 words [A[r][0..5], rho_r, 0] - the FULL matrix - so that the sub-lanes of a DPP quad
// can each take rows of it (distributed elimination, EnvLane::eliminate_rows).  Measured in one call (profiles/r04h_elim_ab.txt): 1.2 k
// fewer vector instructions per step and 20 more LDS words per record buy nothing - G1 107.5 against 107.7 us, GR1T1 126.4 against
// 125.8 us: the elimination is a latency chain (reciprocal, quad sums and broadcasts per joint), not an issue-bound block.  Kept as an
// analysis switch.
#ifdef RL_ELIM_ROWS
constexpr bool REC_ROWS = true;
constexpr int REC_STRIDE = 48;
#else
constexpr bool REC_ROWS = false;
constexpr int REC_STRIDE = 28;
#endif
// a limb's block padded to 4 (mod 32) words: the blocks of the 8 / 16 limbs of a wavefront then start in different banks (b32: 32 banks,
// b128: 64), so that the same word of every limb - what a wavefront instruction touches - is conflict free
constexpr int pad_limb_block(int w) { return ((w - 4 + 31) / 32) * 32 + 4; }
template <class TP>
struct LbLayout {
  enum { JW = 16, CHAINW = TP::NW > 0 ? pad_limb_block(TP::JX * JW) : 0,          // the limb's kinematics: one block per limb, all limbs' blocks together
         REC = 0, VA = TP::CL * REC_STRIDE, RECW = TP::NW > 0 ? pad_limb_block(TP::CL * REC_STRIDE + TP::CL * 12) : 0,  // records of links 1 .. CL | per-joint words
         WORDS = CHAINW + RECW, AUX_WORDS = 0,
         ENV_WORDS = TP::NW > 0 ? (TP::NW + 1) * REC_STRIDE + SELF_CAPS * SELF_CAP_WORDS : 0 };  // trunk accumulators | placed self-collision capsules
};

// STASH > 0: room for the contacts of one link group (x, n, bias, d_n, d_t per sphere slot) so that the sensor
// pass after the solve does not re-evaluate them (quadrupeds in the 16-lane mapping; 4 workgroups x 40 KB of LDS
// still share a CU)
constexpr int CONTACT_WORDS = 9;  // x (3), n (3), bias, d_n, d_t
constexpr int STASH_SLOT_WORDS = 12;  // ... in the lane scratchpad: three granules
template <int ROWS, int STASH = 0, bool GRAN = true, bool STASH_GRAN = true>
struct LsLayout {  // word offsets: ROWS sensor rows (timers 4, force history 3, last force 3, friction 3: a granule each, or - word-major - W words) + the contact stash
  enum { RW = GRAN ? 4 : 3, SSW = STASH_GRAN ? STASH_SLOT_WORDS : CONTACT_WORDS, TIM = 0, HIST = TIM + ROWS * 4, CF = HIST + ROWS * RW, FRIC = CF + ROWS * RW, CT = FRIC + ROWS * RW,
         WORDS = CT + STASH * SSW };
};
template <class TP, int SUB>
struct LsFor {  // lane scratchpad layout of an instance
  // link groups a sub-lane evaluates: g = sub + SUB * it over the groups 0 .. CL (merged instances: 1 .. CL, g = 1 + sub + SUB * it)
  static constexpr int NIT = (TP::CL + SUB - TP::M0) / SUB;
  // every contact of pass 1 is kept for the sensor pass (slot it * SPL + s): in the lane scratchpad - or, with eight sub-lanes per limb
  // (ONE link group per sub-lane: 4 slots x 9 words), in REGISTERS: 9.2 KB of LDS per wavefront less, which is what lets four
  // wavefronts of the six-joint-spine instance (GR1) share a CU (2048 envs: 236 -> one round), and 72 scratchpad instructions per
  // touching lane and substep
#ifdef RL_STASH_REG_QUAD  // (A/B switch: the 16-lane quadruped kernels keep their 3-slot stash in registers as well)
  static constexpr bool STASH_REG = SUB == 8 || (SUB == 4 && TP::NW == 0);
#else
  static constexpr bool STASH_REG = SUB == 8;
#endif
  static constexpr int STASH = (SUB > 1 && !STASH_REG) ? NIT * TP::SPL : 0;
  static constexpr int NOWN = SUB == 1 ? 0 : LaneTabT<TP>::template maxown<SUB>();  // 16- / 8-lane mappings: rows for the owned slots only
  static constexpr bool GRAN = SUB != 1;  // 16-byte granules (see LsMat)
  static constexpr int ROWS = SUB == 1 ? TP::NBS : LaneTabT<TP>::template maxown<SUB>();
  // the stash as granules (12 instead of 9 words per slot) where four wavefronts of the instance still fit a CU's LDS next to the
  // tables (not the 4-joint quadruped with nine stash slots per lane in the 8-lane mapping: it keeps a word-major stash)
  static constexpr bool STASH_GRAN = GRAN && 4 * (16 * ROWS + STASH_SLOT_WORDS * STASH) * 256 <= 140 * 1024;
  using type = LsLayout<ROWS, STASH, GRAN, STASH_GRAN>;
};

template <class Ctx, class TP, class SP = NoSpec>
struct EnvLane {
  static constexpr int CL = TP::CL, NW = TP::NW, JX = TP::JX, NB = TP::NB, SPL = TP::SPL, NBS = TP::NBS, NGRP = TP::CL + 1;
  using LS = typename LsFor<TP, Ctx::SUB>::type;
  static constexpr bool STASH_REG = LsFor<TP, Ctx::SUB>::STASH_REG;
  static constexpr bool STASH = LsFor<TP, Ctx::SUB>::STASH > 0 || STASH_REG;
  float stash_r[STASH_REG ? LsFor<TP, Ctx::SUB>::NIT * TP::SPL : 1][CONTACT_WORDS];  // the register stash (static indices only)
  static constexpr int LSS = Ctx::LS_STRIDE;
  static constexpr int SUB = Ctx::SUB;          // sub-lanes per leg (1: one lane per leg; 4: a DPP quad per leg)
  static constexpr int LPE = NLANE * SUB;       // lanes per environment
  static constexpr int EPT = 64 / LPE;          // environments per wavefront / tile
  static constexpr uint32_t ROW = NLANE * EPT;  // entries of a lane-field row
  static constexpr int NV = NB + CL;  // per-lane system: [omega_b, v_b, trunk joints, limb joints]
  using UI = SymIdx<NV>;
  // G1-sized instances keep the kinematics and the (NV x NV) system in limb-shared LDS words instead of
  // VGPRs (hipcc otherwise spills ~3.5 KB per lane to scratch: 0.9 GB of HBM traffic per step at 2048 envs)
  static constexpr Layout LY{CL, NW, NBS};  // field rows of this instance's HBM tiles
  static constexpr bool LDSU = TP::NW > 0;
  using ChainTP = ChainT<TP, LDSU, 1>;
  enum { LB_REC = LbLayout<TP>::REC, LB_VA = LbLayout<TP>::VA };
  template <class FT = NoJoint, class FL = NoJoint>
  RL_FN void kinematics(ChainTP& C, FT&& on_trunk = FT{}, FL&& on_limb = FL{}) {
    if constexpr (KIN_SCAN && std::is_same<typename std::decay<FT>::type, NoJoint>::value && std::is_same<typename std::decay<FL>::type, NoJoint>::value) {
      SV s0[NW > 0 ? NW : 1], s1[NW > 0 ? NW : 1], s2[NW > 0 ? NW : 1];
      const SV z{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
      kinematics_scan<false>(C, z, z, s0, s1, s2);
      return;
    }
#ifndef RL_KIN_REPLICATED  // (A/B switch: every sub-lane computes every joint transform)
    if constexpr (NW > 0 && SUB > 1) {
      chain_kinematics_dealt<TP, (SUB < 4 ? SUB : 4)>(ctx, SUB > 4 ? (sub & 3) : sub, L, q, C, u.trunk_restart, on_trunk, on_limb);
      return;
    }
#endif
    chain_kinematics<TP, SP>(L, q, C, u.trunk_restart, on_trunk, on_limb);
  }
  RL_FN ChainTP new_chain() const { return ChainTP(LDSU ? ctx.limb_chain() : nullptr); }

  // ---- Kinematics with ONE LIMB JOINT PER SUB-LANE (round 5; eight sub-lanes per limb, a limb has at most seven joints).
  // The dealt form above still walks the limb's chain in every sub-lane - ten joints x (nine broadcasts, R_j = R_parent T_j, origin, axis,
  // twist, bias acceleration): 1.4 k of a substep's 6 k vector instructions on G1, eight times the same numbers.  Here sub-lane s owns limb
  // joint s: its local transform A_s = (rot0_s Rodrigues(axis_s, q_s), origin_s), then an inclusive prefix product
  // A_0 o ... o A_s over the limb's lanes in three DPP steps (row_shr 1, 2, 4: (R1, p1) o (R2, p2) = (R1 R2, p1 + R1 p2)), the limb's
  // attachment frame in front, and the lane writes ITS joint's words of the limb-shared chain.  Link twists V_j = V_attach + sum_{i <= j}
  // S_i qd_i and bias accelerations a_j = a_attach + sum_{i <= j} V_i x S_i qd_i are two prefix SUMS of six words (VEL).  The trunk joints
  // (shared by all limbs) stay a chain in every lane, their local transforms dealt over the lane's DPP quad as before.
  // Products and sums associate as a tree here and left to right there: round-off apart (the parity tiers' tolerances), not bits.
  // Readers of another lane's words need the wave-local fence (ctx.group_sync) first.
#ifdef RL_KIN_DEALT  // (A/B switch: the dealt chain of round 4)
  static constexpr bool KIN_SCAN = false;
#else
  static constexpr bool KIN_SCAN = NW > 0 && SUB == 8 && CL <= 8;
#endif
  template <bool VEL>
  RL_FN void kinematics_scan(ChainTP& C, const SV V0, const SV a0, SV (&Sw)[NW > 0 ? NW : 1], SV (&Vw)[NW > 0 ? NW : 1], SV (&aw)[NW > 0 ? NW : 1]) {
    static_assert(NW > 0 && SUB == 8, "trunk + limbs instance, eight sub-lanes per limb");
    const int sq = sub & 3;
    // the trunk joints' local transforms, dealt over the lane's quad: trunk joint 4 i + sq in round i (a partial last round is clamped, unused)
    constexpr int NT4 = (NW + 3) / 4;
    M3 Tt[NT4];
    static_for<0, NT4>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      float qi = opaque(q[CL + 4 * i]);  // (opaque: rl_math.h - a select over plain loads of q becomes one load through a selected address)
      static_for<1, 4>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s2 = decltype(sc)::value, tq = 4 * i + s2 < NW ? 4 * i + s2 : NW - 1;
        const float cand = opaque(q[CL + tq]);
        qi = sq == s2 ? cand : qi;
      });
      const int jx = CL + imin(4 * i + sq, NW - 1);
      const F4 r0 = ld4(L.rota[jx]), r1 = ld4(L.rota[jx] + 4), r2 = ld4(L.rota[jx] + 8);  // rot0 (row-major) and the axis: three vectors
      Tt[i] = mul(M3{{r0.x, r0.y, r0.z}, {r0.w, r1.x, r1.y}, {r1.z, r1.w, r2.x}}, rodrigues(V3{r2.y, r2.z, r2.w}, qi));
    });
    // this lane's limb joint
    const bool has = sub < CL;
    const int js = has ? sub : CL - 1;
    float qs = opaque(q[0]), qds = opaque(qd[0]);
    static_for<1, CL>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      const float cq = opaque(q[j]), cd = opaque(qd[j]);
      qs = js == j ? cq : qs;
      qds = js == j ? cd : qds;
    });
    M3 Rl;
    V3 pl, al_s;
    {
      const F4 r0 = ld4(L.rota[js]), r1 = ld4(L.rota[js] + 4), r2 = ld4(L.rota[js] + 8), c0 = ld4(L.jc[js]);
      al_s = {r2.y, r2.z, r2.w};
      const M3 Tl = mul(M3{{r0.x, r0.y, r0.z}, {r0.w, r1.x, r1.y}, {r1.z, r1.w, r2.x}}, rodrigues(al_s, qs));
      Rl = select_m3(has, Tl, identity3());
      pl = select3(has, V3{c0.x, c0.y, c0.z}, V3{0.f, 0.f, 0.f});
    }
    // trunk chain (the same in every lane)
    M3 Rp = identity3(), Ra = identity3();
    V3 pp{0.f, 0.f, 0.f}, pa{0.f, 0.f, 0.f};
    SV Vp = V0, ap = a0, Va = V0, aa = a0;
    static_for<0, NW>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value, jx = CL + i;
      if (i > 0 && ((u.trunk_restart >> i) & 1u)) { Rp = identity3(); pp = {0.f, 0.f, 0.f}; Vp = V0; ap = a0; }  // a trunk piece that starts at the base
      const M3 Tj = ctx.template deal_bcast_m3<i % 4>(Tt[i / 4]);
      V3 oj, alj;
      joint_origin_axis(L, jx, oj, alj);
      const V3 pj = pp + mul(Rp, oj);
      const M3 Rj = mul(Rp, Tj);
      const V3 axj = mul(Rj, alj);
      C.setw(i, Rj, pj, axj);
      if constexpr (VEL) {
        Sw[i] = SV{axj, cross(pj, axj)};
        const SV vj = Sw[i] * qd[jx];
        Vw[i] = Vp + vj;
        aw[i] = ap + crm(Vw[i], vj);
        Vp = Vw[i];
        ap = aw[i];
        if (L.attach == i + 1) { Va = Vw[i]; aa = aw[i]; }
      }
      Rp = Rj;
      pp = pj;
      if (L.attach == i + 1) { Ra = Rj; pa = pj; }
    });
    // inclusive prefix product of the local transforms over the limb's sub-lanes
    static_for<0, 3>([&](auto dc) __attribute__((always_inline)) {
      constexpr int D = 1 << decltype(dc)::value;
      float w[12] = {Rl.r0.x, Rl.r0.y, Rl.r0.z, Rl.r1.x, Rl.r1.y, Rl.r1.z, Rl.r2.x, Rl.r2.y, Rl.r2.z, pl.x, pl.y, pl.z};
      ctx.template sub_shr<D>(w);
      const M3 Rq{{w[0], w[1], w[2]}, {w[3], w[4], w[5]}, {w[6], w[7], w[8]}};
      const V3 pq{w[9], w[10], w[11]};
      const bool take = sub >= D;
      const M3 Rn = mul(Rq, Rl);
      const V3 pn = pq + mul(Rq, pl);
      Rl = select_m3(take, Rn, Rl);
      pl = select3(take, pn, pl);
    });
    const M3 Rj = mul(Ra, Rl);
    const V3 pj = pa + mul(Ra, pl);
    const V3 axj = mul(Rj, al_s);  // (Rodrigues(axis, .) leaves its own axis where it is: R_j axis = rot-frame axis)
    if (has) C.set(js, Rj, pj, axj);
    if constexpr (VEL) {
      const SV Sj{axj, cross(pj, axj)};
      const SV vj = Sj * (has ? qds : 0.f);
      float vs[6] = {vj.a.x, vj.a.y, vj.a.z, vj.l.x, vj.l.y, vj.l.z};
      static_for<0, 3>([&](auto dc) __attribute__((always_inline)) {
        constexpr int D = 1 << decltype(dc)::value;
        float w[6] = {vs[0], vs[1], vs[2], vs[3], vs[4], vs[5]};
        ctx.template sub_shr<D>(w);
        const bool take = sub >= D;
#pragma unroll
        for (int i = 0; i < 6; ++i) vs[i] += take ? w[i] : 0.f;
      });
      const SV Vj = Va + SV{{vs[0], vs[1], vs[2]}, {vs[3], vs[4], vs[5]}};
      const SV cj = crm(Vj, vj);
      float cs[6] = {cj.a.x, cj.a.y, cj.a.z, cj.l.x, cj.l.y, cj.l.z};
      static_for<0, 3>([&](auto dc) __attribute__((always_inline)) {
        constexpr int D = 1 << decltype(dc)::value;
        float w[6] = {cs[0], cs[1], cs[2], cs[3], cs[4], cs[5]};
        ctx.template sub_shr<D>(w);
        const bool take = sub >= D;
#pragma unroll
        for (int i = 0; i < 6; ++i) cs[i] += take ? w[i] : 0.f;
      });
      const SV aj = aa + SV{{cs[0], cs[1], cs[2]}, {cs[3], cs[4], cs[5]}};
      if (has) {
        float* w = va_words(js);
        st4(w, F4{Vj.a.x, Vj.a.y, Vj.a.z, Vj.l.x});
        st4(w + 4, F4{Vj.l.y, Vj.l.z, aj.a.x, aj.a.y});
        st4(w + 8, F4{aj.a.z, aj.l.x, aj.l.y, aj.l.z});
      }
    }
  }

  Ctx& ctx;
  const KState& S;
  const TablesT<TP>& T;
  const LaneTabT<TP>& L;
  const Uni u;
  int e, k, sub, li, Np;  // env, leg, sub-lane of the leg, lane index inside the env (k * SUB + sub)
  static constexpr int MAXOWN = SUB == 1 ? NBS : LaneTabT<TP>::template maxown<SUB>();
  int own[MAXOWN];        // body slots this lane updates every substep (all of them when a lane is a whole leg)
  // HBM state tiles.  Two address forms, picked per lane mapping (one-call A/B of both on three instances: profiles/r04i_state_buf_ab.txt):
  //  * per-lane column pointers (lt / et; field f at lt[f * ROW]) - 16 / 8 / 32 lanes per env.  The compiler hoists the loop-invariant
  //    loads of the substeps (link inertias, friction) out of the loop.
  //  * BUFFERS - one lane per limb: a wave-uniform descriptor of the wavefront's tile (scalar registers), the lane's byte offset in a
  //    field row (ONE vector register for all fields) and the field's row offset as the instruction's scalar offset
  //    (buffer_load_dword v, v_off, s[rsrc], s_field offen).  With column pointers that instance - 512 registers in use - materialised
  //    a separate vector address pair for every field beyond the 4 KB immediate range of global_load, hoisted the 27 of them out of the
  //    substep loop and spilled them: 260 B of scratch.  Buffers: no scratch, 44 fewer registers, 108.4 -> 97.6 us at 16384 envs.
  //    (The other mappings LOSE with buffers - A1 42.7 -> 50.5 us, G1 107.4 -> 127.6 us: buffer loads are not hoisted, every substep
  //    re-reads its constants from HBM on the critical path of a lone wavefront.)
#ifdef RL_STATE_BUF_ALL  // buffers in every mapping (A/B)
  static constexpr bool STATE_BUF = true;
#else
  static constexpr bool STATE_BUF = SUB == 1;
#endif
  float* lt;  // this leg's column of the wave tile:   field f -> lt[f * ROW]
  float* et;  // this env's column of the env tile:    field f -> et[f * EPT]
  typename Ctx::StateBuf lt_b, et_b;  // the wavefront's lane-state / env-state tile
  uint32_t lt_off, et_off;            // byte offset of this leg's / this env's column in a field row
  // what LF(f) / EF(f) hand out: reads as a float, assigns as one
  struct BufRef {
    const typename Ctx::StateBuf& b;
    uint32_t voff, soff;
    RL_FN operator float() const { return Ctx::buf_ld(b, voff, soff); }
    RL_FN float operator=(float v) const { Ctx::buf_st(b, voff, soff, v); return v; }
  };
  struct PtrRef {
    float* p;
    RL_FN operator float() const { return *p; }
    RL_FN float operator=(float v) const { *p = v; return v; }
  };
  // persistent state in registers
  V3 pos, vlin, vang;
  Q4 quat;
  float q[JX], qd[JX], kp[JX], kd[JX], act[JX], prev_act[JX];  // [0, CL) limb, [CL, JX) trunk joints
  V3 base_com;  // COM of the root body (root COM velocity)
  V3 wr_com;    // COM of the wrench body in its trunk link frame
  V3 extF, extT;
  float fric0[3];  // merged instances: material of the trunk body in slot 0 (its spheres may sit with any sub-lane; the rows below are the owner's)
  // per-step scratch
  float tau_app[JX], qacc[JX];
  // contact-sensor state + friction in the lane-private LDS scratchpad
  static constexpr int NOWN = LsFor<TP, Ctx::SUB>::NOWN;
  static constexpr bool GRAN = LsFor<TP, Ctx::SUB>::GRAN;
  LsMat<LSS, NOWN, 4, GRAN> tim;     // [slot] (current_air, current_contact, last_air, last_contact)
  LsMat<LSS, NOWN, 3, GRAN> hist_n;  // [slot] (|F| of the last three substeps, newest first; -)
  LsMat<LSS, NOWN, 3, GRAN> cf;      // [slot] (net contact force of the last substep, world; -)
  LsMat<LSS, NOWN, 3, GRAN> fric;    // [slot] (mu_s, mu_d, restitution; -)

  RL_FN EnvLane(Ctx& c, const KState& s)
      : ctx(c), S(s), T(c.template tables<TablesT<TP>>()), L(c.template tables<TablesT<TP>>().lane[c.k()]), u(make_uni(c, c.template tables<TablesT<TP>>())), tim{c.template lane_scratch<GRAN>() + LS::TIM * LSS, {}}, hist_n{c.template lane_scratch<GRAN>() + LS::HIST * LSS, {}},
        cf{c.template lane_scratch<GRAN>() + LS::CF * LSS, {}}, fric{c.template lane_scratch<GRAN>() + LS::FRIC * LSS, {}} {
    e = ctx.env();
    k = ctx.k();
    sub = ctx.sub();
    li = k * SUB + sub;
    Np = S.Npad;
#pragma unroll
    for (int i = 0; i < MAXOWN; ++i) {
      if constexpr (SUB == 8) own[i] = -1;
      else own[i] = SUB == 1 ? i : (SUB == 2 ? L.own_slot2[sub & 1][i < LaneTabT<TP>::MAXOWN2 ? i : 0] : L.own_slot[sub & 3][i < LaneTabT<TP>::MAXOWN ? i : 0]);
    }
    if constexpr (SUB == 8) {  // the used slots of link group `sub`, ascending (the host has checked that they are at most MAXOWN: TaskTab::sub8_ok)
      int n = 0;
#pragma unroll
      for (int sl = 0; sl < NBS; ++sl) {
        const bool mine = (L.slot_body[sl] >= 0 || (sl == 0 && L.base_body_local >= 0)) && (L.slot_grp[sl] & 7) == sub;
#pragma unroll
        for (int i = 0; i < MAXOWN; ++i) own[i] = (mine && n == i) ? sl : own[i];
        n += mine ? 1 : 0;
      }
    }
    tim.set_own(own); hist_n.set_own(own); cf.set_own(own); fric.set_own(own);
    if constexpr (STATE_BUF) {
      lt_b = Ctx::state_buf(ctx.uniform_ptr(S.lane_state + (size_t)ctx.tile() * ((size_t)LY.NF_LANE * ROW)), (uint32_t)LY.NF_LANE * ROW * (uint32_t)sizeof(float));
      et_b = Ctx::state_buf(ctx.uniform_ptr(S.env_state + (size_t)ctx.tile() * ((size_t)LY.NF_ENV * EPT)), (uint32_t)LY.NF_ENV * (uint32_t)EPT * (uint32_t)sizeof(float));
      lt_off = (uint32_t)sizeof(float) * (uint32_t)(ctx.env_in_tile() * NLANE + k);
      et_off = (uint32_t)sizeof(float) * (uint32_t)ctx.env_in_tile();
    } else {
      lt = S.lane_state + (size_t)ctx.tile() * ((size_t)LY.NF_LANE * ROW) + (uint32_t)(ctx.env_in_tile() * NLANE + k);
      et = S.env_state + (size_t)ctx.tile() * ((size_t)LY.NF_ENV * EPT) + (uint32_t)ctx.env_in_tile();
    }
  }
#ifdef RL_PHASE_CLOCK_ON
  long long ph_t0 = 0;
  int ph_cur = 0;
  __device__ __forceinline__ void phase_stamp(int id) {
    __builtin_amdgcn_s_waitcnt(0);
    const long long t = (long long)__builtin_readcyclecounter();
    if (ctx.lane == 0 && ph_t0 != 0) {  // (the first stamp has no predecessor)
      float* row = S.rew_terms + (size_t)RL_PHASE_ROW0 * (size_t)S.Npad + (size_t)ctx.tile() * RL_PHASE_SLOTS;
      row[ph_cur] += (float)(t - ph_t0);
    }
    ph_cur = id;
    __builtin_amdgcn_s_waitcnt(0);
    ph_t0 = (long long)__builtin_readcyclecounter();
  }
#endif
  // (a field index known at compile time rides in the scalar offset; one that differs between lanes - the link of a lane's group, 16 / 32
  // lanes per env - must go into the vector offset: a divergent scalar offset is a waterfall loop per load)
  RL_FN auto LF(int f) const {
    if constexpr (STATE_BUF) {
      const uint32_t o = (uint32_t)f * ROW * (uint32_t)sizeof(float);
      return __builtin_constant_p(f) ? BufRef{lt_b, lt_off, o} : BufRef{lt_b, lt_off + o, 0u};
    } else return PtrRef{lt + (uint32_t)f * ROW};
  }
  RL_FN auto EF(int f) const {
    if constexpr (STATE_BUF) {
      const uint32_t o = (uint32_t)f * (uint32_t)EPT * (uint32_t)sizeof(float);
      return __builtin_constant_p(f) ? BufRef{et_b, et_off, o} : BufRef{et_b, et_off + o, 0u};
    } else return PtrRef{et + (uint32_t)f * (uint32_t)EPT};
  }
  // link group g (0 = base share, 1.. = chain links) is evaluated by sub-lane g % SUB of the leg; merged instances (Topo::M0) have
  // no group 0 - their base-share spheres sit in flagged slots of group 1 -, group g is sub-lane (g - 1) % SUB's and the trunk
  // body's slot (group 0) belongs to sub-lane 0
  static constexpr bool M0 = TP::M0 != 0;
  static constexpr int G0 = M0 ? 1 : 0;  // first link group of the sub-lane mapping
  RL_FN int grp_of(int it) const { return G0 + sub + SUB * it; }
  RL_FN bool owns_group(int g) const { return SUB == 1 || (g < G0 ? sub == 0 : ((g - G0) % SUB) == sub); }
  // does the sphere in slot (g, s) ride on the base link instead of limb link g - 1 (merged instances)
  RL_FN bool on_base(int g, int s) const { return M0 && ((L.sph_base_mask >> (g * SPL + s)) & 1u) != 0u; }
  // componentwise select (a ?: on the struct becomes a select of two ADDRESSES and a private-memory copy of both operands)
  RL_FN static SV pick_sv(bool c, const SV& a, const SV& b) {
    return SV{{c ? a.a.x : b.a.x, c ? a.a.y : b.a.y, c ? a.a.z : b.a.z}, {c ? a.l.x : b.l.x, c ? a.l.y : b.l.y, c ? a.l.z : b.l.z}};
  }
  // activity bits of the lane's base-share spheres
  RL_FN uint32_t base_bits() const { return M0 ? L.sph_base_mask : ((1u << SPL) - 1u); }
  RL_FN bool owns_slot(int s) const { return owns_group(L.slot_grp[s]); }

  // ------------------------------------------------------------------ load / store
  RL_FN void load() {
    pos = {EF(LY.EF_ROOT + 0), EF(LY.EF_ROOT + 1), EF(LY.EF_ROOT + 2)};
    quat = {EF(LY.EF_ROOT + 3), EF(LY.EF_ROOT + 4), EF(LY.EF_ROOT + 5), EF(LY.EF_ROOT + 6)};
    vlin = {EF(LY.EF_ROOT + 7), EF(LY.EF_ROOT + 8), EF(LY.EF_ROOT + 9)};
    vang = {EF(LY.EF_ROOT + 10), EF(LY.EF_ROOT + 11), EF(LY.EF_ROOT + 12)};
    extF = {EF(LY.EF_WRENCH + 0), EF(LY.EF_WRENCH + 1), EF(LY.EF_WRENCH + 2)};
    extT = {EF(LY.EF_WRENCH + 3), EF(LY.EF_WRENCH + 4), EF(LY.EF_WRENCH + 5)};
    base_com = {EF(LY.EF_BASE_COM + 0), EF(LY.EF_BASE_COM + 1), EF(LY.EF_BASE_COM + 2)};
    if (NW > 0) wr_com = {EF(LY.EF_WR_COM + 0), EF(LY.EF_WR_COM + 1), EF(LY.EF_WR_COM + 2)};
    else wr_com = base_com;  // quadrupeds: the wrench body is the root body
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      q[j] = LF(LY.LF_Q + j);
      qd[j] = LF(LY.LF_QD + j);
      kp[j] = LF(LY.LF_KP + j);
      kd[j] = LF(LY.LF_KD + j);
      act[j] = LF(LY.LF_ACT + j);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      q[CL + i] = EF(LY.EF_TQ + i);
      qd[CL + i] = EF(LY.EF_TQD + i);
      kp[CL + i] = EF(LY.EF_TKP + i);
      kd[CL + i] = EF(LY.EF_TKD + i);
      act[CL + i] = EF(LY.EF_TACT + i);
    }
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      tau_app[j] = 0.f;
      qacc[j] = 0.f;
    }
    if (M0) {
#pragma unroll
      for (int t = 0; t < 3; ++t) fric0[t] = LF(LY.LF_FRICTION + t);
    }
#pragma unroll
    for (int i = 0; i < MAXOWN; ++i) {  // sensor state + material of the body slots this lane owns (the others are never read unmasked)
      const int s = own[i];
      if (s < 0) continue;
      tim.st(s, F4{LF(LY.LF_TIMERS + s * 4 + 0), LF(LY.LF_TIMERS + s * 4 + 1), LF(LY.LF_TIMERS + s * 4 + 2), LF(LY.LF_TIMERS + s * 4 + 3)});
      fric.st(s, F4{LF(LY.LF_FRICTION + s * 3 + 0), LF(LY.LF_FRICTION + s * 3 + 1), LF(LY.LF_FRICTION + s * 3 + 2), 0.f});
      cf.st(s, F4{0.f, 0.f, 0.f, 0.f});
      hist_n.st(s, F4{0.f, 0.f, 0.f, 0.f});
    }
  }

  RL_FN void store() {
    if (li == 0) {
      EF(LY.EF_ROOT + 0) = pos.x; EF(LY.EF_ROOT + 1) = pos.y; EF(LY.EF_ROOT + 2) = pos.z;
      EF(LY.EF_ROOT + 3) = quat.w; EF(LY.EF_ROOT + 4) = quat.x; EF(LY.EF_ROOT + 5) = quat.y; EF(LY.EF_ROOT + 6) = quat.z;
      EF(LY.EF_ROOT + 7) = vlin.x; EF(LY.EF_ROOT + 8) = vlin.y; EF(LY.EF_ROOT + 9) = vlin.z;
      EF(LY.EF_ROOT + 10) = vang.x; EF(LY.EF_ROOT + 11) = vang.y; EF(LY.EF_ROOT + 12) = vang.z;
      EF(LY.EF_WRENCH + 0) = extF.x; EF(LY.EF_WRENCH + 1) = extF.y; EF(LY.EF_WRENCH + 2) = extF.z;
      EF(LY.EF_WRENCH + 3) = extT.x; EF(LY.EF_WRENCH + 4) = extT.y; EF(LY.EF_WRENCH + 5) = extT.z;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        EF(LY.EF_TQ + i) = q[CL + i];
        EF(LY.EF_TQD + i) = qd[CL + i];
        EF(LY.EF_TKP + i) = kp[CL + i];
        EF(LY.EF_TKD + i) = kd[CL + i];
        EF(LY.EF_TACT + i) = act[CL + i];
      }
    }
    if (sub == 0) {
#pragma unroll
      for (int j = 0; j < CL; ++j) {
        LF(LY.LF_Q + j) = q[j];
        LF(LY.LF_QD + j) = qd[j];
        LF(LY.LF_KP + j) = kp[j];
        LF(LY.LF_KD + j) = kd[j];
        LF(LY.LF_ACT + j) = act[j];
      }
    }
#pragma unroll
    for (int i = 0; i < MAXOWN; ++i) {
      const int s = own[i];
      if (s < 0 || (s == 0 && SUB > 1 && L.slot_body[0] < 0)) continue;  // slot 0 of a lane that only shares a trunk body's spheres
      const F4 t4 = tim.ld(s);
      LF(LY.LF_TIMERS + s * 4 + 0) = t4.x; LF(LY.LF_TIMERS + s * 4 + 1) = t4.y; LF(LY.LF_TIMERS + s * 4 + 2) = t4.z; LF(LY.LF_TIMERS + s * 4 + 3) = t4.w;
    }
  }

  // ------------------------------------------------------------------ actuators [UPSTREAM B4]
  // returns explicit torque; fills tau_app (applied torque estimate) and the implicit-PD diagonal terms
  RL_FN void actuators(const float (&q_tgt)[JX], const float (&qd_tgt)[JX], float (&tau_e)[JX], float (&pd_diag)[JX], float (&pd_rhs)[JX]) {
    const float dt = u.dt;
    // the joints' actuator constants: one batch of LDS reads, in registers before the first branch on them (rl_pin) - read joint by joint,
    // each read sits behind the branch of the joint before it.  Quadrupeds: A1 36.29 -> 35.68 us.  The trunk + limbs instances lose by it
    // (G1 94.83 -> 95.20: forty more live registers where the kernel has none to spare) and keep the joint-by-joint form
    // (profiles/r05s_actuator_batch_ab.txt)
    float a_eff[JX], a_sat[JX], a_vlim[JX], a_flags[JX];
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      const F4 c1 = ld4(L.jc[j] + 4), c2 = ld4(L.jc[j] + 8);  // [. . eff sat | act_vlim flags . .]
      a_eff[j] = c1.z; a_sat[j] = c1.w; a_vlim[j] = c2.x; a_flags[j] = c2.y;
    }
    if constexpr (NW == 0) { rl_pin(a_eff); rl_pin(a_sat); rl_pin(a_vlim); rl_pin(a_flags); }
#pragma unroll
    for (int j = 0; j < JX; ++j)
      actuator_one(q[j], qd[j], kp[j], kd[j], q_tgt[j], qd_tgt[j], a_eff[j], a_sat[j], a_vlim[j], (int)a_flags[j], tau_app[j], tau_e[j], pd_diag[j], pd_rhs[j]);
  }
  // one joint: applied torque estimate, explicit torque, the implicit PD's diagonal and right-hand-side terms
  RL_FN void actuator_one(float qj, float qdj, float kpj, float kdj, float qtj, float qdtj, float eff, float satq, float vlim, int flags, float& t_app,
                          float& t_e, float& p_diag, float& p_rhs) const {
    const float dt = u.dt;
    float qt = (flags & 2) ? qj : qtj;
    float er = qt - qj, ed = qdtj - qdj;
    float tc = kpj * er + kdj * ed;
    if (flags & 1) {
      float est = clampf(tc, -eff, eff);
      bool sat = fabsf(tc) > eff;
      t_app = est;
      t_e = sat ? est : 0.f;
      p_diag = sat ? 0.f : dt * (kdj + kpj * dt);
      p_rhs = sat ? 0.f : dt * (kpj * er + kdj * qdtj);
    } else {  // DCMotor torque-speed clip (unitree.py:55-63)
      float vr = qdj * frcp(vlim);
      float tmax = clampf(satq * (1.0f - vr), 0.f, eff);
      float tmin = clampf(satq * (-1.0f - vr), -eff, 0.f);
      float t = clampf(tc, tmin, tmax);
      t_app = t;
      t_e = t;
      p_diag = 0.f;
      p_rhs = 0.f;
    }
  }

  // ------------------------------------------------------------------ contact of one sphere slot
  struct Contact {
    bool act;
    V3 x, n;            // contact point and terrain normal, base coordinates
    float bias, dn, dt; // spring bias force, normal / tangential damping
  };
  // joints that move link group g: the first wdepth(g) trunk joints and the first g limb joints
  RL_FN int wdepth(int g) const { return NW == 0 ? 0 : (g == 0 ? L.grp0_depth : L.attach); }
  RL_FN uint32_t trunk_anc(int g) const { return NW == 0 ? 0u : T.trunk_anc[wdepth(g)]; }  // the trunk joints that move link group g
  // sphere centre in base coordinates (cb) and world (cw: x, y as OFFSETS from the root position - the terrain lookup adds the
  // root in fp64 -, z the world height); empty slots (radius <= 0) sit at the group's link origin
  // frame (base coordinates) of link group g: the trunk link of group 0, else limb link g - 1.  Selected once
  // per group (12 selects per candidate) instead of one matrix-vector product per candidate per sphere.
  RL_FN void group_frame(const ChainTP& C, int g, M3& Rg, V3& pg) const {
    Rg = identity3();
    pg = {0.f, 0.f, 0.f};
    if (NW > 0) trunk_frame<TP>(C, L.grp0_depth, Rg, pg);
#pragma unroll
    for (int j = 0; j < CL; ++j)
      if (g == j + 1) { Rg = C.R(j); pg = C.p(j); }
  }
  RL_FN void sphere_center_in(const M3& Rg, V3 pg, const M3& Rwb, int g, int s, float& rad, V3& cb, V3& cw) const {
    const F4 sp = ld4(L.sph[g][s]);  // centre (link frame), radius
    rad = sp.w;
    const V3 c{sp.x, sp.y, sp.z};
    cb = pg + mul(Rg, c);
    if (M0 && on_base(g, s)) cb = c;  // a base-share sphere hosted by this group's slots: base frame
    const V3 ow = mul(Rwb, cb);
    cw = {ow.x, ow.y, pos.z + ow.z};
  }
  RL_FN void sphere_center(const ChainTP& C, const M3& Rwb, int g, int s, float& rad, V3& cb, V3& cw) const {
    M3 Rg;
    V3 pg;
    group_frame(C, g, Rg, pg);
    sphere_center_in(Rg, pg, Rwb, g, s, rad, cb, cw);
  }
  // penetration depth and world normal of a fetched patch (straight-line: three of these interleave)
  RL_FN void patch_phi(const TerrainPatch& tp, float rad, V3 cw, float& phi, V3& nw) const {
    float hz;
    terrain_eval(u, tp, hz, nw);
    phi = rad > 0.f ? rad - (cw.z - hz) * nw.z : -1.f;
  }
  // `Vg`: spatial velocity (base coordinates, referred to the base origin) of the link the sphere rides on - the velocity of a
  // point x of that link is Vg.l + Vg.a x x, whatever the number of joints between it and the base
  RL_FN Contact contact_from_phi(const M3& Rwb, const SV& Vg, int g, int s, float rad, V3 cb, float phi, V3 nw) const {
    Contact c;
    c.act = false;
    if (phi > 0.f) {
      V3 nb = mulT(Rwb, nw);
      V3 x = cb - rad * nb;
      V3 uu = Vg.l + cross(Vg.a, x);
      float un = dot(nb, uu);
      V3 ut = uu - un * nb;
      float utn = norm(ut);
      // (the friction row read ahead, under the penetration arithmetic of the group's slots, was tried: +0.6 % on A1 - profiles/r05t_substep_batches_ab.txt)
      int slot = L.sph_slot[g][s];
      const F4 fr = fric.ld(slot);
      float mus = fr.x, mud = fr.y, rest = fr.z;
      if (M0) {
        const bool ob = on_base(g, s);
        mus = ob ? fric0[0] : mus; mud = ob ? fric0[1] : mud; rest = ob ? fric0[2] : rest;
      }
      float cnrm = u.contact_c * fminf(1.0f, phi * u.inv_phi_ref) * (1.0f - rest);
      float dn = cnrm + u.contact_k * u.dt;
      float bias = fminf(u.contact_k * phi, u.contact_vdep * dn);
      float fn0 = bias - dn * un;
      if (fn0 > 0.f) {
        float mu = utn < u.contact_vstick ? mus : mud;
        c.act = true;
        c.x = x; c.n = nb; c.bias = bias; c.dn = dn;
        c.dt = fminf(u.contact_ct, mu * fn0 * frcp(fmaxf(utn, 1e-6f)));
      }
    }
    return c;
  }
  // spatial velocity of the link of group g from the generalised velocities (the one-lane-per-limb mapping's sensor pass)
  RL_FN SV link_twist(const ChainTP& C, int g, int s, SV V0, const float (&qdv)[JX]) const {
    SV V = V0;
    if (M0 && on_base(g, s)) return V;
    const uint32_t anc = trunk_anc(g);
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if ((anc >> i) & 1u) { V.a += qdv[CL + i] * C.axw(i); V.l += qdv[CL + i] * cross(C.pw(i), C.axw(i)); }
#pragma unroll
    for (int i = 0; i < CL; ++i)
      if (i < g) { V.a += qdv[i] * C.ax(i); V.l += qdv[i] * cross(C.p(i), C.ax(i)); }
    return V;
  }
  RL_FN Contact contact_from_patch(const ChainTP& C, const M3& Rwb, SV V0, const float (&qdv)[JX], int g, int s, float rad, V3 cb, V3 cw,
                                   const TerrainPatch& tp) const {
    float phi;
    V3 nw;
    patch_phi(tp, rad, cw, phi, nw);
    return contact_from_phi(Rwb, link_twist(C, g, s, V0, qdv), g, s, rad, cb, phi, nw);
  }

  // ================================================================== quadruped instances: articulated-body form
  // (H + A) nu+ = rhs is tree structured: every limb is a serial chain hanging off the base.  Instead of assembling the limb's
  // (6 + CL)^2 block and taking its Schur complement onto the base (the form the trunk + limbs instance still uses, below), the
  // limb is eliminated joint by joint from the tip - the articulated-body recursion applied to the velocity-level system:
  //     I^A_j = I_j + C_j + I^a_{j+1}                 6 x 6 symmetric, in BASE coordinates (no transforms between links)
  //     U = I^A_j S_j,   D = S_j^T U + d_j,   u = t_j + S_j^T rho^A_j
  //     I^a_j = I^A_j - U U^T / D,   rho^a_j = rho^A_j - U u / D
  // with I_j the link's rigid spatial inertia, C_j = dt sum P^T D_c P the implicit damping of the contacts on that link (one
  // 6 x 6 block per contact: no joint columns), rho_j = h_j - dt f_j + dt sum bias P^T n, and d_j / t_j the joint-local terms
  // (armature, implicit PD, limit spring-damper).  The base receives sum_limbs (I^a_0, rho^a_0), its 6 x 6 system is solved as
  // before and the joint velocities follow going outwards: qd_j = (u - U^T v_parent) / D.  Same linear system as the Schur form
  // (oracle/physics.py solves it densely), same solution up to round-off; a contact costs ~3x fewer operations and no
  // (6 + CL)^2 system is live in registers.
  //
  // Work split over the SUB sub-lanes of a limb: link group g = sub + SUB * it (it < NIT) - group 0 is the lane's share of the
  // base link's spheres, group j + 1 is limb link j.  The owner of a group evaluates its contacts AND builds its link's rigid
  // record, adds the two, and the 27 numbers travel to the other sub-lanes with DPP quad broadcasts when the recursion gets there.
  static constexpr bool ABA = NW == 0;  // register-resident, software-pipelined form (substeps_aba); NW > 0: substep_aba_trunk
  static constexpr int NIT = LsFor<TP, SUB>::NIT;  // ceil((CL + 1) / SUB) link groups per sub-lane (merged: ceil(CL / SUB))
  using B6 = SymIdx<6>;
  struct LinkRec {
    float A[B6::size];  // 6 x 6 symmetric, [omega; v] order
    float r[6];
  };
#ifdef RL_PK_CONTACT_ALL  // (A/B switches: the contact blocks on packed pairs in every instance / in none)
  static constexpr bool PK_CONTACT = true;
#elif defined(RL_NO_PK_CONTACT)
  static constexpr bool PK_CONTACT = false;
#else
  static constexpr bool PK_CONTACT = NW > 0;
#endif
  struct GroupFetch {
    float rad[SPL];
    V3 cb[SPL], cw[SPL];
    TerrainPatch tp[SPL];
  };

  // stage A of a link group: sphere centres, terrain loads issued (consumed by group_contacts after unrelated work)
  template <int IT>
  RL_FN bool group_fetch(const ChainTP& C, const M3& Rwb, const TerrainBase& tb, uint32_t slot_valid, GroupFetch& gf) {
    const int g = grp_of(IT);
    const bool mine = g <= CL;
    const int gi = mine ? g : CL;  // lanes without a group in this iteration index the last one and evaluate nothing
    if (!ctx.any(mine && ((slot_valid >> (gi * SPL)) & ((1u << SPL) - 1u)) != 0u)) return false;
    M3 Rg;
    V3 pg;
    group_frame(C, gi, Rg, pg);
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
      sphere_center_in(Rg, pg, Rwb, gi, s, gf.rad[s], gf.cb[s], gf.cw[s]);
      if (!mine) gf.rad[s] = -1.f;
      gf.tp[s] = terrain_fetch(u, S.terrain, tb, gf.cw[s].x, gf.cw[s].y);
    }
    return true;
  }

  // stage B: penetration, contact activation, and the contact's 6 x 6 block / bias onto the group's link record
  // `V0`, `acc0`: twist and record of the BASE link, for the base-share spheres a merged instance hosts in this group's slots
  template <int IT>
  RL_FN void group_contacts(const M3& Rwb, const SV& Vg, const GroupFetch& gf, LinkRec& acc, uint32_t& active_mask) {
    static_assert(!M0, "merged instances pass the base link's twist and record");
    group_contacts<IT>(Rwb, Vg, Vg, gf, acc, acc, active_mask);
  }
  template <int IT>
  RL_FN void group_contacts(const M3& Rwb, const SV& Vg, const SV& V0, const GroupFetch& gf, LinkRec& acc, LinkRec& acc0, uint32_t& active_mask) {
    const float dt = u.dt;
    const int g = grp_of(IT);
    const int gi = g <= CL ? g : CL;
    float phi[SPL];
    V3 nw[SPL];
    bool touching = false;
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
      patch_phi(gf.tp[s], gf.rad[s], gf.cw[s], phi[s], nw[s]);
      touching = touching || phi[s] > 0.f;
    }
    if (!ctx.any(touching)) return;  // most link groups of most wavefronts touch nothing
    // SPL predicated copies of the contact code (RL_CONTACT_LOOP: every lane walks ITS touching slots instead - the trip count is
    // the maximum over the wavefront of the touching-slot count and the code exists once; measured 3.5 us SLOWER on A1 Rough,
    // 56.6 vs 53.0 us in one gpurun call: the select chain per trip and the ballot per trip cost more than the skipped copies)
    auto one_slot = [&](const int s, const float rad_s, const V3 cb_s, const float phi_s, const V3 nw_s) __attribute__((always_inline)) {
      const bool onb = M0 && on_base(gi, s);
      SV Vs = Vg;
      if (M0) Vs = pick_sv(onb, V0, Vg);
      Contact c = contact_from_phi(Rwb, Vs, gi, s, rad_s, cb_s, phi_s, nw_s);
      if (c.act) {
        active_mask |= 1u << (gi * SPL + s);
        if constexpr (STASH_REG) {  // keep the contact for the sensor pass (s is a compile-time constant at every call: the slots are unrolled)
          float (&st)[CONTACT_WORDS] = stash_r[IT * SPL + s];
          st[0] = c.x.x; st[1] = c.x.y; st[2] = c.x.z; st[3] = c.n.x; st[4] = c.n.y; st[5] = c.n.z; st[6] = c.bias; st[7] = c.dn; st[8] = c.dt;
        } else if (STASH) {
          if constexpr (LsFor<TP, SUB>::STASH_GRAN) {
            float* st = ctx.template lane_scratch<true>() + (LS::CT + (IT * SPL + s) * STASH_SLOT_WORDS) * LSS;
            st4(st, F4{c.x.x, c.x.y, c.x.z, c.n.x});
            st4(st + 4 * LSS, F4{c.n.y, c.n.z, c.bias, c.dn});
            st4(st + 8 * LSS, F4{c.dt, 0.f, 0.f, 0.f});
          } else {
            float* st = ctx.template lane_scratch<false>() + (LS::CT + (IT * SPL + s) * CONTACT_WORDS) * LSS;
            st[0 * LSS] = c.x.x; st[1 * LSS] = c.x.y; st[2 * LSS] = c.x.z; st[3 * LSS] = c.n.x; st[4 * LSS] = c.n.y; st[5 * LSS] = c.n.z;
            st[6 * LSS] = c.bias; st[7 * LSS] = c.dn; st[8 * LSS] = c.dt;
          }
        }
        // point velocity = P [omega; v] of the link, P = [ [x]x^T | 1 ]:  dt (d_t P^T P + (d_n - d_t) g g^T), g = P^T n = [x x n; n]
        const V3 x = c.x, n = c.n;
        const float kt = dt * c.dt, kn = dt * (c.dn - c.dt), fb = dt * c.bias;
        const V3 ga = cross(x, n);
        const float g6[6] = {ga.x, ga.y, ga.z, n.x, n.y, n.z};
        const float xx = dot(x, x);
        float b6[B6::size];
#pragma unroll
        for (int i = 0; i < B6::size; ++i) b6[i] = 0.f;
        b6[B6::at(0, 0)] = kt * (xx - x.x * x.x); b6[B6::at(1, 1)] = kt * (xx - x.y * x.y); b6[B6::at(2, 2)] = kt * (xx - x.z * x.z);
        b6[B6::at(0, 1)] = -kt * x.x * x.y; b6[B6::at(0, 2)] = -kt * x.x * x.z; b6[B6::at(1, 2)] = -kt * x.y * x.z;
        b6[B6::at(0, 4)] = -kt * x.z; b6[B6::at(0, 5)] = kt * x.y;
        b6[B6::at(1, 3)] = kt * x.z; b6[B6::at(1, 5)] = -kt * x.x;
        b6[B6::at(2, 3)] = -kt * x.y; b6[B6::at(2, 4)] = kt * x.x;
        b6[B6::at(3, 3)] = kt; b6[B6::at(4, 4)] = kt; b6[B6::at(5, 5)] = kt;
        auto add_to = [&](LinkRec& d) __attribute__((always_inline)) {
#ifdef RL_PK  // the same sums pair by pair (eliminate_pk has the pairing of the packed triangle): d.A += b + (kn g) g^T, d.r += fb g
          // (not the merged instances: their two call sites - base share or link - are tail-merged into one body behind a SELECTED record
          // address, and both records leave the registers: 208 - 544 B of private memory per lane, the build's gate)
          if constexpr (!M0 && PK_CONTACT) {
          const F2p g01 = pk2(g6[0], g6[1]), g23 = pk2(g6[2], g6[3]), g45 = pk2(g6[4], g6[5]);
          const F2p k01 = pk_mul(pk1(kn), g01), k23 = pk_mul(pk1(kn), g23), k45 = pk_mul(pk1(kn), g45);
          auto up = [&](int i, float kg, F2p g, bool has_b) __attribute__((always_inline)) {
            F2p a = pk_fma(pk1(kg), g, pk2(d.A[i], d.A[i + 1]));
            if (has_b) a = pk_add(a, pk2(b6[i], b6[i + 1]));
            d.A[i] = a.x; d.A[i + 1] = a.y;
          };
          up(0, k01.x, g01, true); up(2, k01.x, g23, true); up(4, k01.x, g45, true);
          d.A[6] += b6[6] + k01.y * g6[1]; up(7, k01.y, g23, true); up(9, k01.y, g45, true);
          up(11, k23.x, g23, true); up(13, k23.x, g45, true);
          d.A[15] += b6[15] + k23.y * g6[3]; up(16, k23.y, g45, false);
          up(18, k45.x, g45, true);
          d.A[20] += b6[20] + k45.y * g6[5];
          const F2p r01 = pk_fma(pk1(fb), g01, pk2(d.r[0], d.r[1])), r23 = pk_fma(pk1(fb), g23, pk2(d.r[2], d.r[3])), r45 = pk_fma(pk1(fb), g45, pk2(d.r[4], d.r[5]));
          d.r[0] = r01.x; d.r[1] = r01.y; d.r[2] = r23.x; d.r[3] = r23.y; d.r[4] = r45.x; d.r[5] = r45.y;
          return;
          }
#endif
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            d.r[i] += fb * g6[i];
            const float kg = kn * g6[i];
#pragma unroll
            for (int jj = i; jj < 6; ++jj) d.A[B6::at(i, jj)] += b6[B6::at(i, jj)] + kg * g6[jj];
          }
        };
        if (M0 && onb) add_to(acc0);
        else add_to(acc);
      }
    };
#ifndef RL_CONTACT_LOOP
#pragma unroll
    for (int s = 0; s < SPL; ++s) one_slot(s, gf.rad[s], gf.cb[s], phi[s], nw[s]);
#else
    uint32_t tm = 0;
#pragma unroll
    for (int s = 0; s < SPL; ++s) tm |= phi[s] > 0.f ? (1u << s) : 0u;
#pragma unroll 1
    for (; ctx.any(tm != 0u); tm &= tm - 1u) {
      const int s = tm != 0u ? __builtin_ctz(tm) : 0;
      float rad_s = gf.rad[0], phi_s = tm != 0u ? phi[0] : -1.f;
      V3 cb_s = gf.cb[0], nw_s = nw[0];
#pragma unroll
      for (int i = 1; i < SPL; ++i)
        if (s == i) { rad_s = gf.rad[i]; phi_s = tm != 0u ? phi[i] : -1.f; cb_s = gf.cb[i]; nw_s = nw[i]; }
      one_slot(s, rad_s, cb_s, phi_s, nw_s);
    }
#endif
  }

  // rec += rigid spatial inertia (base coordinates) and rho = h - dt f of a body with inertia I moving with V under the bias
  // acceleration a, minus dt x an extra force `fx`
  RL_FN void add_rigid(LinkRec& rec, const SI& I, const SV& V, const SV& a, const SV& fx) const {
    const SV h = apply(I, V);
    const SV f = apply(I, a) + crf(V, h) + fx;
    const SV rho = h - f * u.dt;
    rec.A[B6::at(0, 0)] += I.I.xx; rec.A[B6::at(1, 1)] += I.I.yy; rec.A[B6::at(2, 2)] += I.I.zz;
    rec.A[B6::at(0, 1)] += I.I.xy; rec.A[B6::at(0, 2)] += I.I.xz; rec.A[B6::at(1, 2)] += I.I.yz;
    rec.A[B6::at(3, 3)] += I.m; rec.A[B6::at(4, 4)] += I.m; rec.A[B6::at(5, 5)] += I.m;
    rec.A[B6::at(0, 4)] += -I.h.z; rec.A[B6::at(0, 5)] += I.h.y;
    rec.A[B6::at(1, 3)] += I.h.z;  rec.A[B6::at(1, 5)] += -I.h.x;
    rec.A[B6::at(2, 3)] += -I.h.y; rec.A[B6::at(2, 4)] += I.h.x;
    rec.r[0] += rho.a.x; rec.r[1] += rho.a.y; rec.r[2] += rho.a.z;
    rec.r[3] += rho.l.x; rec.r[4] += rho.l.y; rec.r[5] += rho.l.z;
  }

  // rigid record of the limb link this lane owns in iteration IT (link sub + SUB * IT - 1, if the limb has it)
  template <int IT>
  RL_FN SV link_rigid(const ChainTP& C, const SV V0, const SV (&Vl)[CL], const SV (&al)[CL], LinkRec& rec) const {
    const int l = grp_of(IT) - 1;
    const bool has = l >= 0 && l < CL;
    // the link's frame / velocity / bias acceleration as a 0-1 weighted blend over the links an owner of this iteration can
    // have (a chain of selects on a per-lane index turns into an indexed load from a scratch copy of the arrays)
    M3 Rm{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    V3 pm{0.f, 0.f, 0.f};
    SV Vm{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, am = Vm;
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      if (j < G0 + SUB * IT - 1 || j > G0 + SUB * IT + SUB - 2) continue;
      const float w = l == j ? 1.f : 0.f;
      const M3 Rj = C.R(j);
      Rm.r0 += w * Rj.r0; Rm.r1 += w * Rj.r1; Rm.r2 += w * Rj.r2;
      pm += w * C.p(j);
      Vm.a += w * Vl[j].a; Vm.l += w * Vl[j].l;
      am.a += w * al[j].a; am.l += w * al[j].l;
    }
    const uint32_t li = (uint32_t)(LY.LF_INERTIA + (has ? l : 0) * INERTIA_NF);
    const float mass = has ? LF(li) : 0.f;
    const V3 cb = pm + mul(Rm, V3{LF(li + 1), LF(li + 2), LF(li + 3)});
    const SI Im = make_si(mass, cb, rotate(Rm, S3{LF(li + 4), LF(li + 5), LF(li + 6), LF(li + 7), LF(li + 8), LF(li + 9)}));
    add_rigid(rec, Im, Vm, am, SV{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}});
    return has ? Vm : V0;  // twist of the link the lane's spheres of this iteration ride on (group 0: the base link)
  }
  // the same blend for a per-joint array of twists (the new link twists of the outward pass)
  template <int IT>
  RL_FN SV pick_twist(const SV V0, const SV (&Vl)[CL]) const {
    const int l = grp_of(IT) - 1;
    SV Vm{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < CL; ++j) {
      if (j < G0 + SUB * IT - 1 || j > G0 + SUB * IT + SUB - 2) continue;
      const float w = l == j ? 1.f : 0.f;
      Vm.a += w * Vl[j].a; Vm.l += w * Vl[j].l;
    }
    return (l >= 0 && l < CL) ? Vm : V0;
  }

  RL_FN void fetch_all(const ChainTP& C, const M3& Rwb, uint32_t slot_valid, GroupFetch (&gf)[NIT], bool (&fetched)[NIT]) {
    RL_PHASE(3, "sub.contact_fetch");
#ifdef RL_ABL_NO_CONTACTS  // analysis builds (tools/ablate.sh)
    slot_valid = 0u;
#endif
    const TerrainBase tb = terrain_base(u, pos.x, pos.y);  // the root in grid coordinates, once for all the lane's spheres
    static_for<0, NIT>([&](auto it) { fetched[it.value] = group_fetch<it.value>(C, Rwb, tb, slot_valid, gf[it.value]); });
  }

  // The decimation loop of the quadruped instances, software-pipelined over the terrain loads: the kinematics of substep s + 1
  // are computed right after substep s has moved the joints, its sphere centres follow and the heightfield loads are issued -
  // they fly (L2 / MALL: 300 - 900 cycles) while the sensor timers, the actuators and the rigid link records are worked on.
  // quadrupeds, 16 lanes per env: sub-lane s of a limb also owns limb joint s's actuator and joint-local terms (actuators_owned: three or
  // four joints on four sub-lanes) - what the trunk + limbs instances do with eight.  -DRL_ACT_REPLICATED: every sub-lane every joint (A/B)
#ifdef RL_ACT_REPLICATED
  static constexpr bool ACT_OWNED = false;
#else
  static constexpr bool ACT_OWNED = NW == 0 && SUB == 4 && CL <= 4;
#endif
  RL_FN void substeps_aba(const float (&q_tgt)[JX], const float (&qd_tgt)[JX], int n) {
    const uint32_t slot_valid = (uint32_t)ctx.uniform_i((int)T.slot_valid);
    ChainTP C = new_chain();
    chain_kinematics<TP, SP>(L, q, C);
    M3 Rwb = quat_to_mat(quat);
    GroupFetch gf[NIT];
    bool fetched[NIT];
    fetch_all(C, Rwb, slot_valid, gf, fetched);
    for (int s = 0; s < n; ++s) {
      // keep the compiler from hoisting the (loop-invariant) LDS table reads of all substeps to the top of the kernel
      asm volatile("" ::: "memory");
      RL_PHASE(2, "sub.actuators+kinematics");
      float tau_e[JX], pd_diag[JX], pd_rhs[JX];
      float D_own = 0.f, uu_own = 0.f;  // (ACT_OWNED) the elimination's joint-local terms of this sub-lane's joint
      if constexpr (ACT_OWNED) actuators_owned(q_tgt, qd_tgt, tau_e, pd_diag, pd_rhs, D_own, uu_own);
      else actuators(q_tgt, qd_tgt, tau_e, pd_diag, pd_rhs);
      const SV V0{mulT(Rwb, vang), mulT(Rwb, vlin)};
      const SV a0{{0.f, 0.f, 0.f}, mulT(Rwb, V3{0.f, 0.f, u.gravity})};
      float nu0[NB], qdn[JX];
      uint32_t active_mask = 0;
      SV Vnew[NIT];
      aba_solve(C, Rwb, V0, a0, tau_e, pd_diag, pd_rhs, gf, fetched, nu0, qdn, active_mask, Vnew, D_own, uu_own);
      V3 fown[MAXOWN];
      sensor_forces(C, Rwb, V0, nu0, qdn, active_mask, Vnew, fown);
      integrate(Rwb, V0, nu0, qdn);
      if (s + 1 < n) {
        RL_PHASE(2, "sub.actuators+kinematics");
        chain_kinematics<TP, SP>(L, q, C);
        Rwb = quat_to_mat(quat);
        fetch_all(C, Rwb, slot_valid, gf, fetched);
      }
      RL_PHASE(14, "sub.sensor+integrate");
      sensor_timers(fown);
    }
  }

  // One joint of the elimination on the packed upper triangle (row r of SymIdx<6> holds columns r .. 5 in consecutive words: pairs
  // (0,1) (2,3) (4,5) of row 0, (2,3) (4,5) of rows 1 and 2, (4,5) of rows 3 and 4 start on the same column parity as the pairs of U / D)
  // as packed fp32 arithmetic (rl_math.h F2p): U = A s - every pair once down its columns, (U_c, U_c+1) += (A_rc, A_rc+1) s_r, and once
  // along its row, U_r += (A_rc, A_rc+1) . (s_c, s_c+1); D = d + s . U and u = t + s . rho as three pairs each; the rank-1 update pair by pair.
  // ~58 instead of 82 vector instructions per joint on paper, ~65 as compiled (a pair is an even-aligned register pair: some copies remain).
  RL_FN static void eliminate_pk(float (&A)[B6::size], float (&rho)[6], const float (&s)[6], float D0, float u0, float (&Uh)[6], float& ui) {
    const F2p s23 = pk2(s[2], s[3]), s45 = pk2(s[4], s[5]);
    F2p u01 = pk_mul(pk2(A[0], A[1]), pk1(s[0]));
    F2p u23 = pk_mul(pk2(A[2], A[3]), pk1(s[0]));
    F2p u45 = pk_mul(pk2(A[4], A[5]), pk1(s[0]));
    u23 = pk_fma(pk2(A[7], A[8]), pk1(s[1]), u23);
    u45 = pk_fma(pk2(A[9], A[10]), pk1(s[1]), u45);
    u23 = pk_fma(pk2(A[11], A[12]), pk1(s[2]), u23);
    u45 = pk_fma(pk2(A[13], A[14]), pk1(s[2]), u45);
    u45 = pk_fma(pk2(A[16], A[17]), pk1(s[3]), u45);
    u45 = pk_fma(pk2(A[18], A[19]), pk1(s[4]), u45);
    const F2p R0 = pk_fma(pk2(A[4], A[5]), s45, pk_mul(pk2(A[2], A[3]), s23));
    const F2p R1 = pk_fma(pk2(A[9], A[10]), s45, pk_mul(pk2(A[7], A[8]), s23));
    const F2p R2 = pk_mul(pk2(A[13], A[14]), s45);
    const F2p R3 = pk_mul(pk2(A[16], A[17]), s45);
    float U[6];
    U[0] = fmaf(A[1], s[1], u01.x) + (R0.x + R0.y);
    U[1] = fmaf(A[6], s[1], u01.y) + (R1.x + R1.y);
    U[2] = fmaf(A[12], s[3], u23.x) + (R2.x + R2.y);
    U[3] = fmaf(A[15], s[3], u23.y) + (R3.x + R3.y);
    U[4] = fmaf(A[19], s[5], u45.x);
    U[5] = fmaf(A[20], s[5], u45.y);
    const F2p s01 = pk2(s[0], s[1]), U01 = pk2(U[0], U[1]), U23 = pk2(U[2], U[3]), U45 = pk2(U[4], U[5]);
    const F2p d2 = pk_fma(U45, s45, pk_fma(U23, s23, pk_mul(U01, s01)));
    const F2p w2 = pk_fma(pk2(rho[4], rho[5]), s45, pk_fma(pk2(rho[2], rho[3]), s23, pk_mul(pk2(rho[0], rho[1]), s01)));
    const float inv = frcp(D0 + (d2.x + d2.y));
    ui = (u0 + (w2.x + w2.y)) * inv;
    const F2p h01 = pk_mul(U01, pk1(inv)), h23 = pk_mul(U23, pk1(inv)), h45 = pk_mul(U45, pk1(inv));
    Uh[0] = h01.x; Uh[1] = h01.y; Uh[2] = h23.x; Uh[3] = h23.y; Uh[4] = h45.x; Uh[5] = h45.y;
    auto down = [&](int i, F2p h, float ur) __attribute__((always_inline)) {
      const F2p a = pk_fma(pk1(-ur), h, pk2(A[i], A[i + 1]));
      A[i] = a.x; A[i + 1] = a.y;
    };
    down(0, h01, U[0]); down(2, h23, U[0]); down(4, h45, U[0]);
    A[6] = fmaf(-U[1], Uh[1], A[6]); down(7, h23, U[1]); down(9, h45, U[1]);
    down(11, h23, U[2]); down(13, h45, U[2]);
    A[15] = fmaf(-U[3], Uh[3], A[15]); down(16, h45, U[3]);
    down(18, h45, U[4]);
    A[20] = fmaf(-U[5], Uh[5], A[20]);
    const F2p r01 = pk_fma(pk1(-ui), U01, pk2(rho[0], rho[1])), r23 = pk_fma(pk1(-ui), U23, pk2(rho[2], rho[3])), r45 = pk_fma(pk1(-ui), U45, pk2(rho[4], rho[5]));
    rho[0] = r01.x; rho[1] = r01.y; rho[2] = r23.x; rho[3] = r23.y; rho[4] = r45.x; rho[5] = r45.y;
  }

  RL_FN void aba_solve(const ChainTP& C, const M3& Rwb, const SV V0, const SV a0, const float (&tau_e)[JX], const float (&pd_diag)[JX],
                       const float (&pd_rhs)[JX], const GroupFetch (&gf)[NIT], const bool (&fetched)[NIT], float (&nu0)[NB], float (&qdn)[JX],
                       uint32_t& active_mask, SV (&Vnew)[NIT], const float D_own = 0.f, const float uu_own = 0.f) {
    const float dt = u.dt;
    // ---- link velocities / bias accelerations (every sub-lane: cheap), rigid record of the owned link(s)
    RL_PHASE(4, "sub.link_records");
    SV Sj[CL], Vl[CL], al[CL];
    {
      SV Vp = V0, ap = a0;
#pragma unroll
      for (int j = 0; j < CL; ++j) {
        Sj[j] = SV{C.ax(j), cross(C.p(j), C.ax(j))};
        const SV vj = Sj[j] * qd[j];
        Vl[j] = Vp + vj;
        al[j] = ap + crm(Vl[j], vj);
        Vp = Vl[j];
        ap = al[j];
      }
    }
    LinkRec rec[NIT];
    LinkRec rec0;  // merged instances: contacts of the base-share spheres (sub-lane 0 has them); else unused
    if (M0) {
#pragma unroll
      for (int i = 0; i < B6::size; ++i) rec0.A[i] = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) rec0.r[i] = 0.f;
    }
    SV Vold[NIT];
    static_for<0, NIT>([&](auto it) {
#pragma unroll
      for (int i = 0; i < B6::size; ++i) rec[it.value].A[i] = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) rec[it.value].r[i] = 0.f;
      Vold[it.value] = link_rigid<it.value>(C, V0, Vl, al, rec[it.value]);
    });
    // ---- contacts, stage B
    RL_PHASE(5, "sub.contact_pass1");
    static_for<0, NIT>([&](auto it) {
      if (fetched[it.value]) group_contacts<it.value>(Rwb, Vold[it.value], V0, gf[it.value], rec[it.value], M0 ? rec0 : rec[it.value], active_mask);
    });
    // ---- articulated-body recursion, tip -> base (every sub-lane; the records come from their owners)
    RL_PHASE(9, "sub.aba");
    LinkRec P;  // what hangs below the current joint, as seen from its parent
#pragma unroll
    for (int i = 0; i < B6::size; ++i) P.A[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) P.r[i] = 0.f;
    float Uh[CL][6], ui[CL];
    // the joints' armature / limit words: one batch of LDS reads in front of the recursion (read joint by joint they were three round trips
    // inside a chain that has nothing else to do meanwhile: A1 35.66 -> 35.45 us, profiles/r05t_substep_batches_ab.txt)
    float jarm[CL], jlo[CL], jhi[CL];
    if constexpr (!ACT_OWNED) {
#pragma unroll
      for (int j = 0; j < CL; ++j) {
        const F4 c2 = ld4(L.jc[j] + 8), c3 = ld4(L.jc[j] + 12);
        jarm[j] = c2.z; jlo[j] = c2.w; jhi[j] = c3.x;
      }
      rl_pin(jarm); rl_pin(jlo); rl_pin(jhi);
    }
    static_for_down<CL - 1>([&](auto jc) {
      constexpr int j = jc.value, gq = j + 1 - G0, so = gq % SUB, io = gq / SUB;  // owner sub-lane / iteration of link group j + 1
#pragma unroll
      for (int i = 0; i < B6::size; ++i) P.A[i] += ctx.template leg_bcast<so>(rec[io].A[i]);
#pragma unroll
      for (int i = 0; i < 6; ++i) P.r[i] += ctx.template leg_bcast<so>(rec[io].r[i]);
      const float s6[6] = {Sj[j].a.x, Sj[j].a.y, Sj[j].a.z, Sj[j].l.x, Sj[j].l.y, Sj[j].l.z};
      // joint-local terms: armature, implicit PD, limit spring-damper (hard limits in the reference, a1.urdf:369,411,439)
      float D, uu;
      if constexpr (ACT_OWNED) {  // from the sub-lane that owns joint j (actuators_owned: the same expressions, once per limb)
        D = ctx.template leg_bcast<j>(D_own);
        uu = ctx.template leg_bcast<j>(uu_own);
      } else {
        const float arm = jarm[j];
        const float below = jlo[j] - q[j], above = q[j] - jhi[j];
        const float viol = below > 0.f ? below : (above > 0.f ? -above : 0.f);
        const bool lim = (below > 0.f) || (above > 0.f);
        D = arm + pd_diag[j] + (lim ? dt * (u.limit_k * dt + u.limit_c) : 0.f);
        if (TP::PAD) D += j >= L.nj ? 1.0f : 0.f;  // an inert padding joint (zero axis, no gains): the identity row, as joint_terms() of the trunk + limbs instances
        uu = arm * qd[j] + dt * tau_e[j] + pd_rhs[j] + dt * u.limit_k * viol;
      }
#ifdef RL_PK
      eliminate_pk(P.A, P.r, s6, D, uu, Uh[j], ui[j]);
#else
      float U6[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) t += P.A[B6::at(r, c)] * s6[c];
        U6[r] = t;
        D += s6[r] * t;
        uu += s6[r] * P.r[r];
      }
      const float inv = frcp(D);
      ui[j] = uu * inv;
#pragma unroll
      for (int r = 0; r < 6; ++r) Uh[j][r] = U6[r] * inv;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        P.r[r] -= U6[r] * ui[j];
#pragma unroll
        for (int c = r; c < 6; ++c) P.A[B6::at(r, c)] -= U6[r] * Uh[j][c];
      }
#endif
    });
    // the lane's share of the base link's contacts (group 0, owned by sub-lane 0 in iteration 0), when anybody has one
    if (M0) {  // flagged slots of any sub-lane
      if (ctx.any((active_mask & base_bits()) != 0u)) {
#pragma unroll
        for (int i = 0; i < B6::size; ++i) P.A[i] += ctx.leg_sum(rec0.A[i]);
#pragma unroll
        for (int i = 0; i < 6; ++i) P.r[i] += ctx.leg_sum(rec0.r[i]);
      }
    } else if (ctx.any(sub == 0 && (active_mask & base_bits()) != 0u)) {
#pragma unroll
      for (int i = 0; i < B6::size; ++i) P.A[i] += ctx.template leg_bcast<0>(rec[0].A[i]);
#pragma unroll
      for (int i = 0; i < 6; ++i) P.r[i] += ctx.template leg_bcast<0>(rec[0].r[i]);
    }
    if (k == 0) {  // the base link itself and the persistent external wrench [UPSTREAM B8]: rides with limb 0
      const int bi = LY.EF_BASE_INERTIA;
      const SI I0 = make_si(EF(bi), V3{EF(bi + 1), EF(bi + 2), EF(bi + 3)}, S3{EF(bi + 4), EF(bi + 5), EF(bi + 6), EF(bi + 7), EF(bi + 8), EF(bi + 9)});
      add_rigid(P, I0, V0, a0, SV{-(extT + cross(base_com, extF)), -extF});
    }
    // ---- cross-limb reduction, 6 x 6 base solve
    RL_PHASE(10, "sub.cross_leg_sum");
    float Cb[B6::size], db[6];
#pragma unroll
    for (int i = 0; i < B6::size; ++i) Cb[i] = ctx.gsum(P.A[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) db[i] = ctx.gsum(P.r[i]);
    RL_PHASE(11, "sub.trunk_solve");
    {
      float G[6][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float sacc = Cb[B6::at(j, j)];
#pragma unroll
        for (int m = 0; m < j; ++m) sacc -= G[j][m] * G[j][m];
        const float inv = frsqrt(sacc);
        G[j][j] = inv;  // 1 / G_jj
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
          float t = Cb[B6::at(j, i)];
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
        y6[j] = t * G[j][j];
      }
#pragma unroll
      for (int j = 5; j >= 0; --j) {
        float t = y6[j];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) t -= G[i][j] * nu0[i];
        nu0[j] = t * G[j][j];
      }
    }
    // ---- joint velocities, base -> tip
    RL_PHASE(12, "sub.back_subst");
    {
      float vp[6] = {nu0[0], nu0[1], nu0[2], nu0[3], nu0[4], nu0[5]};
      // new link twists for the contact sensor: built from the velocity-limited joint velocities (the limit is applied to the
      // solution, then the forces are evaluated - oracle/physics.py), while the recursion itself runs on the solution
      SV vc{{nu0[0], nu0[1], nu0[2]}, {nu0[3], nu0[4], nu0[5]}};
      SV Vn[CL];
#pragma unroll
      for (int j = 0; j < CL; ++j) {
        float t = ui[j];
#pragma unroll
        for (int r = 0; r < 6; ++r) t -= Uh[j][r] * vp[r];
        qdn[j] = t;
        vp[0] += Sj[j].a.x * t; vp[1] += Sj[j].a.y * t; vp[2] += Sj[j].a.z * t;
        vp[3] += Sj[j].l.x * t; vp[4] += Sj[j].l.y * t; vp[5] += Sj[j].l.z * t;
        const float tc = clampf(t, -L.vel_limit[j], L.vel_limit[j]);
        vc.a += tc * Sj[j].a; vc.l += tc * Sj[j].l;
        Vn[j] = vc;
      }
      const SV V0n{{nu0[0], nu0[1], nu0[2]}, {nu0[3], nu0[4], nu0[5]}};
      static_for<0, NIT>([&](auto it) { Vnew[it.value] = pick_twist<it.value>(V0n, Vn); });
    }
  }

  // ================================================================== trunk + limbs instance: articulated-body form
  // Same recursion as substeps_aba, for a base that carries NW serial trunk joints (G1: the waist) with limbs hanging off the base
  // (legs) and off the last trunk link (arms), and 7-joint limbs.  Nothing of that size stays in registers: the kinematics, the
  // link records and the per-joint elimination results live in limb-shared LDS words (the 4 sub-lanes of a limb hold identical
  // values: same word, a broadcast), and the limbs meet in 27-word per-trunk-link accumulators of the ENV (ds_add_f32), which
  // every lane then reads to eliminate the trunk joints and solve the 6 x 6 base system redundantly.
  RL_FN float* rec_words(int g) const { return ctx.limb_rec() + LB_REC + (g - 1) * REC_STRIDE; }  // link group g = limb link g - 1 (1 .. CL)
  RL_FN float* va_words(int j) const { return ctx.limb_rec() + LB_VA + j * 12; }
  RL_FN float* trunk_words(int d) const { return ctx.env_scratch() + d * REC_STRIDE; }
  // a link record <-> LDS (REC_ROWS: six rows of [A[r][0..5], rho_r, 0]; else seven vectors of the packed upper triangle + rho)
  RL_FN static void st_rec(float* w, const LinkRec& r) {
    if constexpr (REC_ROWS) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        st4(w + 8 * q, F4{r.A[B6::at(q, 0)], r.A[B6::at(q, 1)], r.A[B6::at(q, 2)], r.A[B6::at(q, 3)]});
        st4(w + 8 * q + 4, F4{r.A[B6::at(q, 4)], r.A[B6::at(q, 5)], r.r[q], 0.f});
      }
    } else {  // seven vectors in the order of rec_pos_A / rec_pos_r: every 8-byte half is one pair of the packed arithmetic
      st4(w, F4{r.A[0], r.A[1], r.A[2], r.A[3]});
      st4(w + 4, F4{r.A[4], r.A[5], r.A[7], r.A[8]});
      st4(w + 8, F4{r.A[9], r.A[10], r.A[11], r.A[12]});
      st4(w + 12, F4{r.A[13], r.A[14], r.A[16], r.A[17]});
      st4(w + 16, F4{r.A[18], r.A[19], r.A[6], r.A[15]});
      st4(w + 20, F4{r.r[0], r.r[1], r.r[2], r.r[3]});
      st4(w + 24, F4{r.r[4], r.r[5], r.A[20], 0.f});
    }
  }
  // word of a packed record in LDS that holds entry i of the upper triangle / rho_r: the pairs eliminate_pk works on - (0,1) (2,3) (4,5)
  // (7,8) (9,10) (11,12) (13,14) (16,17) (18,19), rho (0,1) (2,3) (4,5) - are the 8-byte halves of the record's 16-byte vectors, so
  // that add_rec is 13 packed additions on the loaded register pairs as they are (+ the three lone diagonal entries 6, 15, 20)
  static constexpr int rec_pos_A(int i) { return i <= 5 ? i : i == 6 ? 18 : i <= 14 ? i - 1 : i == 15 ? 19 : i <= 19 ? i - 2 : 26; }
  static constexpr int rec_pos_r(int r) { return 20 + r; }
  RL_FN static void add_rec(const float* w, LinkRec& P) {  // (the replicated elimination: the whole record into every lane)
    if constexpr (REC_ROWS) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const F4 a = ld4(w + 8 * q), c = ld4(w + 8 * q + 4);
        const float row[6] = {a.x, a.y, a.z, a.w, c.x, c.y};
#pragma unroll
        for (int cc = q; cc < 6; ++cc) P.A[B6::at(q, cc)] += row[cc];
        P.r[q] += c.z;
      }
    } else {
      F4 v[7];
      ld_rec(w, v);
      acc_rec(v, P);
    }
  }
  // the two halves of add_rec for the packed layout: the record's seven vectors into registers / onto P (the limb elimination issues
  // the loads of the NEXT joint's record before it eliminates the current one - substep_aba_trunk)
#ifdef RL_NO_REC_PREFETCH
  static constexpr bool REC_PREFETCH = false;
#else
  static constexpr bool REC_PREFETCH = !REC_ROWS;
#endif
  RL_FN static void ld_rec(const float* w, F4 (&v)[7]) {
#pragma unroll
    for (int i = 0; i < 7; ++i) v[i] = ld4(w + 4 * i);
  }
  RL_FN static void acc_rec(const F4 (&v)[7], LinkRec& P) {
    {
      auto acc = [&](float& a, float& b, float x, float y) __attribute__((always_inline)) {
#ifdef RL_PK
        const F2p t = pk_add(pk2(a, b), pk2(x, y));
        a = t.x; b = t.y;
#else
        a += x; b += y;
#endif
      };
      acc(P.A[0], P.A[1], v[0].x, v[0].y); acc(P.A[2], P.A[3], v[0].z, v[0].w);
      acc(P.A[4], P.A[5], v[1].x, v[1].y); acc(P.A[7], P.A[8], v[1].z, v[1].w);
      acc(P.A[9], P.A[10], v[2].x, v[2].y); acc(P.A[11], P.A[12], v[2].z, v[2].w);
      acc(P.A[13], P.A[14], v[3].x, v[3].y); acc(P.A[16], P.A[17], v[3].z, v[3].w);
      acc(P.A[18], P.A[19], v[4].x, v[4].y);
      P.A[6] += v[4].z; P.A[15] += v[4].w;
      acc(P.r[0], P.r[1], v[5].x, v[5].y); acc(P.r[2], P.r[3], v[5].z, v[5].w);
      acc(P.r[4], P.r[5], v[6].x, v[6].y);
      P.A[20] += v[6].z;
    }
  }
  RL_FN static float* rho_word(float* w, int r) { return REC_ROWS ? w + 8 * r + 6 : w + rec_pos_r(r); }  // rho_r of a record at w
  RL_FN static void atomic_add_rec(float* w, const LinkRec& r) {  // ds_add_f32 of a whole record (several lanes of an env add into one)
    if constexpr (REC_ROWS) {
#pragma unroll
      for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int c = 0; c < 6; ++c) Ctx::limb_atomic_add(w + 8 * q + c, r.A[B6::at(q, c)]);
        Ctx::limb_atomic_add(w + 8 * q + 6, r.r[q]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < B6::size; ++i) Ctx::limb_atomic_add(w + rec_pos_A(i), r.A[i]);
#pragma unroll
      for (int i = 0; i < 6; ++i) Ctx::limb_atomic_add(w + rec_pos_r(i), r.r[i]);
    }
  }

  // ---- DISTRIBUTED elimination (trunk + limbs instances, SUB >= 4): the four lanes of a DPP quad hold identical copies of everything the
  // limb recursion needs, so they SHARE a joint's 6 x 6 work by rows instead of each repeating it: quad lane q carries row q of the
  // articulated inertia (and row q + 4 where q < 2; the other two lanes a zero phantom row) and its rho entries.  Per joint: the lane's
  // one or two rows of the link record from LDS (2 - 4 vectors instead of 7), U_r = P_r . s (12 FMA instead of 36), D and u as two quad
  // sums, U / D gathered into every lane with six quad broadcasts, the rank-1 update of the lane's rows (14 FMA instead of 27).
  // ~60 vector instructions per joint instead of ~140; same linear system, the sums in tree order.
  static constexpr bool ELIM_DIST = REC_ROWS && SUB >= 4 && NW > 0;
  struct Rows {
    float a[6], ra, b[6], rb;
  };
  RL_FN static void rows_zero(Rows& P) {
#pragma unroll
    for (int c = 0; c < 6; ++c) P.a[c] = P.b[c] = 0.f;
    P.ra = P.rb = 0.f;
  }
  RL_FN void rows_add(const float* w, Rows& P) const {
    const int q = sub & 3, rb = q < 2 ? q + 4 : 5;
    const float wb = q < 2 ? 1.f : 0.f;
    const F4 a0 = ld4(w + 8 * q), a1 = ld4(w + 8 * q + 4), b0 = ld4(w + 8 * rb), b1 = ld4(w + 8 * rb + 4);
    P.a[0] += a0.x; P.a[1] += a0.y; P.a[2] += a0.z; P.a[3] += a0.w; P.a[4] += a1.x; P.a[5] += a1.y; P.ra += a1.z;
    P.b[0] += wb * b0.x; P.b[1] += wb * b0.y; P.b[2] += wb * b0.z; P.b[3] += wb * b0.w; P.b[4] += wb * b1.x; P.b[5] += wb * b1.y; P.rb += wb * b1.z;
  }
  RL_FN static void rows_acc(Rows& P, const Rows& Q) {
#pragma unroll
    for (int c = 0; c < 6; ++c) { P.a[c] += Q.a[c]; P.b[c] += Q.b[c]; }
    P.ra += Q.ra; P.rb += Q.rb;
  }
  RL_FN void eliminate_rows(Rows& P, const float (&s6)[6], float D0, float u0, float (&Uh)[6], float& ui) const {
    const int q = sub & 3;
    float Ua = 0.f, Ub = 0.f;
#pragma unroll
    for (int c = 0; c < 6; ++c) { Ua += P.a[c] * s6[c]; Ub += P.b[c] * s6[c]; }
    // the lane's own entries of s: a 0/1-weighted sum (a select chain on q is lowered to a jump table of exec-masked blocks)
    const float e0 = q == 0 ? 1.f : 0.f, e1 = q == 1 ? 1.f : 0.f, e2 = q == 2 ? 1.f : 0.f, e3 = q == 3 ? 1.f : 0.f;
    const float sa = e0 * s6[0] + e1 * s6[1] + e2 * s6[2] + e3 * s6[3], sb = e0 * s6[4] + e1 * s6[5];
    const float D = D0 + ctx.quad_sum(sa * Ua + sb * Ub), uu = u0 + ctx.quad_sum(sa * P.ra + sb * P.rb);
    const float inv = frcp(D);
    ui = uu * inv;
    const float Uha = Ua * inv, Uhb = Ub * inv;
    Uh[0] = ctx.template quad_bcast<0>(Uha); Uh[1] = ctx.template quad_bcast<1>(Uha); Uh[2] = ctx.template quad_bcast<2>(Uha);
    Uh[3] = ctx.template quad_bcast<3>(Uha); Uh[4] = ctx.template quad_bcast<0>(Uhb); Uh[5] = ctx.template quad_bcast<1>(Uhb);
#pragma unroll
    for (int c = 0; c < 6; ++c) { P.a[c] -= Ua * Uh[c]; P.b[c] -= Ub * Uh[c]; }
    P.ra -= Ua * ui;
    P.rb -= Ub * ui;
  }
  RL_FN void rows_atomic_add(float* w, const Rows& P) const {  // the lane's rows into an accumulator record (every row by exactly one lane)
    const int q = sub & 3;
#pragma unroll
    for (int c = 0; c < 6; ++c) Ctx::limb_atomic_add(w + 8 * q + c, P.a[c]);
    Ctx::limb_atomic_add(w + 8 * q + 6, P.ra);
    if (q < 2) {
#pragma unroll
      for (int c = 0; c < 6; ++c) Ctx::limb_atomic_add(w + 8 * (q + 4) + c, P.b[c]);
      Ctx::limb_atomic_add(w + 8 * (q + 4) + 6, P.rb);
    }
  }
  RL_FN void rows_gather(const Rows& P, float (&A)[B6::size], float (&r)[6]) const {  // the whole (symmetric) system into every lane of the quad
    static_for<0, 6>([&](auto rc) __attribute__((always_inline)) {
      constexpr int R = decltype(rc)::value, Q = R & 3;
      static_for<R, 6>([&](auto cc) __attribute__((always_inline)) {
        constexpr int Cc = decltype(cc)::value;
        A[B6::at(R, Cc)] = ctx.template quad_bcast<Q>(R < 4 ? P.a[Cc] : P.b[Cc]);
      });
      r[R] = ctx.template quad_bcast<Q>(R < 4 ? P.ra : P.rb);
    });
  }
  RL_FN static SV ld_sv(const float* w) {  // six words, the first two vectors of a record / per-joint block
    const F4 a = ld4(w), c = ld4(w + 4);
    return SV{{a.x, a.y, a.z}, {a.w, c.x, c.y}};
  }

  // ------------------------------------------------------------------ self-collision (trunk + limbs instance)
  // enabled_self_collisions of the reference's ArticulationCfg (assets/unitree.py:482 G1, assets/roboparty.py:33 ATOM01): up to 16
  // links carry a capsule, the listed pairs repel with an explicit penalty force k * penetration along the line between the
  // segments' closest points, once per substep, at the positions of the start of the substep (oracle/physics.py does the same).
  // Dealt to the env's 16 virtual lanes (env_tables.h SelfLaneTab): each places at most one capsule - centre and half axis in base
  // coordinates into 8 env-shared words, ahead of the group_sync the link records need anyway - and, once the records are
  // complete, tests at most five pairs; a hit ds_adds dt * [x cross F; F] into the bias of the two links' records.
  static constexpr int SELF_V = SUB >= 4 ? 1 : 4 / SUB;  // virtual lanes this lane plays: v = 4 k + sub + SUB * i (eight sub-lanes per limb: the first four play one each)
  SelfLaneTab self_tab[NW > 0 ? SELF_V : 1];
  RL_FN bool self_on() const { return NW > 0 && S.self_k > 0.f; }
  RL_FN float* cap_words(int c) const { return ctx.env_scratch() + (NW + 1) * REC_STRIDE + c * SELF_CAP_WORDS; }
  RL_FN void self_load() {  // once per launch: this lane's share of the dealing, from the table image in HBM into registers
    if constexpr (NW > 0) {
      if (!self_on()) return;
      const auto& Tg = ctx.template tables_global<TablesT<TP>>();
#pragma unroll
      for (int i = 0; i < SELF_V; ++i) {
        self_tab[i] = Tg.self_lane[4 * k + ((sub + SUB * i) & 3)];
        if (SUB > 4 && sub >= 4) {  // no virtual lane of its own: nothing to place, no pair to test
          self_tab[i].cap = -1;
#pragma unroll
          for (int p = 0; p < SELF_PPL; ++p) self_tab[i].pair[p] = -1;
        }
      }
    }
  }
  RL_FN void self_place(const ChainTP& C) {
    if constexpr (NW > 0) {
#pragma unroll
      for (int i = 0; i < SELF_V; ++i) {
        const SelfLaneTab& t = self_tab[i];
        if (t.cap < 0) continue;
        M3 Rf = identity3();
        V3 pf{0.f, 0.f, 0.f};
        if (t.frame >= 0) C.frame(t.frame, Rf, pf);
        const V3 p0 = pf + mul(Rf, ld3(t.p0)), p1 = pf + mul(Rf, ld3(t.p1));
        const V3 c = 0.5f * (p0 + p1), h = 0.5f * (p1 - p0);
        float* w = cap_words(t.cap);  // centre, half axis, radius, radius of the bounding sphere
        st4(w, F4{c.x, c.y, c.z, h.x});
        st4(w + 4, F4{h.y, h.z, t.r, t.r + fsqrt(dot(h, h))});
      }
    }
  }
  RL_FN void self_add(int kk, int gg, V3 x, V3 F) {  // dt * [x cross F; F] onto the bias of a link record (limb kk's group gg; kk = 7: trunk depth gg)
    const V3 m = u.dt * cross(x, F), f = u.dt * F;
    const float v6[6] = {m.x, m.y, m.z, f.x, f.y, f.z};
    float* w = kk == 7 ? trunk_words(gg) : ctx.limb_rec_of(kk) + LB_REC + (gg - 1) * REC_STRIDE;
#pragma unroll
    for (int i = 0; i < 6; ++i) Ctx::limb_atomic_add(rho_word(w, i), v6[i]);
  }
  RL_FN void self_apply() {
    if constexpr (NW > 0) {
      // bounding spheres of all this lane's pairs first (one batch of LDS reads, no vote in between): the host deals the pairs
      // closest-in-the-default-pose first, so the later trips hold pairs that are far apart in every env of the wavefront
      bool near[SELF_V * SELF_PPL];
#pragma unroll
      for (int i = 0; i < SELF_V; ++i)
#pragma unroll
        for (int p = 0; p < SELF_PPL; ++p) {
          near[i * SELF_PPL + p] = false;
          if (p >= S.self_trips) continue;  // (uniform: the model has fewer pairs than slots)
          const int wd = self_tab[i].pair[p];
          const float* wa = cap_words(wd >= 0 ? (wd & 15) : 0);
          const float* wb = cap_words(wd >= 0 ? ((wd >> 4) & 15) : 0);
          const F4 a0 = ld4(wa), a1 = ld4(wa + 4), b0 = ld4(wb), b1 = ld4(wb + 4);
          const V3 dc = V3{a0.x, a0.y, a0.z} - V3{b0.x, b0.y, b0.z};
          const float rb = a1.w + b1.w;
          near[i * SELF_PPL + p] = wd >= 0 && dot(dc, dc) < rb * rb;
        }
#pragma unroll
      for (int i = 0; i < SELF_V; ++i)
#pragma unroll
        for (int p = 0; p < SELF_PPL; ++p) {
          if (p >= S.self_trips || !ctx.any(near[i * SELF_PPL + p])) continue;
          const int wd = self_tab[i].pair[p];
          const float* wa = cap_words(wd >= 0 ? (wd & 15) : 0);
          const float* wb = cap_words(wd >= 0 ? ((wd >> 4) & 15) : 0);
          const F4 a0 = ld4(wa), a1 = ld4(wa + 4), b0 = ld4(wb), b1 = ld4(wb + 4);
          const V3 ca{a0.x, a0.y, a0.z}, cb{b0.x, b0.y, b0.z}, ha{a0.w, a1.x, a1.y}, hb{b0.w, b1.x, b1.y};
          V3 xa, xb;
          segment_closest(ca - ha, ca + ha, cb - hb, cb + hb, xa, xb);
          const V3 dv = xa - xb;
          const float d2 = dot(dv, dv), rr = a1.z + b1.z;
          const bool hit = near[i * SELF_PPL + p] && d2 < rr * rr;
          if (!ctx.any(hit)) continue;  // the usual case: nothing of this trip touches anywhere in the wavefront
          if (hit) {
            const float dist = fsqrt(d2);
            const V3 F = (S.self_k * (rr - dist) * frcp(fmaxf(dist, 1e-9f))) * dv;  // on a; -F on b
            self_add((wd >> 8) & 7, (wd >> 11) & 15, xa, F);
            self_add((wd >> 15) & 7, (wd >> 18) & 15, xb, V3{-F.x, -F.y, -F.z});
          }
        }
    }
  }

  // twist of trunk link `depth` (0 = the base: `base`; i + 1 = behind trunk joint i: arr[i]) for a per-lane depth.  A 0/1-weighted
  // sum, not a select chain: hipcc turns the chain into an indexed load from a SCRATCH copy of the array (112 bytes of private
  // memory per lane and 48 scratch instructions on the G1 instance).
  RL_FN SV pick_trunk(const SV& base, const SV (&arr)[NW > 0 ? NW : 1], int depth) const {
    float w0 = depth == 0 ? 1.f : 0.f;
    SV r{w0 * base.a, w0 * base.l};
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const float w = depth == i + 1 ? 1.f : 0.f;
      r.a += w * arr[i].a;
      r.l += w * arr[i].l;
    }
    return r;
  }

  // P -= (P S)(P S)^T / D etc. for one joint with motion subspace s6 and joint-local terms (D0, u0); returns U / D and u / D
  RL_FN void eliminate(LinkRec& P, const float (&s6)[6], float D, float uu, float (&Uh)[6], float& ui) const {
#ifdef RL_PK
    eliminate_pk(P.A, P.r, s6, D, uu, Uh, ui);
    return;
#endif
    float U6[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c) t += P.A[B6::at(r, c)] * s6[c];
      U6[r] = t;
      D += s6[r] * t;
      uu += s6[r] * P.r[r];
    }
    const float inv = frcp(D);
    ui = uu * inv;
#pragma unroll
    for (int r = 0; r < 6; ++r) Uh[r] = U6[r] * inv;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      P.r[r] -= U6[r] * ui;
#pragma unroll
      for (int c = r; c < 6; ++c) P.A[B6::at(r, c)] -= U6[r] * Uh[c];
    }
  }
  // joint-local terms of joint jx (limb joint j or trunk joint CL + i): armature, implicit PD, limit spring-damper, and the
  // identity row of an inert padding joint
  RL_FN void joint_terms(int jx, bool padding, const float (&tau_e)[JX], const float (&pd_diag)[JX], const float (&pd_rhs)[JX], float& D, float& uu) const {
    joint_terms_one(jx, q[jx], qd[jx], padding, tau_e[jx], pd_diag[jx], pd_rhs[jx], D, uu);
  }
  RL_FN void joint_terms_one(int jx, float qj, float qdj, bool padding, float t_e, float p_diag, float p_rhs, float& D, float& uu) const {
    const F4 c2 = ld4(L.jc[jx] + 8), c3 = ld4(L.jc[jx] + 12);  // [. . armature lower | upper . . .]
    const float dt = u.dt, arm = c2.z;
    const float below = c2.w - qj, above = qj - c3.x;
    const float viol = below > 0.f ? below : (above > 0.f ? -above : 0.f);
    const bool lim = (below > 0.f) || (above > 0.f);
    D = arm + p_diag + (lim ? dt * (u.limit_k * dt + u.limit_c) : 0.f) + (padding ? 1.0f : 0.f);
    uu = arm * qdj + dt * t_e + p_rhs + dt * u.limit_k * viol;
  }
  // Eight sub-lanes per limb (KIN_SCAN): sub-lane s is also the one that evaluates limb joint s's ACTUATOR and joint-local terms (D, u of
  // the elimination); the recursion fetches them with a limb broadcast per joint, the applied torques go back to every sub-lane (the reward
  // stage and the write-back read their own copies).  The trunk joints stay in every lane.  Before: every sub-lane all ten joints - 500 + 105
  // vector instructions per substep against 55 + 165 + 63 here.  Same arithmetic per joint, same bits.
  RL_FN void actuators_owned(const float (&q_tgt)[JX], const float (&qd_tgt)[JX], float (&tau_e)[JX], float (&pd_diag)[JX], float (&pd_rhs)[JX], float& D_own,
                             float& uu_own) {
    static_assert((NW > 0 && SUB == 8) || (NW == 0 && SUB == 4 && CL <= 4), "a limb joint per sub-lane");
#pragma unroll
    for (int j = CL; j < JX; ++j) {
      const F4 c1 = ld4(L.jc[j] + 4), c2 = ld4(L.jc[j] + 8);
      actuator_one(q[j], qd[j], kp[j], kd[j], q_tgt[j], qd_tgt[j], c1.z, c1.w, c2.x, (int)c2.y, tau_app[j], tau_e[j], pd_diag[j], pd_rhs[j]);
    }
    const int js = sub < CL ? sub : CL - 1;
    float qs = opaque(q[0]), qds = opaque(qd[0]), kps = opaque(kp[0]), kds = opaque(kd[0]), qts = opaque(q_tgt[0]), qdts = opaque(qd_tgt[0]);
    static_for<1, CL>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      const float c0 = opaque(q[j]), c1 = opaque(qd[j]), c2 = opaque(kp[j]), c3 = opaque(kd[j]), c4 = opaque(q_tgt[j]), c5 = opaque(qd_tgt[j]);
      const bool m = js == j;
      qs = m ? c0 : qs; qds = m ? c1 : qds; kps = m ? c2 : kps; kds = m ? c3 : kds; qts = m ? c4 : qts; qdts = m ? c5 : qdts;
    });
    const F4 c1 = ld4(L.jc[js] + 4), c2 = ld4(L.jc[js] + 8);
    float ta, te, pdg, prh;
    actuator_one(qs, qds, kps, kds, qts, qdts, c1.z, c1.w, c2.x, (int)c2.y, ta, te, pdg, prh);
    joint_terms_one(js, qs, qds, TP::PAD && js >= L.nj, te, pdg, prh, D_own, uu_own);
    static_for<0, CL>([&](auto jc) __attribute__((always_inline)) {
      constexpr int j = decltype(jc)::value;
      this->tau_app[j] = this->ctx.template leg_bcast<j>(ta);
    });
  }

  RL_FN void substep_aba_trunk(const float (&q_tgt)[JX], const float (&qd_tgt)[JX]) {
    static_assert(NW > 0, "trunk + limbs instance");
    asm volatile("" ::: "memory");
    RL_PHASE(2, "sub.actuators+kinematics");
    const float dt = u.dt;
    float tau_e[JX], pd_diag[JX], pd_rhs[JX];
    float D_own = 0.f, uu_own = 0.f;  // (KIN_SCAN) the elimination's joint-local terms of this sub-lane's limb joint
    if constexpr (KIN_SCAN) actuators_owned(q_tgt, qd_tgt, tau_e, pd_diag, pd_rhs, D_own, uu_own);
    else actuators(q_tgt, qd_tgt, tau_e, pd_diag, pd_rhs);
    const M3 Rwb = quat_to_mat(quat);
    const SV V0{mulT(Rwb, vang), mulT(Rwb, vlin)};
    const SV a0{{0.f, 0.f, 0.f}, mulT(Rwb, V3{0.f, 0.f, u.gravity})};
    ChainTP C = new_chain();
    // Kinematics, and - while a joint's axis and origin are in registers - the motion subspaces, link velocities and bias accelerations
    // along the chain (trunk joints: redundant in all lanes, kept in registers; limb joints: identical in the limb's sub-lanes, parked
    // in the limb-shared words for the owners of the link groups).  RL_VEL_SEPARATE: the round-3 form, a second pass that reads the
    // chain words back (A/B).
    SV Sw[NW], Vw[NW], aw[NW];
#ifndef RL_VEL_SEPARATE
    if constexpr (KIN_SCAN) {
      kinematics_scan<true>(C, V0, a0, Sw, Vw, aw);
      if (self_on()) ctx.group_sync();  // (self_place reads the limb's chain words: other sub-lanes wrote them)
    } else {
      SV Vp = V0, ap = a0;
      kinematics(C,
                 [&](int i, V3 ax, V3 pj) __attribute__((always_inline)) {
                   if (i > 0 && ((u.trunk_restart >> i) & 1u)) { Vp = V0; ap = a0; }  // a trunk piece that starts at the base
                   Sw[i] = SV{ax, cross(pj, ax)};
                   const SV vj = Sw[i] * qd[CL + i];
                   Vw[i] = Vp + vj;
                   aw[i] = ap + crm(Vw[i], vj);
                   Vp = Vw[i];
                   ap = aw[i];
                   if (i == NW - 1) {  // the limb starts from the link it hangs off
                     Vp = pick_trunk(V0, Vw, L.attach);
                     ap = pick_trunk(a0, aw, L.attach);
                   }
                 },
                 [&](int j, V3 ax, V3 pj) __attribute__((always_inline)) {
                   const SV Sj{ax, cross(pj, ax)};
                   const SV vj = Sj * qd[j];
                   const SV Vj = Vp + vj;
                   const SV aj = ap + crm(Vj, vj);
                   float* w = va_words(j);
                   st4(w, F4{Vj.a.x, Vj.a.y, Vj.a.z, Vj.l.x});
                   st4(w + 4, F4{Vj.l.y, Vj.l.z, aj.a.x, aj.a.y});
                   st4(w + 8, F4{aj.a.z, aj.l.x, aj.l.y, aj.l.z});
                   Vp = Vj;
                   ap = aj;
                 });
    }
    if (self_on()) self_place(C);  // (a lane reads frames of its own limb's words only: written by itself, identically in every sub-lane)
#else
    kinematics(C);
    if (self_on()) self_place(C);  // (a lane reads frames of its own limb's words only: written by itself, identically in every sub-lane)
    // trunk joints: motion subspaces, link velocities and bias accelerations (redundant in all lanes)
    {
      SV Vp = V0, ap = a0;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        V3 axi, pi;
        C.axp(CL + i, axi, pi);
        if (i > 0 && ((u.trunk_restart >> i) & 1u)) { Vp = V0; ap = a0; }
        Sw[i] = SV{axi, cross(pi, axi)};
        const SV vj = Sw[i] * qd[CL + i];
        Vw[i] = Vp + vj;
        aw[i] = ap + crm(Vw[i], vj);
        Vp = Vw[i];
        ap = aw[i];
      }
    }
    // limb link velocities / bias accelerations -> limb-shared words (identical in the 4 sub-lanes); env accumulators cleared
    {
      SV Vp = V0, ap = a0;
#pragma unroll
      for (int i = 0; i < NW; ++i)
        if (L.attach == i + 1) { Vp = Vw[i]; ap = aw[i]; }
#pragma unroll
      for (int j = 0; j < CL; ++j) {
        V3 ax, pj;
        C.axp(j, ax, pj);
        const SV Sj{ax, cross(pj, ax)};
        const SV vj = Sj * qd[j];
        const SV Vj = Vp + vj;
        const SV aj = ap + crm(Vj, vj);
        float* w = va_words(j);
        st4(w, F4{Vj.a.x, Vj.a.y, Vj.a.z, Vj.l.x});
        st4(w + 4, F4{Vj.l.y, Vj.l.z, aj.a.x, aj.a.y});
        st4(w + 8, F4{aj.a.z, aj.l.x, aj.l.y, aj.l.z});
        Vp = Vj;
        ap = aj;
      }
    }
#endif
    // The env's accumulator records (one per trunk link, 0 = the base link) start as the trunk links' own rigid records - plain stores,
    // no clearing pass: the first sub-lane of lane group k owns the trunk links k, k + 4 (a spine of more than three joints has more
    // trunk links than limbs); the persistent external wrench [UPSTREAM B8] rides on its link's record.  Contacts of trunk-link spheres,
    // self-collision forces and the eliminated limbs are ADDED behind the fence below.
    if (sub == 0) {
#pragma unroll
      for (int tq = 0; tq < (NW + NLANE) / NLANE; ++tq) {
        const int tk = k + NLANE * tq;
        if (tk > NW) continue;
        const int bi = LY.EF_BASE_INERTIA + tk * INERTIA_NF;
        M3 Rf;
        V3 pf;
        trunk_frame<TP>(C, tk, Rf, pf);
        const SV Vl = pick_trunk(V0, Vw, tk), al = pick_trunk(a0, aw, tk);
        const V3 cbl = pf + mul(Rf, V3{EF(bi + 1), EF(bi + 2), EF(bi + 3)});
        const SI I0 = make_si(EF(bi), cbl, rotate(Rf, S3{EF(bi + 4), EF(bi + 5), EF(bi + 6), EF(bi + 7), EF(bi + 8), EF(bi + 9)}));
        SV fx{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        if (tk == T.wrench_depth) {
          const V3 Fb = mul(Rf, extF), xc = pf + mul(Rf, wr_com);
          fx.a = -(mul(Rf, extT) + cross(xc, Fb));
          fx.l = -Fb;
        }
        LinkRec tr;
#pragma unroll
        for (int i = 0; i < B6::size; ++i) tr.A[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) tr.r[i] = 0.f;
        add_rigid(tr, I0, Vl, al, fx);
        st_rec(trunk_words(tk), tr);
      }
    }
    ctx.group_sync();
    // ---- per link group this lane owns: contacts + rigid record -> the limb's record words (group 0 = the lane's share of a
    // trunk link's spheres: straight into that trunk link's accumulator)
    uint32_t active_mask = 0;
    const uint32_t slot_valid = (uint32_t)ctx.uniform_i((int)T.slot_valid);
    const TerrainBase tb = terrain_base(u, pos.x, pos.y);
    static_for<0, NIT>([&](auto it) {
      const int g = sub + SUB * it.value;
      RL_PHASE(3, "sub.contact_fetch");
      GroupFetch gf;
      const bool fetched = group_fetch<it.value>(C, Rwb, tb, slot_valid, gf);
      RL_PHASE(4, "sub.link_records");
      LinkRec rec;
#pragma unroll
      for (int i = 0; i < B6::size; ++i) rec.A[i] = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) rec.r[i] = 0.f;
      const int l = g - 1;
      const bool has = l >= 0 && l < CL;
      {  // rigid part of limb link l (mass 0 when the lane has no link in this iteration, or the limb is shorter)
        const int lc = has ? l : 0;
        const M3 Rl = C.R(lc);
        const float* w = va_words(lc);
        const F4 w0 = ld4(w), w1 = ld4(w + 4), w2 = ld4(w + 8);
        const SV Vl{{w0.x, w0.y, w0.z}, {w0.w, w1.x, w1.y}};
        const SV al{{w1.z, w1.w, w2.x}, {w2.y, w2.z, w2.w}};
        const uint32_t fi = (uint32_t)(LY.LF_INERTIA + lc * INERTIA_NF);
        // (a lane without a link in this iteration - its group 0, the share of a trunk link's spheres - adds NOTHING rigid: mass AND rotational
        // inertia zero.  Until round 6 only the mass was: the record of a trunk-link share with an active contact carried the rotational
        // inertia of limb link 0 onto the trunk link - ~1 % of the contact forces of a robot lying on its torso, inside every fp32 envelope;
        // found by the fp64 lane program, tests/test_fp64_lane_program.py::test_fp64_robot_on_the_ground)
        const float hm = has ? 1.f : 0.f;
        const float mass = hm * LF(fi);
        const V3 cb = C.p(lc) + mul(Rl, V3{LF(fi + 1), LF(fi + 2), LF(fi + 3)});
        const SI Il = make_si(mass, cb, rotate(Rl, S3{hm * LF(fi + 4), hm * LF(fi + 5), hm * LF(fi + 6), hm * LF(fi + 7), hm * LF(fi + 8), hm * LF(fi + 9)}));
        add_rigid(rec, Il, Vl, al, SV{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}});
      }
      RL_PHASE(5, "sub.contact_pass1");
      const uint32_t before = active_mask;
      if (fetched) {
        SV Vg = V0;  // twist of the link the group's spheres ride on: limb link l, or the trunk link of the lane's share
        if (has) {
          Vg = ld_sv(va_words(l));
        } else {
          Vg = pick_trunk(V0, Vw, L.grp0_depth);
        }
        group_contacts<it.value>(Rwb, Vg, gf, rec, active_mask);
      }
      if (has) {
        st_rec(rec_words(g), rec);
      } else if (g == 0 && active_mask != before) {  // contacts of the trunk-link share
        atomic_add_rec(trunk_words(L.grp0_depth), rec);
      }
    });
    ctx.group_sync();
    if (self_on()) {  // the records are complete: pair forces onto their bias words, then the elimination may read them
      self_apply();
      ctx.group_sync();
    }
    // ---- limb elimination, tip -> attachment (results parked for the outward pass), the limb as seen from its attachment link into the
    // env's accumulator; then the trunk elimination, trunk pieces that hang off the base parked on the way, and the base solve.
    // ELIM_DIST: the 6 x 6 work of a joint shared by rows among the lanes of a DPP quad (see eliminate_rows); otherwise every sub-lane
    // eliminates whole records.
    RL_PHASE(9, "sub.aba");
    float Uhw[NW][6], uiw[NW];
    float nu0[NB], qdn[JX];
    if constexpr (ELIM_DIST) {
      Rows P;
      rows_zero(P);
      static_for_down<CL - 1>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        rows_add(this->rec_words(j + 1), P);
        V3 ax, pj;
        C.axp(j, ax, pj);
        const V3 lx = cross(pj, ax);
        const float s6[6] = {ax.x, ax.y, ax.z, lx.x, lx.y, lx.z};
        float D, uu, Uh[6], ui;
        if constexpr (KIN_SCAN) {
          D = this->ctx.template leg_bcast<j>(D_own);
          uu = this->ctx.template leg_bcast<j>(uu_own);
        } else {
          this->joint_terms(j, j >= this->L.nj, tau_e, pd_diag, pd_rhs, D, uu);
        }
        this->eliminate_rows(P, s6, D, uu, Uh, ui);
        float* o = this->va_words(j);  // (the link velocities parked here are no longer needed)
        st4(o, F4{Uh[0], Uh[1], Uh[2], Uh[3]});
        st4(o + 4, F4{Uh[4], Uh[5], ui, 0.f});
      });
      if (sub < 4) rows_atomic_add(trunk_words(L.attach), P);  // (eight sub-lanes per limb: its second quad holds a copy)
      ctx.group_sync();
      RL_PHASE(10, "sub.cross_leg_sum");
      rows_zero(P);
      Rows Pb;  // what the trunk pieces already eliminated hand to the base (a piece ends where its first joint hangs off the base)
      rows_zero(Pb);
#pragma unroll
      for (int d = NW; d >= 1; --d) {
        rows_add(trunk_words(d), P);
        const float s6[6] = {Sw[d - 1].a.x, Sw[d - 1].a.y, Sw[d - 1].a.z, Sw[d - 1].l.x, Sw[d - 1].l.y, Sw[d - 1].l.z};
        float D, uu;
        joint_terms(CL + d - 1, d - 1 >= T.nw_used, tau_e, pd_diag, pd_rhs, D, uu);
        eliminate_rows(P, s6, D, uu, Uhw[d - 1], uiw[d - 1]);
        if (d > 1 && ((u.trunk_restart >> (d - 1)) & 1u)) {  // (wave-uniform) joint d - 1 hangs off the base: park the piece, the next joint starts another
          rows_acc(Pb, P);
          rows_zero(P);
        }
      }
      rows_add(trunk_words(0), P);
      rows_acc(P, Pb);
      RL_PHASE(11, "sub.trunk_solve");
      float A6[B6::size], r6[6], n6[6];
      rows_gather(P, A6, r6);
      solve6(A6, r6, n6);
#pragma unroll
      for (int i = 0; i < 6; ++i) nu0[i] = n6[i];
    } else {
      LinkRec P;
#pragma unroll
      for (int i = 0; i < B6::size; ++i) P.A[i] = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) P.r[i] = 0.f;
      // REC_PREFETCH: the NEXT joint's link record and axis / origin are in flight while this joint is eliminated (and below: the trunk's
      // accumulators likewise, the outward pass's U / D, u / D, axes and origins of all limb joints in one batch in front of its chain) -
      // a lone wavefront otherwise waits out one LDS round trip per joint, behind stores the compiler cannot move the reads across.
      // G1 82.05 -> 80.14 us (profiles/r06t_g1_prefetch_ab.txt, r06u_g1_prefetch2_ab.txt); -DRL_NO_REC_PREFETCH: read where consumed.
      F4 rnext[7];
      V3 axn{0.f, 0.f, 0.f}, pjn{0.f, 0.f, 0.f};
      if constexpr (REC_PREFETCH) {
        ld_rec(this->rec_words(CL), rnext);
        C.axp(CL - 1, axn, pjn);
      }
      static_for_down<CL - 1>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        V3 ax, pj;
        if constexpr (REC_PREFETCH) {
          acc_rec(rnext, P);
          ax = axn; pj = pjn;
          if constexpr (j > 0) {
            ld_rec(this->rec_words(j), rnext);
            C.axp(j - 1, axn, pjn);
          }
        } else {
          add_rec(this->rec_words(j + 1), P);
          C.axp(j, ax, pj);
        }
        const V3 lx = cross(pj, ax);
        const float s6[6] = {ax.x, ax.y, ax.z, lx.x, lx.y, lx.z};
        float D, uu, Uh[6], ui;
        if constexpr (KIN_SCAN) {
          D = this->ctx.template leg_bcast<j>(D_own);
          uu = this->ctx.template leg_bcast<j>(uu_own);
        } else {
          this->joint_terms(j, j >= this->L.nj, tau_e, pd_diag, pd_rhs, D, uu);
        }
        this->eliminate(P, s6, D, uu, Uh, ui);
        float* o = this->va_words(j);  // (the link velocities parked here are no longer needed)
        st4(o, F4{Uh[0], Uh[1], Uh[2], Uh[3]});
        st4(o + 4, F4{Uh[4], Uh[5], ui, 0.f});
      });
      if (sub == 0) atomic_add_rec(trunk_words(L.attach), P);
      ctx.group_sync();
      RL_PHASE(10, "sub.cross_leg_sum");
#pragma unroll
      for (int i = 0; i < B6::size; ++i) P.A[i] = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) P.r[i] = 0.f;
      LinkRec Pb;  // what the trunk pieces already eliminated hand to the base (a piece ends where its first joint hangs off the base)
#pragma unroll
      for (int i = 0; i < B6::size; ++i) Pb.A[i] = 0.f;
#pragma unroll
      for (int i = 0; i < 6; ++i) Pb.r[i] = 0.f;
      if constexpr (REC_PREFETCH) ld_rec(trunk_words(NW), rnext);
#pragma unroll
      for (int d = NW; d >= 1; --d) {
        if constexpr (REC_PREFETCH) {
          acc_rec(rnext, P);
          ld_rec(trunk_words(d - 1), rnext);
        } else add_rec(trunk_words(d), P);
        const float s6[6] = {Sw[d - 1].a.x, Sw[d - 1].a.y, Sw[d - 1].a.z, Sw[d - 1].l.x, Sw[d - 1].l.y, Sw[d - 1].l.z};
        float D, uu;
        joint_terms(CL + d - 1, d - 1 >= T.nw_used, tau_e, pd_diag, pd_rhs, D, uu);
        eliminate(P, s6, D, uu, Uhw[d - 1], uiw[d - 1]);
        if (d > 1 && ((u.trunk_restart >> (d - 1)) & 1u)) {  // (wave-uniform) joint d - 1 hangs off the base: park the piece, the next joint starts another
#pragma unroll
          for (int i = 0; i < B6::size; ++i) { Pb.A[i] += P.A[i]; P.A[i] = 0.f; }
#pragma unroll
          for (int i = 0; i < 6; ++i) { Pb.r[i] += P.r[i]; P.r[i] = 0.f; }
        }
      }
      if constexpr (REC_PREFETCH) acc_rec(rnext, P);
      else add_rec(trunk_words(0), P);
#pragma unroll
      for (int i = 0; i < B6::size; ++i) P.A[i] += Pb.A[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) P.r[i] += Pb.r[i];
      RL_PHASE(11, "sub.trunk_solve");
      float n6[6];
      solve6(P.A, P.r, n6);
#pragma unroll
      for (int i = 0; i < 6; ++i) nu0[i] = n6[i];
    }
    RL_PHASE(12, "sub.back_subst");
    SV Vnew[NIT];
    {
      float vp[6] = {nu0[0], nu0[1], nu0[2], nu0[3], nu0[4], nu0[5]};
      float va[6] = {vp[0], vp[1], vp[2], vp[3], vp[4], vp[5]};  // velocity of the link this limb hangs off
      // the contact sensor sees the velocity-LIMITED joint velocities (limit applied to the solution, then the forces -
      // oracle/physics.py): a second running twist built from the clamped values, kept per link for the sensor pass
      SV vc{{vp[0], vp[1], vp[2]}, {vp[3], vp[4], vp[5]}}, vca = vc, vcg = vc;  // vcg: the trunk link the lane's group 0 rides on (LaneTab::grp0_depth)
      const SV vc0 = vc;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        if (i > 0 && ((u.trunk_restart >> i) & 1u)) {  // a trunk piece that starts at the base: its parent's velocity is the base's
#pragma unroll
          for (int r = 0; r < 6; ++r) vp[r] = nu0[r];
          vc = vc0;
        }
        float t = uiw[i];
#pragma unroll
        for (int r = 0; r < 6; ++r) t -= Uhw[i][r] * vp[r];
        qdn[CL + i] = t;
        nu0[6 + i] = t;
        vp[0] += Sw[i].a.x * t; vp[1] += Sw[i].a.y * t; vp[2] += Sw[i].a.z * t;
        vp[3] += Sw[i].l.x * t; vp[4] += Sw[i].l.y * t; vp[5] += Sw[i].l.z * t;
        const float tc = clampf(t, -L.vel_limit[CL + i], L.vel_limit[CL + i]);
        vc.a += tc * Sw[i].a; vc.l += tc * Sw[i].l;
        if (L.attach == i + 1) {
#pragma unroll
          for (int r = 0; r < 6; ++r) va[r] = vp[r];
          vca = vc;
        }
        if (L.grp0_depth == i + 1) vcg = vc;
      }
      vc = vca;
      F4 po0[CL], po1[CL];
      V3 pax[CL], ppj[CL];
      if constexpr (REC_PREFETCH) {
#pragma unroll
        for (int j = 0; j < CL; ++j) { po0[j] = ld4(va_words(j)); po1[j] = ld4(va_words(j) + 4); }
#pragma unroll
        for (int j = 0; j < CL; ++j) C.axp(j, pax[j], ppj[j]);
      }
#pragma unroll
      for (int j = 0; j < CL; ++j) {
        F4 o0, o1;
        if constexpr (REC_PREFETCH) { o0 = po0[j]; o1 = po1[j]; }
        else { o0 = ld4(va_words(j)); o1 = ld4(va_words(j) + 4); }
        const float t = o1.z - (o0.x * va[0] + o0.y * va[1] + o0.z * va[2] + o0.w * va[3] + o1.x * va[4] + o1.y * va[5]);
        qdn[j] = t;
        V3 ax, pj;
        if constexpr (REC_PREFETCH) { ax = pax[j]; pj = ppj[j]; }
        else C.axp(j, ax, pj);
        const V3 lx = cross(pj, ax);
        va[0] += ax.x * t; va[1] += ax.y * t; va[2] += ax.z * t;
        va[3] += lx.x * t; va[4] += lx.y * t; va[5] += lx.z * t;
        const float tc = clampf(t, -L.vel_limit[j], L.vel_limit[j]);
        vc.a += tc * ax; vc.l += tc * lx;
        float* nw = rec_words(j + 1);  // (the record of link j was consumed by the elimination)
        st4(nw, F4{vc.a.x, vc.a.y, vc.a.z, vc.l.x});
        st4(nw + 4, F4{vc.l.y, vc.l.z, 0.f, 0.f});
      }
      ctx.group_sync();
      static_for<0, NIT>([&](auto it) {
        const int g = sub + SUB * it.value;
        Vnew[it.value] = vcg;  // group 0: the trunk link its spheres ride on (usually the one the limb hangs off)
        if (g >= 1 && g <= CL) {
          Vnew[it.value] = ld_sv(rec_words(g));
        }
      });
    }
    sensor_and_integrate(C, Rwb, V0, nu0, qdn, active_mask, Vnew);
  }

  // 6 x 6 SPD solve (Cholesky) of the base system
  RL_FN void solve6(const float (&Cb)[B6::size], const float (&db)[6], float (&x)[6]) const {
    float G[6][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float sacc = Cb[B6::at(j, j)];
#pragma unroll
      for (int m = 0; m < j; ++m) sacc -= G[j][m] * G[j][m];
      const float inv = frsqrt(sacc);
      G[j][j] = inv;  // 1 / G_jj
#pragma unroll
      for (int i = j + 1; i < 6; ++i) {
        float t = Cb[B6::at(j, i)];
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
      y6[j] = t * G[j][j];
    }
#pragma unroll
    for (int j = 5; j >= 0; --j) {
      float t = y6[j];
#pragma unroll
      for (int i = j + 1; i < 6; ++i) t -= G[i][j] * x[i];
      x[j] = t * G[j][j];
    }
  }

  // The tail of a substep (both formulations): contact-sensor forces with the new velocities, integration, sensor timers.
  RL_FN void sensor_and_integrate(const ChainTP& C, const M3& Rwb, const SV V0, const float (&nu0)[NB], float (&qdn)[JX], const uint32_t active_mask,
                                  const SV (&Vnew)[NIT]) {
    V3 fown[MAXOWN];
    sensor_forces(C, Rwb, V0, nu0, qdn, active_mask, Vnew, fown);
    sensor_timers(fown);
    integrate(Rwb, V0, nu0, qdn);
  }

  // net contact force per owned body slot with the NEW velocities (world frame): what was actually applied.
  // `Vnew[it]`: new twist of the link of the lane's group in iteration it (16-lane mapping; unused with one lane per limb)
  RL_FN void sensor_forces(const ChainTP& C, const M3& Rwb, const SV V0, const float (&nu0)[NB], float (&qdn)[JX], const uint32_t active_mask,
                           const SV (&Vnew)[NIT], V3 (&fown)[MAXOWN]) {
#pragma unroll
    for (int j = 0; j < JX; ++j) qdn[j] = clampf(qdn[j], -L.vel_limit[j], L.vel_limit[j]);
    RL_PHASE(13, "sub.contact_pass2");
#pragma unroll
    for (int i = 0; i < MAXOWN; ++i) fown[i] = {0.f, 0.f, 0.f};
    V3 fb0{0.f, 0.f, 0.f};  // merged instances: force on the trunk body from the flagged slots of this lane (whoever owns slot 0)
    auto apply = [&](const Contact& c, const SV& Vg, int slot, bool onb) __attribute__((always_inline)) {
      V3 uu = Vg.l + cross(Vg.a, c.x);
      float un = dot(c.n, uu);
      V3 Fb = (c.bias - (c.dn - c.dt) * un) * c.n - c.dt * uu;
      V3 Fw = mul(Rwb, Fb);
      if (M0 && onb) {
        fb0 += Fw;
        return;
      }
#pragma unroll
      for (int i = 0; i < MAXOWN; ++i)
        if (own[i] == slot) fown[i] += Fw;
    };
    const SV V0n{{nu0[0], nu0[1], nu0[2]}, {nu0[3], nu0[4], nu0[5]}};
    if constexpr (STASH) {
      // every contact of pass 1 sits in the lane's stash: SPL slots per link group, masked by the activity bits, read in one batch
      static_for<0, NIT>([&](auto it) {
        const int g = grp_of(it.value);
        const int gi = g <= CL ? g : CL;
        const uint32_t bits = g <= CL ? (active_mask >> (gi * SPL)) & ((1u << SPL) - 1u) : 0u;
        if (!ctx.any(bits != 0u)) return;
        Contact c[SPL];
        int slot[SPL];
#pragma unroll
        for (int s2 = 0; s2 < SPL; ++s2) {
          c[s2].act = (bits >> s2) & 1u;
          if constexpr (STASH_REG) {
            const float (&st)[CONTACT_WORDS] = stash_r[it.value * SPL + s2];
            c[s2].x = {st[0], st[1], st[2]};
            c[s2].n = {st[3], st[4], st[5]};
            c[s2].bias = st[6]; c[s2].dn = st[7]; c[s2].dt = st[8];
          } else {
            if constexpr (LsFor<TP, SUB>::STASH_GRAN) {
              const float* st = ctx.template lane_scratch<true>() + (LS::CT + (it.value * SPL + s2) * STASH_SLOT_WORDS) * LSS;
              const F4 g0 = ld4(st), g1 = ld4(st + 4 * LSS), g2 = ld4(st + 8 * LSS);
              c[s2].x = {g0.x, g0.y, g0.z};
              c[s2].n = {g0.w, g1.x, g1.y};
              c[s2].bias = g1.z; c[s2].dn = g1.w; c[s2].dt = g2.x;
            } else {
              const float* st = ctx.template lane_scratch<false>() + (LS::CT + (it.value * SPL + s2) * CONTACT_WORDS) * LSS;
              c[s2].x = {st[0 * LSS], st[1 * LSS], st[2 * LSS]};
              c[s2].n = {st[3 * LSS], st[4 * LSS], st[5 * LSS]};
              c[s2].bias = st[6 * LSS]; c[s2].dn = st[7 * LSS]; c[s2].dt = st[8 * LSS];
            }
          }
          slot[s2] = L.sph_slot[gi][s2];
        }
#pragma unroll
        for (int s2 = 0; s2 < SPL; ++s2)
          if (c[s2].act) {
            SV Vs = Vnew[it.value];
            const bool onb = on_base(gi, s2);
            if (M0) Vs = pick_sv(onb, V0n, Vs);
            apply(c[s2], Vs, slot[s2], onb);
          }
      });
    } else {
      // one lane per limb (RL_ENV_SUB=1 / the CPU emulator's default): no stash - the spheres that were active in pass 1 are
      // evaluated again (same state -> same contact), twists from the generalised velocities
#pragma unroll 1
      for (uint32_t m = active_mask; m != 0; m &= m - 1) {
        const int ci = __builtin_ctz(m);
        const int g = ci / SPL, s2 = ci - g * SPL;
        float rad;
        V3 cb, cw;
        sphere_center(C, Rwb, g, s2, rad, cb, cw);
        const Contact c = contact_from_patch(C, Rwb, V0, qd, g, s2, rad, cb, cw, terrain_fetch(u, S.terrain, pos.x, pos.y, cw.x, cw.y));
        if (c.act) apply(c, link_twist(C, g, s2, V0n, qdn), L.sph_slot[g][s2], on_base(g, s2));
      }
    }
    RL_PHASE(14, "sub.sensor+integrate");
    // trunk-link bodies can be fed by several lanes (A1: the trunk box corners are spread over 4 lanes); slot 0,
    // when a lane has it, is the first entry of its list.  Nothing to sum when no lane of the wavefront touches with group 0.
    if (ctx.any((active_mask & base_bits()) != 0u)) {
      for (int bi = 0; bi < T.n_base_bodies; ++bi) {
        const bool member = L.base_body_local == bi, mine = member && own[0] == 0;
        const V3 part = M0 ? fb0 : fown[0];
        const bool has = M0 ? member : mine;
        V3 f{ctx.esum(has ? part.x : 0.f), ctx.esum(has ? part.y : 0.f), ctx.esum(has ? part.z : 0.f)};
        if (mine) fown[0] = f;
      }
    }
  }

  // [UPSTREAM B5] ContactSensor: history roll + air/contact timers, every physics step - each lane for the slots it owns.
  // All scratchpad reads first, then the arithmetic, then the writes: the rows are addressed through computed indices, so the
  // compiler cannot move a read above an earlier write on its own and every read-modify-write became its own LDS round trip.
  RL_FN void sensor_timers(const V3 (&fown)[MAXOWN]) {
    const float dt = u.dt;
    F4 h[MAXOWN], t[MAXOWN];
#pragma unroll
    for (int i = 0; i < MAXOWN; ++i) {
      const int b = own[i] < 0 ? 0 : own[i];
      h[i] = hist_n.ld(b);
      t[i] = tim.ld(b);
    }
#pragma unroll
    for (int i = 0; i < MAXOWN; ++i) {
      const int b = own[i];
      if (b < 0) continue;
      cf.st(b, F4{fown[i].x, fown[i].y, fown[i].z, 0.f});
      const float fn = norm(fown[i]);
      hist_n.st(b, F4{fn, h[i].x, h[i].y, 0.f});
      const bool contact = fn > u.force_threshold;
      const float ca = t[i].x, cc = t[i].y;
      const bool first_contact = (ca > 0.f) && contact, first_detach = (cc > 0.f) && !contact;
      tim.st(b, F4{contact ? 0.f : ca + dt, contact ? cc + dt : 0.f, first_contact ? ca + dt : t[i].z, first_detach ? cc + dt : t[i].w});
    }
  }

  // semi-implicit Euler: the new velocities move the positions
  RL_FN void integrate(const M3& Rwb, const SV V0, const float (&nu0)[NB], const float (&qdn)[JX]) {
    const float dt = u.dt;
#pragma unroll
    for (int j = 0; j < JX; ++j) {
      qacc[j] = (qdn[j] - qd[j]) * u.inv_dt;
      q[j] += dt * qdn[j];
      qd[j] = qdn[j];
    }
    Q4 dq = quat_normalize(Q4{1.0f, 0.5f * dt * nu0[0], 0.5f * dt * nu0[1], 0.5f * dt * nu0[2]});
    quat = quat_normalize(quat_mul(quat, dq));
    // nu+ lives in the fixed frame coincident with the body frame at t, referred to the old origin:
    // rotate with the OLD orientation and shift the reference point (+ dt omega x v) - see oracle/physics.py
    vang = mul(Rwb, V3{nu0[0], nu0[1], nu0[2]});
    vlin = mul(Rwb, V3{nu0[3], nu0[4], nu0[5]} + dt * cross(V0.a, V0.l));
    pos = pos + dt * vlin;
  }
};

}  // namespace rl
