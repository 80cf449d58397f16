// rl_math.h - small fixed-size linear algebra + Philox4x32-10 for the env-step lane program.
// Plain C++17, no HIP types: compiled by hipcc for gfx950 (RL_FN = __device__ __host__) and by g++
// for the CPU lane emulator used by the `-m "not gpu"` tests (tests/emu).
#pragma once
#include <math.h>
#include <stdint.h>

#ifndef RL_FN
#define RL_FN inline
#endif

namespace rl {

// rl_pin(x): the value is in its register HERE (device: an empty asm that takes it as an operand; host: nothing).  For a batch of LDS / HBM
// loads whose uses are conditional: without it the compiler sinks each load into the branch that uses it, behind the load the condition
// came from - a chain of dependent round trips in a lone wavefront where one batch would do (env_terms.h reset_env).
#if defined(__HIP_DEVICE_COMPILE__)
RL_FN void rl_pin(float& x) { asm volatile("" : "+v"(x)); }
RL_FN void rl_pin(int& x) { asm volatile("" : "+v"(x)); }
#else
RL_FN void rl_pin(float&) {}
RL_FN void rl_pin(int&) {}
#endif
template <int N>
RL_FN void rl_pin(float (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) rl_pin(a[i]);
}

// Fast reciprocal / square root: one hardware instruction (v_rcp_f32 / v_sqrt_f32 / v_rsq_f32, 1 ulp)
// on gfx950 instead of the IEEE division / sqrt expansion (~10 instructions each); exact on the host.
// -DRL_EXACT_MATH (analysis builds, tools/build_variant.sh): the correctly rounded division / square root and libm's expf / sinf / cosf
// on the device as well - what the teacher-forced parity tier's tolerance floor is made of (the share of entries
// inside a flat 1e-5 with and without the hardware approximations, profiles/r05_exact_math_*).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RL_EXACT_MATH)
RL_FN float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
RL_FN float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
RL_FN float frsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#else
RL_FN float frcp(float x) { return 1.0f / x; }
RL_FN float fsqrt(float x) { return sqrtf(x); }
RL_FN float frsqrt(float x) { return 1.0f / sqrtf(x); }
#endif
RL_FN float fdiv(float a, float b) { return a * frcp(b); }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RL_EXACT_MATH)
RL_FN float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }  // v_exp_f32: ~1 ulp, fine at reward tolerances
#else
RL_FN float fexp(float x) { return expf(x); }
#endif
// A value the optimizer does not see through.  Needed where lanes SELECT among elements of an array that lives in the lane object
// (`sub == 1 ? q[1] : sub == 2 ? q[2] : ...`): InstCombine folds a select over loads of one object into ONE load through a selected
// address, the object can then no longer be split into registers and the whole kernel runs out of scratch (env_step.h
// chain_kinematics_dealt; seen as ScratchSize 1232 and ds_read -> flat_load all over the step kernel).
#if defined(__HIP_DEVICE_COMPILE__)
RL_FN float opaque(float x) {
  asm("" : "+v"(x));
  return x;
}
#else
RL_FN float opaque(float x) { return x; }
#endif
// Two fp32 values in an even-aligned register pair: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do both lanes of the pair in the four
// cycles one v_fma_f32 takes (a scalar operand is broadcast through op_sel, a negation is a source modifier).  Written out by hand where
// the data IS a stream of pairs (the packed upper triangle of a link record, env_step.h -DRL_PK) - the SLP vectoriser's own pairing
// costs ~90 registers and 14 % of the step (__graft_entry__.py ENV_FLAGS).  Host: two fmaf - the same arithmetic in the same order.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float F2p __attribute__((ext_vector_type(2)));
RL_FN F2p pk_fma(F2p a, F2p b, F2p c) { return __builtin_elementwise_fma(a, b, c); }
#else
struct F2p {
  float x, y;
};
RL_FN F2p pk_fma(F2p a, F2p b, F2p c) { return F2p{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
RL_FN F2p pk_mul(F2p a, F2p b) { return a * b; }
RL_FN F2p pk_add(F2p a, F2p b) { return a + b; }
#else
RL_FN F2p pk_mul(F2p a, F2p b) { return F2p{a.x * b.x, a.y * b.y}; }
RL_FN F2p pk_add(F2p a, F2p b) { return F2p{a.x + b.x, a.y + b.y}; }
#endif
RL_FN F2p pk2(float a, float b) { return F2p{a, b}; }
RL_FN F2p pk1(float a) { return F2p{a, a}; }

RL_FN float ftanh(float x) {  // x >= 0 in every use (speed norms); 1 - 2 / (e^{2x} + 1)
  return 1.0f - 2.0f * frcp(fexp(2.0f * fminf(x, 20.0f)) + 1.0f);
}

struct V3 {
  float x, y, z;
};
RL_FN V3 v3(float x, float y, float z) { return V3{x, y, z}; }
// c ? a : b, component by component.  (`c ? a : b` on two struct LVALUES is an lvalue itself - the compiler selects between two ADDRESSES; when
// one of them is a member of the lane program object, that object no longer lives in registers: round 5 measured 38 -> 130 us for one such line)
RL_FN float select1(bool c, float a, float b) { return c ? a : b; }  // (operands by value: rvalues)
RL_FN V3 select3(bool c, V3 a, V3 b) { return {c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
RL_FN V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RL_FN V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RL_FN V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
RL_FN V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
RL_FN V3 operator*(V3 a, float s) { return {s * a.x, s * a.y, s * a.z}; }
RL_FN V3& operator+=(V3& a, V3 b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
RL_FN V3& operator-=(V3& a, V3 b) { a.x -= b.x; a.y -= b.y; a.z -= b.z; return a; }
RL_FN float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RL_FN V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RL_FN float norm(V3 a) { return fsqrt(dot(a, a)); }

// 3x3 matrix, row major
struct M3 {
  V3 r0, r1, r2;
};
RL_FN M3 select_m3(bool c, const M3& a, const M3& b) { return {select3(c, a.r0, b.r0), select3(c, a.r1, b.r1), select3(c, a.r2, b.r2)}; }  // (by components: select3)
RL_FN V3 mul(const M3& m, V3 v) { return {dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)}; }
RL_FN V3 mulT(const M3& m, V3 v) { return v.x * m.r0 + v.y * m.r1 + v.z * m.r2; }
RL_FN V3 col0(const M3& m) { return {m.r0.x, m.r1.x, m.r2.x}; }
RL_FN V3 col1(const M3& m) { return {m.r0.y, m.r1.y, m.r2.y}; }
RL_FN V3 col2(const M3& m) { return {m.r0.z, m.r1.z, m.r2.z}; }
RL_FN M3 mul(const M3& a, const M3& b) {
  V3 c0 = mul(a, col0(b)), c1 = mul(a, col1(b)), c2 = mul(a, col2(b));
  return {{c0.x, c1.x, c2.x}, {c0.y, c1.y, c2.y}, {c0.z, c1.z, c2.z}};
}
RL_FN M3 identity3() { return {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}; }

// symmetric 3x3
struct S3 {
  float xx, yy, zz, xy, xz, yz;
};
RL_FN V3 mul(const S3& s, V3 v) {
  return {s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z, s.xz * v.x + s.yz * v.y + s.zz * v.z};
}
RL_FN S3 operator+(S3 a, S3 b) { return {a.xx + b.xx, a.yy + b.yy, a.zz + b.zz, a.xy + b.xy, a.xz + b.xz, a.yz + b.yz}; }
// R S R^T for rotation R
RL_FN S3 rotate(const M3& R, const S3& s) {
  V3 c0 = mul(R, V3{s.xx, s.xy, s.xz}), c1 = mul(R, V3{s.xy, s.yy, s.yz}), c2 = mul(R, V3{s.xz, s.yz, s.zz});
  // T = R S (columns c0,c1,c2); result = T R^T : out_ij = sum_k T_ik R_jk
  V3 t0{c0.x, c1.x, c2.x}, t1{c0.y, c1.y, c2.y}, t2{c0.z, c1.z, c2.z};  // rows of T
  return {dot(t0, R.r0), dot(t1, R.r1), dot(t2, R.r2), dot(t0, R.r1), dot(t0, R.r2), dot(t1, R.r2)};
}

// rotation about unit axis by angle (child -> parent coordinates)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RL_EXACT_MATH)
// v_sin_f32 / v_cos_f32 take revolutions; |error| ~1e-6 for the joint-angle range, two instructions each
// (the libm sinf / cosf expand to ~50 instructions apiece and sat in every substep's kinematics)
RL_FN void fsincos(float x, float& s, float& c) {
  const float r = x * 0.15915494309189535f;
  s = __builtin_amdgcn_sinf(r);
  c = __builtin_amdgcn_cosf(r);
}
#else
RL_FN void fsincos(float x, float& s, float& c) { s = sinf(x); c = cosf(x); }
#endif
RL_FN int imin(int a, int b) { return a < b ? a : b; }
RL_FN int imax(int a, int b) { return a > b ? a : b; }
RL_FN M3 rodrigues(V3 a, float ang) {
  float s, c;
  fsincos(ang, s, c);
  const float t = 1.0f - c;
  return {{c + t * a.x * a.x, t * a.x * a.y - s * a.z, t * a.x * a.z + s * a.y},
          {t * a.x * a.y + s * a.z, c + t * a.y * a.y, t * a.y * a.z - s * a.x},
          {t * a.x * a.z - s * a.y, t * a.y * a.z + s * a.x, c + t * a.z * a.z}};
}

struct Q4 {
  float w, x, y, z;
};
RL_FN M3 quat_to_mat(Q4 q) {
  float w = q.w, x = q.x, y = q.y, z = q.z;
  return {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)},
          {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
          {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
}
RL_FN Q4 quat_mul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
RL_FN Q4 quat_normalize(Q4 q) {
  float inv = frsqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return {q.w * inv, q.x * inv, q.y * inv, q.z * inv};
}
RL_FN Q4 quat_from_euler_xyz(float roll, float pitch, float yaw) {
  // (fsincos: v_sin_f32 / v_cos_f32 on the device, 1e-6 absolute - the same primitives the kinematics of every substep use.  libm's sinf / cosf
  // were ~300 instructions of every reset, and the wavefronts that reset an env are the ones a steady-state launch waits for)
  float cy, sy, cr, sr, cp, sp;
  fsincos(yaw * 0.5f, sy, cy);
  fsincos(roll * 0.5f, sr, cr);
  fsincos(pitch * 0.5f, sp, cp);
  return {cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp};
}

// spatial vector [angular; linear] and spatial inertia, all in base coordinates about the base origin
struct SV {
  V3 a, l;
};
RL_FN SV operator+(SV p, SV q) { return {p.a + q.a, p.l + q.l}; }
RL_FN SV operator-(SV p, SV q) { return {p.a - q.a, p.l - q.l}; }
RL_FN SV operator*(SV p, float s) { return {p.a * s, p.l * s}; }
RL_FN float dot(SV p, SV q) { return dot(p.a, q.a) + dot(p.l, q.l); }
RL_FN SV crm(SV v, SV s) { return {cross(v.a, s.a), cross(v.a, s.l) + cross(v.l, s.a)}; }  // motion x motion
RL_FN SV crf(SV v, SV f) { return {cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; }  // motion x* force
struct SI {
  float m;
  V3 h;  // m * c
  S3 I;  // rotational inertia about the origin
};
RL_FN SV apply(const SI& s, SV v) { return {mul(s.I, v.a) + cross(s.h, v.l), s.m * v.l - cross(s.h, v.a)}; }
RL_FN SI operator+(const SI& p, const SI& q) { return {p.m + q.m, p.h + q.h, p.I + q.I}; }
// rigid body (mass m, com c, inertia about com Ic - all already in base coordinates) -> spatial inertia about origin
RL_FN SI make_si(float m, V3 c, S3 Ic) {
  float cc = dot(c, c);
  S3 I{Ic.xx + m * (cc - c.x * c.x), Ic.yy + m * (cc - c.y * c.y), Ic.zz + m * (cc - c.z * c.z),
       Ic.xy - m * c.x * c.y, Ic.xz - m * c.x * c.z, Ic.yz - m * c.y * c.z};
  return {m, m * c, I};
}

// lo <= hi at every call site: on the GPU one v_med3_f32 instead of v_max_f32 + v_min_f32 (same value for non-NaN inputs)
RL_FN float clampf(float v, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fmed3f(v, lo, hi);
#else
  return fminf(fmaxf(v, lo), hi);
#endif
}
RL_FN float wrap_to_pi(float a) {  // (a + pi) mod 2 pi - pi without fmodf
  const float PI = 3.14159265358979323846f;
  return a - 2.0f * PI * floorf((a + PI) * (0.5f / PI));
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. SC'11).  uniform(seed, env, counter, stream, index) is the single
// randomness primitive of the env; tests/test_philox.py pins it to the Random123 known answers and
// to oracle/philox.py.
// ---------------------------------------------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};
RL_FN uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
RL_FN U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // ONE 64-bit product per multiplier (v_mad_u64_u32): written as mulhi + mullo hipcc emits v_mul_hi_u32 AND v_mul_lo_u32 - forty
    // quarter-rate multiplies per block instead of twenty
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c.x, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
RL_FN float u24(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
RL_FN float uniform01(uint64_t seed, uint32_t env, uint32_t counter, uint32_t stream, uint32_t index) {
  U4 r = philox4x32_10(U4{env, counter, stream, index >> 2}, (uint32_t)seed, (uint32_t)(seed >> 32));
  uint32_t s = index & 3u;
  return u24(s == 0 ? r.x : s == 1 ? r.y : s == 2 ? r.z : r.w);
}
// the 4 uniforms of one Philox block (indices 4*blk .. 4*blk+3)
RL_FN void uniform01x4(uint64_t seed, uint32_t env, uint32_t counter, uint32_t stream, uint32_t blk, float (&u)[4]) {
  U4 r = philox4x32_10(U4{env, counter, stream, blk}, (uint32_t)seed, (uint32_t)(seed >> 32));
  u[0] = u24(r.x); u[1] = u24(r.y); u[2] = u24(r.z); u[3] = u24(r.w);
}
// lo + (hi - lo) u as ONE fused multiply-add, said explicitly: left to -ffp-contract=fast the compiler fuses it or not depending on
// whether the product has other uses after its unrolling / CSE decisions, which differ between the kernels one source is compiled
// into (single- and four-wavefront workgroups, specialised and interpreted term stacks) - one ulp of difference in a draw, and the
// bit-equality canary of tests/test_gpu_canary.py can no longer tell a benign rounding from a clobbered register
RL_FN float lerp_draw(float lo, float hi, float u) { return fmaf(hi - lo, u, lo); }
RL_FN float uniform_range(uint64_t seed, uint32_t env, uint32_t counter, uint32_t stream, uint32_t index, float lo, float hi) {
  return lerp_draw(lo, hi, uniform01(seed, env, counter, stream, index));
}

enum : uint32_t { STREAM_RESET = 1, STREAM_COMMAND = 2, STREAM_PUSH = 3, STREAM_NOISE = 4, STREAM_STARTUP = 5, STREAM_ACTION = 6 };
enum : uint32_t {
  IDX_WRENCH = 0, IDX_JPOS = 8, IDX_JVEL = 40, IDX_KP = 72, IDX_KD = 104, IDX_POSE = 136, IDX_VEL = 142, IDX_CMD = 148,
  IDX_CMD_TIME = 154, IDX_PUSH_TIME = 155, IDX_LEVEL = 156,
  IDX_BUCKET = 0, IDX_MASS_ADD = 64, IDX_MASS_SCALE = 128, IDX_COM = 192, IDX_INIT_LEVEL = 300
};
static const uint32_t GLOBAL_ENV = 0xFFFFFFFFu;

}  // namespace rl
