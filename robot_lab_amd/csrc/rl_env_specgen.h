// rl_env_specgen.h - host: the C++ source of a task's `Spec` (env_spec.h) from its descriptor, through the SAME descriptor -> Tables
// compile rl_env_create runs (rl_env_host.h compile_tables).  Build-time tooling: exported by the CPU lane emulator library only
// (tests/emu, `rl_env_spec_source`), driven by tools/gen_specs.py, which writes csrc/spec/env_specs_gen.h; tests/test_specs.py
// regenerates the file and demands zero diff.
#pragma once
#include <stdio.h>

#include <string>

#include "env_spec.h"
#include "rl_env_host.h"

namespace rl {

inline std::string spec_flt(float v) {  // exact: a hexadecimal floating literal
  char b[64];
  if (v == 0.f) return std::string(signbit(v) ? "-0.0f" : "0.0f");
  snprintf(b, sizeof(b), "%af", (double)v);
  return b;
}

// returns "" (and sets last_error) when the task cannot be specialised
inline std::string spec_source(const rl_env_desc& d, const char* struct_name, const char* task, int id) {
  Tables* Tp = new Tables();
  Tables& T = *Tp;
  std::vector<int> bl, bs, ll, lp;
  std::string out;
  auto done = [&](const std::string& msg) {
    delete Tp;
    if (!msg.empty()) fail(msg);
    return msg.empty() ? out : std::string();
  };
  if (compile_tables(d, T, bl, bs, ll, lp)) return done(last_error());
  if (T.cur_lin || T.cur_ang) return done("command_levels_* curricula: the split step runs the interpreter");
  const char* topo = T.NW > 3 ? "TopoGR" : T.NW > 0 ? "TopoG1" : (T.rotpad ? "TopoQuad4R" : T.CL == 4 ? (T.merged ? "TopoQuad4M" : "TopoQuad4") : "TopoQuad3");
  const int inst = T.CL + (T.merged ? 100 : 0) + (T.NW > 3 ? 200 : 0) + (T.rotpad ? 400 : 0);
  int jk[RL_MAX_DOF], jj[RL_MAX_DOF];
  for (int i = 0; i < RL_MAX_DOF; ++i) jk[i] = jj[i] = -1;
  for (int k = 0; k < NLANE; ++k)
    for (int j = 0; j < T.CL + T.NW; ++j) {
      const int jid = T.lane[k].joint_own[j] ? T.lane[k].joint_id[j] : -1;
      if (jid < 0) continue;
      if (jid >= T.D || jk[jid] >= 0) return done("joint map: a task joint is owned twice");
      jk[jid] = k; jj[jid] = j;
    }
  for (int i = 0; i < T.D; ++i)
    if (jk[i] < 0) return done("joint map: a task joint has no owner lane");
  for (int t = 0; t < T.n_rewards; ++t)
    if (!spec_kind_supported(T.rew[t].kind)) return done("reward kind " + std::to_string(T.rew[t].kind) + " has no specialised evaluation");
  char b[512];
  auto add = [&](const char* fmt, auto... a) {
    snprintf(b, sizeof(b), fmt, a...);
    out += b;
  };
  add("struct %s {  // %s\n", struct_name, task);
  add("  static constexpr bool ON = true;\n  static constexpr int ID = %d;\n  static constexpr const char* TASK = \"%s\";\n", id, task);
  add("  using TP = %s;\n  static constexpr int INST = %d, D = %d, N_BODIES = %d, N_REW = %d;\n", topo, inst, T.D, T.n_bodies, T.n_rewards);
  out += "  static constexpr int JOINT_K[" + std::to_string(T.D) + "] = {";
  for (int i = 0; i < T.D; ++i) out += std::to_string(jk[i]) + (i + 1 < T.D ? ", " : "};\n");
  out += "  static constexpr int JOINT_J[" + std::to_string(T.D) + "] = {";
  for (int i = 0; i < T.D; ++i) out += std::to_string(jj[i]) + (i + 1 < T.D ? ", " : "};\n");
  {  // axis kinds of the limb joints (env_spec.h spec_axis_kind): the quadruped instances without padding joints only
    out += "  static constexpr int AXIS_KIND[" + std::to_string(T.CL) + "] = {";
    for (int j = 0; j < T.CL; ++j) {
      int kind = (T.NW == 0 && !T.rotpad) ? table_axis_kind(T.lane[0], j) : 3;
      for (int k = 0; k < NLANE; ++k)
        if (!T.lane[k].joint_own[j] || table_axis_kind(T.lane[k], j) != kind) kind = 3;
      out += std::to_string(kind) + (j + 1 < T.CL ? ", " : "};\n");
    }
  }
  add("  static constexpr RewSpec REW[%d] = {\n", T.n_rewards);
  for (int t = 0; t < T.n_rewards; ++t) {
    const RewTab& R = T.rew[t];
    const int nidx = R.kind == RL_REW_FEET_GAIT ? 4 : R.n_idx;
    out += "      {" + std::to_string(R.kind) + ", " + spec_flt(R.weight) + ", {" + spec_flt(R.p[0]) + ", " + spec_flt(R.p[1]) + ", " + spec_flt(R.p[2]) + ", " + spec_flt(R.p[3]) + "}, ";
    add("0x%xu, 0x%llxull, %d, {", R.joint_mask, (unsigned long long)R.body_mask, R.n_idx);
    for (int i = 0; i < 16; ++i) out += std::to_string(i < nidx ? T.idx_pool_a[R.idx_off + i] : 0) + (i < 15 ? ", " : "}, {");
    for (int i = 0; i < 16; ++i) out += std::to_string(i < nidx ? T.idx_pool_b[R.idx_off + i] : 0) + (i < 15 ? ", " : "}},\n");
  }
  out += "  };\n";
  add("  static constexpr int N_OBS[2] = {%d, %d}, OBS_DIM[2] = {%d, %d}, OBS_CORRUPT[2] = {%d, %d};\n", T.n_policy, T.n_critic, T.policy_dim, T.critic_dim,
      T.policy_corrupt ? 1 : 0, T.critic_corrupt ? 1 : 0);
  add("  static constexpr ObsSpec OBS[2][%d] = {\n", MAX_OBS);
  for (int g = 0; g < 2; ++g) {
    out += "      {";
    const int n = g == 0 ? T.n_policy : T.n_critic;
    for (int i = 0; i < MAX_OBS; ++i) {
      const ObsTab z{};
      const ObsTab& O = i < n ? (g == 0 ? T.policy[i] : T.critic[i]) : z;
      out += "{" + std::to_string(O.kind) + ", " + spec_flt(O.scale) + ", " + spec_flt(O.clip_lo) + ", " + spec_flt(O.clip_hi) + ", " + spec_flt(O.noise_lo) + ", " +
             spec_flt(O.noise_hi) + ", " + std::to_string(O.has_noise ? 1 : 0) + ", " + std::to_string(O.offset) + "}" + (i + 1 < MAX_OBS ? ", " : "");
      if (i % 3 == 2 && i + 1 < MAX_OBS) out += "\n       ";
    }
    out += g == 0 ? "},\n" : "}};\n";
  }
  out += "};\n";
  return done("");
}

}  // namespace rl
