// rl_env_emu.cpp - CPU lane emulator of the env-step lane program.  TEST INFRASTRUCTURE ONLY.
//
// Runs the *same* source (robot_lab_amd/csrc/env_step.h, env_terms.h) that hipcc compiles for gfx950,
// with the 4 lanes of an environment group played by 4 host threads and the wavefront shuffles by a
// shared slot + spin barrier.  It lets the `-m "not gpu"` tests check the lane program against the fp64
// oracle without a GPU.  The product path (robot_lab_amd.env) never loads this library: it requires
// librl_env_hip.so and fails loudly without it.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define RL_FN inline
#include "../../robot_lab_amd/csrc/env_aos.h"
#include "../../robot_lab_amd/csrc/env_terms.h"
#include "../../robot_lab_amd/csrc/rl_env_host.h"

namespace {

struct Team {
  std::atomic<int> count{0};
  std::atomic<int> sense{0};
  float slot[rl::NLANE];
  float rstage[rl::MAX_T];
  std::vector<float> stage[2];
  void barrier(int& local_sense) {
    local_sense ^= 1;
    if (count.fetch_add(1, std::memory_order_acq_rel) == rl::NLANE - 1) {
      count.store(0, std::memory_order_relaxed);
      sense.store(local_sense, std::memory_order_release);
    } else {
      while (sense.load(std::memory_order_acquire) != local_sense) {
      }
    }
  }
};

struct HostCtx {
  static constexpr int LS_STRIDE = 1;
  float scratch[rl::LS_WORDS];
  float* lane_scratch() { return scratch; }
  float uniform(float v) const { return v; }
  int uniform_i(int v) const { return v; }
  bool any(bool c) const { return c; }
  Team* team;
  const rl::Tables* T;
  int k_, e_, sense_ = 0;
  const rl::Tables& tables() const { return *T; }
  int k() const { return k_; }
  int env() const { return e_; }
  int tile() const { return e_ / rl::ENVS_PER_WAVE; }
  int lane_in_tile() const { return (e_ % rl::ENVS_PER_WAVE) * rl::NLANE + k_; }
  float gsum(float v) {
    team->slot[k_] = v;
    team->barrier(sense_);
    float s = (team->slot[0] + team->slot[1]) + (team->slot[2] + team->slot[3]);
    team->barrier(sense_);
    return s;
  }
  float gshfl(float v, int src) {
    team->slot[k_] = v;
    team->barrier(sense_);
    float r = team->slot[src];
    team->barrier(sense_);
    return r;
  }
  void atomic_add(float* p, float v) {
    static std::atomic_flag lock = ATOMIC_FLAG_INIT;
    while (lock.test_and_set(std::memory_order_acquire)) {
    }
    *p += v;
    lock.clear(std::memory_order_release);
  }
  float* obs_stage(int g) { return team->stage[g].data(); }
  float* rew_stage() { return team->rstage; }
  void group_sync() { team->barrier(sense_); }
  void flush_obs(float* out, int dim, int g) {
    team->barrier(sense_);
    for (int i = k_; i < dim; i += rl::NLANE) out[(size_t)e_ * dim + i] = team->stage[g][i];
    team->barrier(sense_);
  }
};

template <int CL>
void run(const rl::KState& S, const rl::Tables* T, int reset) {
  Team team;
  team.stage[0].assign(std::max(1, T->policy_dim), 0.f);
  team.stage[1].assign(std::max(1, T->critic_dim), 0.f);
  std::vector<std::thread> th;
  for (int k = 0; k < rl::NLANE; ++k)
    th.emplace_back([&, k]() {
      HostCtx ctx;
      ctx.team = &team; ctx.T = T; ctx.k_ = k; ctx.e_ = 0;
      for (int e = 0; e < S.Npad; ++e) {
        ctx.e_ = e;
        rl::EnvProgram<HostCtx, CL> prog(ctx, S);
        if (reset)
          prog.reset_entry();
        else
          prog.step();
      }
    });
  for (auto& t : th) t.join();
}

struct Backend {
  std::string err;
  const std::string& error() const { return err; }
  int init(int) { return 0; }
  int configure(const rl::Tables&) { return 0; }
  void* alloc(size_t n) { return std::malloc(n ? n : 1); }
  void free(void* p) { std::free(p); }
  void zero(void* p, size_t n) { std::memset(p, 0, n); }
  void h2d(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
  void h2d_stream(void* d, const void* s, size_t n, void*) { std::memcpy(d, s, n); }
  int launch(const rl::KState& S, const rl::Tables* T, int CL, int reset, void*) {
    switch (CL) {
      case 3: run<3>(S, T, reset); return 0;
      case 4: run<4>(S, T, reset); return 0;
      default: err = "unsupported chain length"; return -1;
    }
  }
  int launch_export(const rl::KState& S, const rl::Tables* T, const rl::AosPtrs& A, void*) {
    for (int e = 0; e < S.Npad; ++e) rl::export_env(S, *T, A, e);
    return 0;
  }
  int launch_import(const rl::KState& S, const rl::Tables* T, const float* r, const float* q, const float* qd, int N, int D, void*) {
    for (int e = 0; e < N; ++e) rl::import_env(S, *T, r, q, qd, e);
    return 0;
  }
  int read_and_zero(void* out, void* src, size_t n, void*) {
    std::memcpy(out, src, n);
    std::memset(src, 0, n);
    return 0;
  }
};

}  // namespace

#include "../../robot_lab_amd/csrc/rl_env_capi.inl"

// test hook (emulator library only): the lane program's randomness primitive, for tests/test_philox.py
extern "C" float rl_test_uniform01(uint64_t seed, uint32_t env, uint32_t counter, uint32_t stream, uint32_t index) {
  return rl::uniform01(seed, env, counter, stream, index);
}
extern "C" void rl_test_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  rl::U4 r = rl::philox4x32_10(rl::U4{ctr[0], ctr[1], ctr[2], ctr[3]}, key[0], key[1]);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
