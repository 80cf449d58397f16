// rl_env_emu.cpp - CPU lane emulator of the env-step lane program.  TEST INFRASTRUCTURE ONLY.
//
// Runs the *same* source (robot_lab_amd/csrc/env_step.h, env_terms.h) that hipcc compiles for gfx950,
// with the 4 x SUB lanes of an environment played by host threads and the wavefront shuffles by a shared
// slot + pthread barrier ("limb-shared" LDS words are per-thread copies: see Ctx::LIMB_ATOMICS).  It lets the `-m "not gpu"` tests check the lane program against the fp64
// oracle without a GPU.  The product path (robot_lab_amd.env) never loads this library: it requires
// librl_env_hip.so and fails loudly without it.
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define RL_FN inline
#include "../../robot_lab_amd/csrc/env_aos.h"
#include "../../robot_lab_amd/csrc/env_terms.h"
#include "../../robot_lab_amd/csrc/rl_env_host.h"
#include "../../robot_lab_amd/csrc/rl_env_specgen.h"

namespace {

// Sense-reversing barrier of the LPE lane threads of one environment: spins (the lanes meet every few hundred
// instructions), yields once the wait gets long - 16 lane threads may outnumber the host cores in the test tier.
struct LaneBarrier {
  std::atomic<int> count{0}, gen{0};
  int n = 1;
  void wait() {
    const int g = gen.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) == n - 1) {
      count.store(0, std::memory_order_relaxed);
      gen.store(g + 1, std::memory_order_release);
    } else {
      for (int spins = 0; gen.load(std::memory_order_acquire) == g; ++spins) {
        if (spins < 2000) __builtin_ia32_pause();
        else std::this_thread::yield();
      }
    }
  }
};

// ---- lane FIBERS: the LPE lanes of an environment as coroutines of ONE host thread (RL_EMU_FIBERS=1) --------------------------------
// The lane program is SPMD with collectives (gsum, leg_bcast, any ...); with a host thread per lane every collective is two
// inter-core barriers, which is what bench.py's "lane-emulator" CPU figure measures.  A CPU program would run an environment on
// one core: here the lanes are fibers that hand the core to the next lane at every barrier (round robin: when the last lane
// arrives, the first one is past it), so an environment costs no inter-core traffic at all and the box runs one environment per
// core (bench.py `cpu_baseline`, kind "port").  A switch saves / restores the six callee-saved registers and the stack pointer.
#if defined(__x86_64__)
extern "C" void rl_fiber_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl rl_fiber_switch
.type rl_fiber_switch,@function
rl_fiber_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size rl_fiber_switch,.-rl_fiber_switch
)");
#define RL_HAVE_FIBERS 1
#else
#define RL_HAVE_FIBERS 0
#endif

struct FiberSet {
  static constexpr size_t STACK = 512 * 1024;
  int n = 0, cur = -1;
  void* sp[33] = {};     // saved stack pointers of the lanes (up to 32: eight sub-lanes per limb); [n] = the host thread's own context
  std::vector<std::unique_ptr<char[]>> stacks;
  std::function<void(int)> body;
  static thread_local FiberSet* active;
  static void entry() {  // first activation of a lane: run its body, then hand on to the next lane (the last one: back to the host)
    FiberSet* fs = active;
    const int me = fs->cur;
    fs->body(me);
    fs->cur = me + 1;
#if RL_HAVE_FIBERS
    void* dead;
    rl_fiber_switch(&dead, fs->sp[me + 1]);
#endif
    __builtin_trap();
  }
  void run(int lanes, const std::function<void(int)>& f) {
#if RL_HAVE_FIBERS
    n = lanes;
    body = f;
    stacks.resize(n);
    for (int i = 0; i < n; ++i) {
      if (!stacks[i]) stacks[i].reset(new char[STACK]);
      uintptr_t top = ((uintptr_t)stacks[i].get() + STACK) & ~(uintptr_t)63;
      void** s = (void**)(top - 64);  // 16-byte aligned slot for the return address: rsp = 16 k + 8 at `entry`, as after a call
      s[0] = (void*)&FiberSet::entry;
      for (int r = 1; r <= 6; ++r) s[-r] = nullptr;
      sp[i] = (void*)(s - 6);
    }
    active = this;
    cur = 0;
    rl_fiber_switch(&sp[n], sp[0]);
#else
    (void)lanes; (void)f;
#endif
  }
  void yield(int me) {  // called by lane `me` at a barrier: the next lane runs
#if RL_HAVE_FIBERS
    const int nx = me + 1 == n ? 0 : me + 1;
    cur = nx;
    rl_fiber_switch(&sp[me], sp[nx]);
#else
    (void)me;
#endif
  }
};
thread_local FiberSet* FiberSet::active = nullptr;

template <int LPE>
struct alignas(64) Team {
  LaneBarrier bar;
  FiberSet* fibers = nullptr;  // set: the lanes are fibers of one thread
  alignas(64) float slot[LPE];
  float slot9[LPE][9];
  float slot12[LPE][12];
  float rstage[rl::MAX_T];
  float feat[rl::feat_count(rl::TopoMax::DMAX)];
  alignas(16) float lbchain[rl::NLANE][rl::LbLayout<rl::TopoGR>::CHAINW + 4];  // limb-shared words (really shared by the limb's sub-lane threads): kinematics
  alignas(16) float lbrec[rl::NLANE][rl::LbLayout<rl::TopoGR>::RECW + 4];      // ... link records and per-joint words
  alignas(16) float envw[rl::LbLayout<rl::TopoGR>::ENV_WORDS + 4];             // env-shared words
  float rtab[rl::REW_JS_ROWS * RL_MAX_DOF + rl::REW_BT_NF * RL_MAX_BODIES];
  float rand[rl::RESET_RAND_WORDS];
  std::vector<float> stage[2];
  Team() { bar.n = LPE; }
  void barrier(int lane) {
    if (fibers) fibers->yield(lane);
    else bar.wait();
  }
};

// Persistent worker pool: RL_EMU_TEAMS teams (default 1) of LPE lane threads; team t simulates the environments
// e = t, t + teams, ...  bench.py's cpu_baseline leg sets RL_EMU_TEAMS = host cores / 4 to time the lane program on the
// whole box; the tests use one team.
class Pool {
 public:
  static Pool& get() {
    static Pool p;
    return p;
  }
  void run(int n_threads, const std::function<void(int)>& job) {
    std::unique_lock<std::mutex> lk(m_);
    while ((int)th_.size() < n_threads) {
      const int id = (int)th_.size();
      seen_.push_back(gen_);
      th_.emplace_back([this, id] { worker(id); });
    }
    job_ = &job;
    active_ = n_threads;
    pending_ = n_threads;
    ++gen_;
    cv_.notify_all();
    done_.wait(lk, [this] { return pending_ == 0; });
    job_ = nullptr;
  }
  ~Pool() {
    {
      std::unique_lock<std::mutex> lk(m_);
      stop_ = true;
      cv_.notify_all();
    }
    for (auto& t : th_) t.join();
  }

 private:
  void worker(int id) {
    if (const char* v = std::getenv("RL_EMU_CPUS")) {  // RL_EMU_CPUS=<c0,c1,...>: pool thread i on logical CPU c[i mod n] (bench.py: one per physical core)
      std::vector<int> cpus;
      for (const char* q = v; *q;) {
        char* end;
        const long c = std::strtol(q, &end, 10);
        if (end == q) break;
        cpus.push_back((int)c);
        q = *end == ',' ? end + 1 : end;
      }
      if (!cpus.empty()) {
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(cpus[(size_t)id % cpus.size()], &one);
        pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
      }
    } else if (const char* v = std::getenv("RL_EMU_PIN")) {  // RL_EMU_PIN=<stride>: lane thread i of the pool on allowed CPU (i * stride) mod n -
      const int stride = std::max(1, std::atoi(v));   // the 4 lanes of a team on neighbouring CPUs (a barrier every few hundred instructions)
      cpu_set_t allowed;
      if (sched_getaffinity(0, sizeof(allowed), &allowed) == 0) {
        std::vector<int> cpus;
        for (int c = 0; c < CPU_SETSIZE; ++c)
          if (CPU_ISSET(c, &allowed)) cpus.push_back(c);
        if (!cpus.empty()) {
          cpu_set_t one;
          CPU_ZERO(&one);
          CPU_SET(cpus[((size_t)id * stride) % cpus.size()], &one);
          pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
        }
      }
    }
    for (;;) {
      const std::function<void(int)>* job;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return stop_ || seen_[id] != gen_; });
        if (stop_) return;
        seen_[id] = gen_;
        if (id >= active_) continue;
        job = job_;
      }
      (*job)(id);
      std::unique_lock<std::mutex> lk(m_);
      if (--pending_ == 0) done_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::vector<uint64_t> seen_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* job_ = nullptr;
  uint64_t gen_ = 0;
  int active_ = 0, pending_ = 0;
  bool stop_ = false;
};

// SUB_ sub-lanes per leg: 4 * SUB_ host threads play the lanes of one environment
template <int SUB_>
struct HostCtx {
  static constexpr int LS_STRIDE = 1;
  static constexpr bool LIMB_ATOMICS = true;
  static void limb_atomic_add(float* p, float v) {  // ds_add_f32 of the kernel: lane threads of several limbs add into env words
    static std::atomic_flag lock = ATOMIC_FLAG_INIT;
    while (lock.test_and_set(std::memory_order_acquire)) {
    }
    *p += v;
    lock.clear(std::memory_order_release);
  }
  static constexpr int SUB = SUB_;
  static constexpr int LPE = rl::NLANE * SUB_;
  static constexpr int EPT = 64 / LPE;
  alignas(16) float scratch[rl::LsLayout<rl::MAX_NBS, rl::MAX_NGRP * rl::MAX_SPL>::WORDS];  // (rows for every body slot + a stash slot for every sphere slot: more than any instance / mapping uses)
  template <bool GRAN>
  float* lane_scratch() { return scratch; }
  float* limb_chain() { return team->lbchain[k_]; }
  float* limb_rec() { return team->lbrec[k_]; }
  float* limb_rec_of(int k2) { return team->lbrec[k2]; }
  float* env_scratch() { return team->envw; }
  float uniform(float v) const { return v; }
  int uniform_i(int v) const { return v; }
  template <class P>
  static P* uniform_ptr(P* p) { return p; }
  struct StateBuf {
    char* p = nullptr;
    uint32_t bytes = 0;
  };
  static StateBuf state_buf(float* base, uint32_t bytes) { return StateBuf{reinterpret_cast<char*>(base), bytes}; }
  static float buf_ld(const StateBuf& b, uint32_t voff, uint32_t soff) { return voff + soff < b.bytes ? *reinterpret_cast<const float*>(b.p + voff + soff) : 0.f; }
  static void buf_st(const StateBuf& b, uint32_t voff, uint32_t soff, float v) {
    if (voff + soff < b.bytes) *reinterpret_cast<float*>(b.p + voff + soff) = v;
  }
  Team<LPE>* team;
  const void* T;
  int k_, sub_, e_, sense_ = 0;
  template <class TT>
  const TT& tables() const { return *static_cast<const TT*>(T); }
  template <class TT>
  const TT& tables_global() const { return *static_cast<const TT*>(T); }  // (here the lanes read the whole image in place)
  int k() const { return k_; }
  int sub() const { return sub_; }
  int env() const { return e_; }
  int tile() const { return e_ / EPT; }
  int env_in_tile() const { return e_ % EPT; }
  int li() const { return k_ * SUB + sub_; }
  // sum over the 4 legs of the values held at the same sub-lane index (replicated inputs -> leg sum)
  float gsum(float v) {
    team->slot[li()] = v;
    team->barrier(li());
    const float* p = team->slot;
    float a = p[0 * SUB + sub_] + p[1 * SUB + sub_], b = p[2 * SUB + sub_] + p[3 * SUB + sub_];
    float s = k_ < 2 ? a + b : b + a;
    team->barrier(li());
    return s;
  }
  // sum over the SUB sub-lanes of this leg (butterfly order of the DPP quad: (x0+x1)+(x2+x3))
  float leg_sum(float v) {
    if (SUB == 1) return v;
    team->slot[li()] = v;
    team->barrier(li());
    const float* p = team->slot + k_ * SUB;
    float s = SUB == 2 ? p[0] + p[1] : (p[0] + p[1]) + (p[2 % SUB] + p[3 % SUB]);
    if (SUB == 8) s += (p[4 % SUB] + p[5 % SUB]) + (p[6 % SUB] + p[7 % SUB]);  // the limb's second quad (the kernel: one more half-row mirror; a + b == b + a bitwise)
    team->barrier(li());
    return s;
  }
  float esum(float v) { return gsum(leg_sum(v)); }
  // min over all lanes of the env
  float emin(float v) {
    team->slot[li()] = v;
    team->barrier(li());
    float r = team->slot[0];
    for (int i = 1; i < LPE; ++i) r = r < team->slot[i] ? r : team->slot[i];
    team->barrier(li());
    return r;
  }
  // wave-level vote on the GPU; here an env-level OR (it may guard collectives, so it must be uniform)
  bool any(bool c) {
    team->slot[li()] = c ? 1.f : 0.f;
    team->barrier(li());
    bool r = false;
    for (int i = 0; i < LPE; ++i) r = r || team->slot[i] != 0.f;
    team->barrier(li());
    return r;
  }
  template <int J>
  float leg_bcast(float v) {
    if (SUB == 1) return v;  // a lane is the whole leg
    team->slot[li()] = v;
    team->barrier(li());
    float r = team->slot[k_ * SUB + (J < SUB ? J : 0)];
    team->barrier(li());
    return r;
  }
  // the lane's DPP quad (SUB >= 4): the four sub-lanes sub_ & ~3 .. of its limb
  float quad_sum(float v) {
    team->slot[li()] = v;
    team->barrier(li());
    const float* p = team->slot + k_ * SUB + (sub_ & ~3);
    const float a = p[0] + p[1], b = p[2 % SUB] + p[3 % SUB];
    const float s = (sub_ & 2) ? b + a : a + b;
    team->barrier(li());
    return s;
  }
  template <int J>
  float quad_bcast(float v) {
    team->slot[li()] = v;
    team->barrier(li());
    const float r = team->slot[k_ * SUB + (((sub_ & ~3) + J) % SUB)];
    team->barrier(li());
    return r;
  }
  template <int J>
  rl::M3 leg_bcast_m3(const rl::M3& m) {  // one barrier pair for the nine words
    if (SUB == 1) return m;
    float* w = team->slot9[li()];
    w[0] = m.r0.x; w[1] = m.r0.y; w[2] = m.r0.z; w[3] = m.r1.x; w[4] = m.r1.y; w[5] = m.r1.z; w[6] = m.r2.x; w[7] = m.r2.y; w[8] = m.r2.z;
    team->barrier(li());
    const float* r = team->slot9[k_ * SUB + (J < SUB ? J : 0)];
    const rl::M3 out{{r[0], r[1], r[2]}, {r[3], r[4], r[5]}, {r[6], r[7], r[8]}};
    team->barrier(li());
    return out;
  }
  template <int J>
  rl::M3 deal_bcast_m3(const rl::M3& m) {  // the dealing quad: the limb's four sub-lanes, or - eight sub-lanes per limb - the lane's half of them
    if (SUB <= 4) return leg_bcast_m3<J>(m);
    float* w = team->slot9[li()];
    w[0] = m.r0.x; w[1] = m.r0.y; w[2] = m.r0.z; w[3] = m.r1.x; w[4] = m.r1.y; w[5] = m.r1.z; w[6] = m.r2.x; w[7] = m.r2.y; w[8] = m.r2.z;
    team->barrier(li());
    const float* r = team->slot9[k_ * SUB + (sub_ & 4) + J];
    const rl::M3 out{{r[0], r[1], r[2]}, {r[3], r[4], r[5]}, {r[6], r[7], r[8]}};
    team->barrier(li());
    return out;
  }
  // w[] of sub-lane (sub - D) of this lane's limb (the kernel: row_shr:D inside the limb's half row); lanes with sub < D keep theirs
  template <int D, int N>
  void sub_shr(float (&w)[N]) {
    static_assert(N <= 12, "exchange scratch");
    float* mine = team->slot12[li()];
    for (int i = 0; i < N; ++i) mine[i] = w[i];
    team->barrier(li());
    if (sub_ >= D) {
      const float* r = team->slot12[k_ * SUB + sub_ - D];
      for (int i = 0; i < N; ++i) w[i] = r[i];
    }
    team->barrier(li());
  }
  template <int X>
  float limb_xor(float v) { return gshfl(v, k_ ^ X); }
  float gshfl(float v, int leg) {
    team->slot[li()] = v;
    team->barrier(li());
    float r = team->slot[leg * SUB + sub_];
    team->barrier(li());
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
  float* feat_stage() { return team->feat; }
  float* rew_tab() { return team->rtab; }
  float* rand_tab() { return team->rand; }
  void group_sync() { team->barrier(li()); }
  void flush_obs(float* out, int dim, int g) {
    team->barrier(li());
    for (int i = li(); i < dim; i += LPE) out[(size_t)e_ * dim + i] = team->stage[g][i];
    team->barrier(li());
  }
};

template <class TP, int SUB, class SP = rl::NoSpec>
void run(const rl::KState& S_launch, const void* Tv) {
  rl::KState S = S_launch;
  S.step_counter += *S.step_base;  // as the kernel entry does
  const rl::TablesT<TP>* T = static_cast<const rl::TablesT<TP>*>(Tv);
  using Ctx = HostCtx<SUB>;
  int teams = 1;
  if (const char* v = std::getenv("RL_EMU_TEAMS")) teams = std::max(1, std::atoi(v));
  // A team owns whole state TILES (EPT consecutive environments): every field of a tile is one 256-byte row in which neighbouring
  // environments share cache lines, so dealing single environments round robin made 16 cores write the same lines at once (false
  // sharing: 2.5 k env-steps/s per core on a 128-core box against 15 k on one core alone - round 3's cpu_baseline).
  const int tiles = ((int)S.Npad + Ctx::EPT - 1) / Ctx::EPT;
  teams = std::min(teams, tiles);
  std::vector<std::unique_ptr<Team<Ctx::LPE>>> team(teams);
  for (auto& t : team) {
    t.reset(new Team<Ctx::LPE>());
    t->stage[0].assign(std::max(1, T->policy_dim), 0.f);
    t->stage[1].assign(std::max(1, T->critic_dim), 0.f);
  }
  const auto lane_loop = [&](int tm, int l) {
    Ctx ctx;
    ctx.team = team[tm].get(); ctx.T = T; ctx.k_ = l / SUB; ctx.sub_ = l % SUB; ctx.e_ = 0;
    for (int tile = tm; tile < tiles; tile += teams)
      for (int e = tile * Ctx::EPT; e < std::min((tile + 1) * Ctx::EPT, (int)S.Npad); ++e) {
        ctx.e_ = e;
        rl::EnvProgram<Ctx, TP, SP> prog(ctx, S);
        if (S.mode == rl::KMODE_RESET || S.mode == rl::KMODE_STEP_TAIL)
          prog.reset_entry();
        else if (S.mode == rl::KMODE_STEP_HEAD)
          prog.step_head();
        else
          prog.step();
      }
  };
  if (RL_HAVE_FIBERS && std::getenv("RL_EMU_FIBERS") && std::atoi(std::getenv("RL_EMU_FIBERS")) != 0) {
    // one host thread per team, the team's lanes as fibers of that thread: an environment never leaves its core
    const std::function<void(int)> fjob = [&](int tm) {
      static thread_local FiberSet fs;
      team[tm]->fibers = &fs;
      fs.run(Ctx::LPE, [&](int l) { lane_loop(tm, l); });
    };
    Pool::get().run(teams, fjob);
    return;
  }
  const std::function<void(int)> job = [&](int id) {
    const int tm = id / Ctx::LPE, l = id % Ctx::LPE;
    Ctx ctx;
    ctx.team = team[tm].get(); ctx.T = T; ctx.k_ = l / SUB; ctx.sub_ = l % SUB; ctx.e_ = 0;
    for (int tile = tm; tile < tiles; tile += teams)
      for (int e = tile * Ctx::EPT; e < std::min((tile + 1) * Ctx::EPT, (int)S.Npad); ++e) {
        ctx.e_ = e;
        rl::EnvProgram<Ctx, TP, SP> prog(ctx, S);
        if (S.mode == rl::KMODE_RESET || S.mode == rl::KMODE_STEP_TAIL)
          prog.reset_entry();
        else if (S.mode == rl::KMODE_STEP_HEAD)
          prog.step_head();
        else
          prog.step();
      }
  };
  Pool::get().run(teams * Ctx::LPE, job);
}

struct Backend {
  std::string err;
  const std::string& error() const { return err; }
  int sub = 1;  // RL_EMU_SUB=4 selects the 16-lanes-per-env mapping (16 host threads per env: slow), 8 the 32-lane one of the trunk + limbs instances
  int activate() { return 0; }
  int init(int) {
    if (const char* v = std::getenv("RL_EMU_SUB")) sub = std::atoi(v) == 8 ? 8 : (std::atoi(v) == 4 ? 4 : (std::atoi(v) == 2 ? 2 : 1));
    return 0;
  }
  int envs_per_wave(const rl::Tables&, int) {
    if (const char* v = std::getenv("RL_EMU_SUB")) sub = std::atoi(v) == 8 ? 8 : (std::atoi(v) == 4 ? 4 : (std::atoi(v) == 2 ? 2 : 1));
    return 16 / sub;
  }
  int configure(const rl::Tables&) { return 0; }
  int plan(const rl::Tables& T, int Npad, int, int32_t out[4]) {  // (the emulator has no launch geometry beyond its lane mapping)
    envs_per_wave(T, Npad);
    out[0] = sub; out[1] = 1; out[2] = 0; out[3] = 0;
    return 0;
  }
  void* alloc(size_t n) { return std::malloc(n ? n : 1); }
  void free(void* p) { std::free(p); }
  void zero(void* p, size_t n) { std::memset(p, 0, n); }
  void h2d(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
  void h2d_stream(void* d, const void* s, size_t n, void*) { std::memcpy(d, s, n); }
  int match_plugin(const void*) { return 0; }  // (run-time compiled Specs are HIP code objects: the emulator has none)
  int spec_id = 0;  // env_spec.h: the Spec whose constants equal the env's tables (0: the interpreter)
  // the specialised lane programs the emulator carries: every Spec with one lane per limb and with the mapping its kernel runs at the
  // BASELINE size (quadrupeds: 16 lanes per env; trunk + limbs: 32), A1 also with two sub-lanes per limb
  template <class SP>
  bool run_spec(const rl::KState& S, const void* T) {
    using TP = typename SP::TP;
    if constexpr (TP::NW > 0) {
      if (sub == 1) { run<TP, 1, SP>(S, T); return true; }
      if (sub == 8) { run<TP, 8, SP>(S, T); return true; }
    } else {
      if (sub == 1) { run<TP, 1, SP>(S, T); return true; }
      if (sub == 4) { run<TP, 4, SP>(S, T); return true; }
      if constexpr (SP::ID == 1) {
        if (sub == 2) { run<TP, 2, SP>(S, T); return true; }
      }
    }
    return false;
  }
  int launch(const rl::KState& S, const void* T, int CL, void*) {
#if !defined(RL_EMU_ONLY)
    if (spec_id != 0 && S.mode == rl::KMODE_STEP) {
      switch (spec_id) {
#define RL_SPEC_CASE(NAME, ID) case ID: if (run_spec<rl::NAME>(S, T)) return 0; break;
        RL_SPEC_LIST(RL_SPEC_CASE)
#undef RL_SPEC_CASE
        default: break;
      }
    }
#endif
    // -DRL_EMU_ONLY=<CL * 10 + sub> (+ 1000 merged, + 2000 six-joint trunk): instantiate that one lane program only (bench.py's CPU baseline
    // builds the A1 one-lane-per-limb instance in 20 s instead of all fifteen in minutes)
    switch (CL * 10 + sub) {
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 31
      case 31: run<rl::TopoQuad3, 1>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 41
      case 41: run<rl::TopoQuad4, 1>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 34
      case 34: run<rl::TopoQuad3, 4>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 32
      case 32: run<rl::TopoQuad3, 2>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 42
      case 42: run<rl::TopoQuad4, 2>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 1042
      case 1042: run<rl::TopoQuad4M, 2>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 44
      case 44: run<rl::TopoQuad4, 4>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 1041
      case 1041: run<rl::TopoQuad4M, 1>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 1044
      case 1044: run<rl::TopoQuad4M, 4>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 4041
      case 4041: run<rl::TopoQuad4R, 1>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 4042
      case 4042: run<rl::TopoQuad4R, 2>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 4044
      case 4044: run<rl::TopoQuad4R, 4>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 71
      case 71: run<rl::TopoG1, 1>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 74
      case 74: run<rl::TopoG1, 4>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 78
      case 78: run<rl::TopoG1, 8>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 2078
      case 2078: run<rl::TopoGR, 8>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 2071
      case 2071: run<rl::TopoGR, 1>(S, T); return 0;
#endif
#if !defined(RL_EMU_ONLY) || RL_EMU_ONLY == 2074
      case 2074: run<rl::TopoGR, 4>(S, T); return 0;
#endif
      default: err = "unsupported chain length"; return -1;
    }
  }
  int launch_export(const rl::KState& S, const rl::Tables* T, const rl::AosPtrs& A, void*) {
    for (int e = 0; e < S.Npad; ++e) rl::export_env(S, *T, A, e);
    return 0;
  }
  int launch_cmd_levels(float* lv, const rl::CmdLevelParams& P, const uint32_t* step_base, uint32_t step_offset, uint32_t period, void*) {
    if ((*step_base + step_offset) % period == 0u) rl::apply_cmd_levels(lv, P);
    return 0;
  }
  int launch_u32(uint32_t* p, uint32_t v, int add, void*) {
    *p = add ? *p + v : v;
    return 0;
  }
  int launch_commit(const rl::KState& S, const rl::Tables* T, const rl::AosPtrs& A, void*) {
    for (int e = 0; e < S.Npad; ++e) rl::commit_env(S, *T, A, e);
    return 0;
  }
  int d2h_sync(void* out, const void* src, size_t n, void*) {
    std::memcpy(out, src, n);
    return 0;
  }
};

}  // namespace

#include "../../robot_lab_amd/csrc/rl_env_capi.inl"

// build-time tooling (emulator library only): the C++ source of the Spec of a task (csrc/rl_env_specgen.h; tools/gen_specs.py).
// Returns the length written (0: the task cannot be specialised - rl_env_last_error says why; -1: `cap` too small).
// (rl_env_spec_source: csrc/rl_env_capi.inl - the HIP library exports it too, for specialisation at run time)

// test hook (emulator library only): the lane program's randomness primitive, for tests/test_philox.py
extern "C" float rl_test_uniform01(uint64_t seed, uint32_t env, uint32_t counter, uint32_t stream, uint32_t index) {
  return rl::uniform01(seed, env, counter, stream, index);
}
extern "C" void rl_test_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  rl::U4 r = rl::philox4x32_10(rl::U4{ctr[0], ctr[1], ctr[2], ctr[3]}, key[0], key[1]);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
