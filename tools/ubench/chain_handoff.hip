// Micro-benchmark [round 5]: can launch k + 1 hide its wave start and prologue behind launch k?
//
// The step kernel at 1024 environments is 512 one-wave workgroups; wave t of every launch works on tile t, so
// step k + 1 of a tile depends on step k of the SAME tile only.  Today consecutive launches are separated by the
// stream's barrier: kernel k drains, the command processor dispatches k + 1, every wave starts, loads its tables
// and state from a cold cache (~2 k cycles), computes, stores.  This benchmark measures the alternative the round-4
// review asked to settle first: launch k + 1 WITHOUT a dependency on launch k (second stream / second graph branch),
// let its waves start and load their tables while k is still running, and hand the state over per tile through a
// flag word that wave (k, t) writes behind its stores and wave (k + 1, t) polls.
//
// The model kernel mirrors the step kernel's shape: `TAB` table loads of 16 bytes per lane (6.5 KB per wave, never
// written), `ROWS` state dwords per lane in and out (5 -> 1.3 KB per wave), and a dependent FMA chain of `work`
// iterations as the arithmetic (4 instructions per iteration).  Every step adds 1 to every state word, so the final
// state proves that every hand-over delivered fresh data.
//
//   variants:  seq      one stream, plain launches (barrier between kernels)
//              graph    one stream, a captured graph of `B` launches replayed (what jxs_step_repeat does)
//              two      two streams, alternating plain launches, flag hand-over
//              gtwo     a captured graph with two branches (even / odd launches), flag hand-over
//              any      one stream, hipExtAnyOrderLaunch (documented as unsupported on gfx9: measured anyway), flags
//   scope of the hand-over loads / stores: 0 = plain, 1 = sc1 (agent), 2 = sc0 sc1 (system)
//
// build: hipcc --offload-arch=gfx950 -O3 -o chain_handoff chain_handoff.hip ; run: ./chain_handoff [waves] [steps] [work]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

constexpr int TAB = 7;    // 16-byte table loads per lane
constexpr int ROWS = 5;   // state dwords per lane
constexpr unsigned kSpinCap = 1000000u;  // a stuck hand-over must never hang the box: give up, flag the error

template <int SCOPE>
__device__ __forceinline__ float ld(const float* p) {
  float v;
  if (SCOPE == 0) asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (SCOPE == 1) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  if (SCOPE == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int SCOPE>
__device__ __forceinline__ void st(float* p, float v) {
  if (SCOPE == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (SCOPE == 1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (SCOPE == 2) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// CHAIN = 0: the barrier between launches orders everything (today's kernel).  CHAIN = 1: flag hand-over per tile.
template <int CHAIN, int SCOPE>
__global__ __launch_bounds__(64) void step(const float4* __restrict__ tab, const float* in, float* out, unsigned* flag,
                                           unsigned* err, unsigned want, int work, long long* stamps) {
  const int wg = blockIdx.x, lane = threadIdx.x;
  const long long t0 = stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
  // tables: issued first, used last
  float4 tv[TAB];
#pragma unroll
  for (int k = 0; k < TAB; ++k) tv[k] = tab[k * 64 + lane];
  if (CHAIN) {
    unsigned spins = 0;
    for (;;) {
      const unsigned v = __hip_atomic_load(flag + wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v >= want) break;
      // (a broken chain must not cost kSpinCap polls per wave and launch: once anybody gave up, everybody leaves at once)
      if (++spins > kSpinCap || ((spins & 1023u) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
        if (lane == 0) atomicAdd(err, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  const long long t1 = stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
  float s[ROWS];
  const size_t base = (size_t)wg * ROWS * 64 + lane;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) s[r] = CHAIN ? ld<SCOPE>(in + base + r * 64) : in[base + r * 64];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = stamps ? (long long)__builtin_amdgcn_s_memtime() : 0;
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < TAB; ++k) acc += tv[k].x * 0.f;  // (keeps the table loads alive; tables hold finite numbers)
  float x = s[0] * 0.f + acc;
  for (int i = 0; i < work; ++i) {  // dependent chain: 4 VALU per iteration
    x = __builtin_fmaf(x, 0.5f, 1.0f);
    x = __builtin_fmaf(x, 0.5f, -1.0f);
    x = __builtin_fmaf(x, 0.25f, 0.5f);
    x = __builtin_fmaf(x, 0.0f, 0.0f);
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const float v = s[r] + 1.0f + x;
    if (CHAIN) st<SCOPE>(out + base + r * 64, v);
    else out[base + r * 64] = v;
  }
  if (CHAIN) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores are acknowledged before the flag goes out
    if (lane == 0) __hip_atomic_store(flag + wg, want + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (stamps && lane == 0) {
    const long long t3 = (long long)__builtin_amdgcn_s_memtime();
    long long* o = stamps + 4 * (size_t)wg;
    o[0] += t1 - t0, o[1] += t2 - t1, o[2] += t3 - t2, o[3] += 1;
  }
}

using KernelFn = void (*)(const float4*, const float*, float*, unsigned*, unsigned*, unsigned, int, long long*);

struct Bufs {
  float4* tab;
  float* st[2];
  unsigned *flag, *err;
  long long* stamps;
  int waves;
};

static void reset(const Bufs& b) {
  CK(hipMemset(b.st[0], 0, sizeof(float) * b.waves * ROWS * 64));
  CK(hipMemset(b.st[1], 0, sizeof(float) * b.waves * ROWS * 64));
  CK(hipMemset(b.flag, 0, sizeof(unsigned) * b.waves));
  CK(hipMemset(b.err, 0, sizeof(unsigned)));
  CK(hipMemset(b.stamps, 0, sizeof(long long) * 4 * b.waves));
  CK(hipDeviceSynchronize());
}

static void launch(KernelFn f, hipStream_t s, const Bufs& b, unsigned k, int work, bool stamps, int ext_flags = -1) {
  const float* in = b.st[k & 1];
  float* out = b.st[(k + 1) & 1];
  long long* sp = stamps ? b.stamps : nullptr;
  if (ext_flags >= 0) {
    hipExtLaunchKernelGGL(f, dim3(b.waves), dim3(64), 0, s, nullptr, nullptr, (unsigned)ext_flags, (const float4*)b.tab, in, out, b.flag,
                          b.err, k, work, sp);
  } else {
    hipLaunchKernelGGL(f, dim3(b.waves), dim3(64), 0, s, (const float4*)b.tab, in, out, b.flag, b.err, k, work, sp);
  }
}

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool check(const Bufs& b, unsigned steps, const char* what) {
  std::vector<float> h((size_t)b.waves * ROWS * 64);
  unsigned err = 0;
  CK(hipMemcpy(h.data(), b.st[steps & 1], h.size() * sizeof(float), hipMemcpyDeviceToHost));
  CK(hipMemcpy(&err, b.err, sizeof err, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (float v : h) bad += (v != (float)steps);
  if (bad || err) std::printf("  !! %s: %zu wrong state words of %zu, %u spin time-outs\n", what, bad, h.size(), err);
  return bad == 0 && err == 0;
}

static void report_stamps(const Bufs& b, const char* what) {
  std::vector<long long> h(4 * (size_t)b.waves);
  CK(hipMemcpy(h.data(), b.stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
  double a = 0, c = 0, d = 0, n = 0;
  for (int w = 0; w < b.waves; ++w) a += h[4 * w], c += h[4 * w + 1], d += h[4 * w + 2], n += h[4 * w + 3];
  if (n > 0) std::printf("  %s per wave and step (s_memtime ticks, 100 MHz): tables+wait %.1f, state loads %.1f, compute+stores %.1f\n", what, a / n, c / n, d / n);
}

int main(int argc, char** argv) {
  const int waves = argc > 1 ? std::atoi(argv[1]) : 512;
  const unsigned steps = argc > 2 ? (unsigned)std::atoi(argv[2]) : 2000u;
  const int work = argc > 3 ? std::atoi(argv[3]) : 450;  // 1800 dependent VALU ~ the step kernel's arithmetic
  Bufs b{};
  b.waves = waves;
  CK(hipMalloc(&b.tab, sizeof(float4) * TAB * 64));
  {
    std::vector<float> t(4 * TAB * 64, 1.0f);
    CK(hipMemcpy(b.tab, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  for (int i = 0; i < 2; ++i) CK(hipMalloc(&b.st[i], sizeof(float) * waves * ROWS * 64));
  CK(hipMalloc(&b.flag, sizeof(unsigned) * waves));
  CK(hipMalloc(&b.err, sizeof(unsigned)));
  CK(hipMalloc(&b.stamps, sizeof(long long) * 4 * waves));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  std::printf("chain_handoff: %d waves, %u steps, work %d (x4 dependent VALU)\n", waves, steps, work);

  auto timed = [&](const char* name, auto&& body) {
    for (int rep = 0; rep < 3; ++rep) {
      reset(b);
      const double t0 = now_us();
      body(false);
      CK(hipDeviceSynchronize());
      const double t1 = now_us();
      const bool ok = check(b, steps, name);
      std::printf("%-28s rep %d: %8.3f us per step%s\n", name, rep, (t1 - t0) / steps, ok ? "" : "  (WRONG)");
    }
    reset(b);
    body(true);
    CK(hipDeviceSynchronize());
    report_stamps(b, name);
  };

  // ---- seq: one stream, plain launches ----
  timed("seq (barrier, plain)", [&](bool stamps) {
    for (unsigned k = 0; k < steps; ++k) launch(step<0, 0>, s0, b, k, work, stamps);
  });
  // ---- graph: B launches captured, replayed ----
  {
    const unsigned B = 250;
    hipGraph_t g[2];
    hipGraphExec_t ge[2];
    for (int st_ = 0; st_ < 2; ++st_) {
      CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
      for (unsigned k = 0; k < B; ++k) launch(step<0, 0>, s0, b, k, work, st_ == 1);
      CK(hipStreamEndCapture(s0, &g[st_]));
      CK(hipGraphInstantiate(&ge[st_], g[st_], nullptr, nullptr, 0));
    }
    timed("graph (barrier, 250/replay)", [&](bool stamps) {
      for (unsigned k = 0; k < steps; k += B) CK(hipGraphLaunch(ge[stamps ? 1 : 0], s0));
    });
  }
  for (int scope = 0; scope < 3; ++scope) {
    KernelFn f = scope == 0 ? step<1, 0> : scope == 1 ? step<1, 1> : step<1, 2>;
    char name[64];
    // ---- two: two streams, alternating plain launches ----
    std::snprintf(name, sizeof name, "two streams, scope %d", scope);
    timed(name, [&](bool stamps) {
      for (unsigned k = 0; k < steps; ++k) launch(f, (k & 1) ? s1 : s0, b, k, work, stamps);
    });
    // ---- gtwo: a graph with two branches ----
    {
      const unsigned B = 250;  // even: the flags continue across replays (want = absolute step index is baked in: rebuild per replay index)
      // the step index is a kernel argument, so one graph per block of B steps (steps / B graphs)
      std::vector<hipGraphExec_t> execs[2];
      for (int st_ = 0; st_ < 2; ++st_)
        for (unsigned k0 = 0; k0 < steps; k0 += B) {
          hipEvent_t fork, join;
          CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
          CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
          hipGraph_t g;
          hipGraphExec_t ge;
          CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
          CK(hipEventRecord(fork, s0));
          CK(hipStreamWaitEvent(s1, fork, 0));
          for (unsigned k = k0; k < k0 + B && k < steps; ++k) launch(f, (k & 1) ? s1 : s0, b, k, work, st_ == 1);
          CK(hipEventRecord(join, s1));
          CK(hipStreamWaitEvent(s0, join, 0));
          CK(hipStreamEndCapture(s0, &g));
          CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
          execs[st_].push_back(ge);
        }
      std::snprintf(name, sizeof name, "graph 2 branches, scope %d", scope);
      timed(name, [&](bool stamps) {
        for (auto ge : execs[stamps ? 1 : 0]) CK(hipGraphLaunch(ge, s0));
      });
    }
    // ---- any: one stream, any-order launches ----
    std::snprintf(name, sizeof name, "any-order launch, scope %d", scope);
    timed(name, [&](bool stamps) {
      for (unsigned k = 0; k < steps; ++k) launch(f, s0, b, k, work, stamps, hipExtAnyOrderLaunch);
    });
  }
  return 0;
}
