// Device lane backend: one link per lane, G lanes per environment, 64/G environments per
// wavefront.  Cross-lane traffic along the kinematic tree is ds_bpermute (wavefront shuffle);
// nothing goes through LDS allocations or __syncthreads -- a block is a single wave.
#pragma once
#include <hip/hip_runtime.h>

#include "jxs_params.h"

namespace jxs {

// ---- scalar "vector" primitives (V = T on the device) --------------------------------------
// fp32 uses the hardware reciprocal / square root refined to ~0.5-1 ulp instead of the IEEE
// division / sqrt expansions (10+ instructions each on a path that is latency bound), and a
// short Cody-Waite + minimax sincos (joint angles are O(1) rad; valid for |x| < 1e4).
__device__ __forceinline__ float vsel(bool m, float a, float b) { return m ? a : b; }
__device__ __forceinline__ double vsel(bool m, double a, double b) { return m ? a : b; }
__device__ __forceinline__ int vsel(bool m, int a, int b) { return m ? a : b; }
__device__ __forceinline__ float vrcp(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);  // one Newton step
}
// [round 4] fp64: v_rcp_f64 (24 good bits) and two Newton steps -- 5 instructions, 0 ulp from 1.0 / x over 1e6 random
// arguments (tools/ubench/rcp64.hip, profiles/r04_experiments.md) -- instead of the compiler's correctly rounded
// division (div_scale, rcp, six fma, div_fmas, div_fixup: ~14).  Like the fp32 form it returns NaN, not infinity, for
// x = 0: every caller guards its argument or discards that lane (the same code runs in fp32).
__device__ __forceinline__ double vrcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
}
// 1 / x to half an ulp (a second, residual-based Newton step): the joint accelerations of the ABA are
// (u - U'a) / d with values of 1e4 rad/s^2 on the contact links, where the ~1.5 ulp of vrcp is a step error of
// 1e-6 on its own (measured: GPU median 1.4e-6 against 5e-7 in IEEE emulation)
__device__ __forceinline__ float vrcp_acc(float x) {
  const float r = vrcp(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ double vrcp_acc(double x) { return vrcp(x); }
__device__ __forceinline__ float vsqrt(float x) {
  // x * rsq(x) with one Newton step; exact zeros stay zero
  const float y = __builtin_amdgcn_rsqf(x);
  const float s = x * y;
  const float e = __builtin_fmaf(-s, s, x);
  const float r = __builtin_fmaf(0.5f * y, e, s);
  return x > 0.0f ? r : x;
}
__device__ __forceinline__ double vrsqrt(double x);
__device__ __forceinline__ double vsqrt(double x) {
  // x * rsqrt(x) and one residual step; exact zeros (and negative arguments, as sqrt's NaN is never used) stay as they are
  const double y = vrsqrt(x);
  const double s = x * y;
  const double r = __builtin_fma(0.5 * y, __builtin_fma(-s, s, x), s);
  return x > 0.0 ? r : x;
}
// 1 / sqrt(x), x > 0: hardware rsq and one Newton step (4 instructions instead of the 10 of vrcp(vsqrt(x)))
__device__ __forceinline__ float vrsqrt(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float e = __builtin_fmaf(-x * y, y, 1.0f);
  return __builtin_fmaf(0.5f * y, e, y);
}
// fp64: v_rsq_f64 and two Newton steps (3.4 ulp worst over 1e6 random arguments, tools/ubench/rcp64.hip)
__device__ __forceinline__ double vrsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  return y * __builtin_fma(-0.5 * x * y, y, 1.5);
}
__device__ __forceinline__ float vabs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double vabs(double x) { return fabs(x); }
__device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double vmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double vmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float vpow(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double vpow(double a, double b) { return pow(a, b); }
__device__ __forceinline__ void vsincos(float x, float& s, float& c) {
  // k = round(x * 2/pi); r = x - k*pi/2 (3-term Cody-Waite); polynomials on [-pi/4, pi/4]
  const float kf = __builtin_rintf(x * 0.63661977236758134f);
  float r = __builtin_fmaf(kf, -1.5707962513e+00f, x);
  r = __builtin_fmaf(kf, -7.5497894159e-08f, r);
  r = __builtin_fmaf(kf, -5.3903029534e-15f, r);
  const float r2 = r * r;
  // sin(r) ~ r + r^3 * P(r^2), cos(r) ~ 1 - r^2/2 + r^4 * Q(r^2)   (cephes sinf/cosf kernels)
  float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = __builtin_fmaf(r2, ps, -1.6666654611e-1f);
  const float sr = __builtin_fmaf(r * r2, ps, r);
  float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = __builtin_fmaf(r2, pc, 4.166664568298827e-2f);
  const float cr = __builtin_fmaf(r2 * r2, pc, __builtin_fmaf(r2, -0.5f, 1.0f));
  const int k = (int)kf;
  const bool swap = (k & 1) != 0;
  const float s0 = swap ? cr : sr;
  const float c0 = swap ? sr : cr;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ void vsincos(double x, double& s, double& c) { sincos(x, &s, &c); }

template <typename T_, int G_>
struct DeviceLanes {
  using T = T_;
  using V = T_;
  using VI = int;
  using VM = bool;
  static constexpr int G = G_;

  // Packed fp32 math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two flops-pairs per issue slot).
  // Measured on the step kernel:
  // the 3x3 products alone 9.92 -> 9.84 us (round 2).  Each helper returns false where the packed form does
  // not exist (fp64, host emulation) and the caller runs the scalar loop.
  // small integers (lane and slot indices) kept in the LDS next to real numbers: exact both ways
  static __device__ __forceinline__ V to_real(VI i) { return (V)i; }
  static __device__ __forceinline__ VI to_int(V x) { return (VI)x; }
  typedef float f2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ bool mat3mul_packed(const float* a, const float* b, float* o) {
    const f2 b0 = {b[0], b[1]}, b1 = {b[3], b[4]}, b2 = {b[6], b[7]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const f2 a0 = {a[3 * i], a[3 * i]}, a1 = {a[3 * i + 1], a[3 * i + 1]}, a2 = {a[3 * i + 2], a[3 * i + 2]};
      const f2 r = __builtin_elementwise_fma(a2, b2, __builtin_elementwise_fma(a1, b1, a0 * b0));
      o[3 * i] = r.x, o[3 * i + 1] = r.y;
      o[3 * i + 2] = a[3 * i] * b[2] + a[3 * i + 1] * b[5] + a[3 * i + 2] * b[8];
    }
    return true;
  }
  // o[j] = a[j] + s * b[j], j < 6
  static __device__ __forceinline__ bool axpy6_packed(const float* a, float s, const float* b, float* o) {
    const f2 ss = {s, s};
#pragma unroll
    for (int j = 0; j < 6; j += 2) {
      const f2 r = __builtin_elementwise_fma(ss, f2{b[j], b[j + 1]}, f2{a[j], a[j + 1]});
      o[j] = r.x, o[j + 1] = r.y;
    }
    return true;
  }
  // o[j] = a[j] * s
  static __device__ __forceinline__ bool scale6_packed(const float* a, float s, float* o) {
    const f2 ss = {s, s};
#pragma unroll
    for (int j = 0; j < 6; j += 2) {
      const f2 r = f2{a[j], a[j + 1]} * ss;
      o[j] = r.x, o[j + 1] = r.y;
    }
    return true;
  }
  // o[j] = a[j] + b[j]
  static __device__ __forceinline__ bool add6_packed(const float* a, const float* b, float* o) {
#pragma unroll
    for (int j = 0; j < 6; j += 2) {
      const f2 r = f2{a[j], a[j + 1]} + f2{b[j], b[j + 1]};
      o[j] = r.x, o[j + 1] = r.y;
    }
    return true;
  }
  // sum_j a[j] b[j]
  static __device__ __forceinline__ bool dot6_packed(const float* a, const float* b, float* o) {
    f2 r = f2{a[0], a[1]} * f2{b[0], b[1]};
    r = __builtin_elementwise_fma(f2{a[2], a[3]}, f2{b[2], b[3]}, r);
    r = __builtin_elementwise_fma(f2{a[4], a[5]}, f2{b[4], b[5]}, r);
    *o = r.x + r.y;
    return true;
  }
  // o[j] = a[j] + s * b[j] for j in [lo, hi): pairs as v_pk_fma_f32, an odd tail as one v_fma (bounds are constants after
  // unrolling at every call site)
  static __device__ __forceinline__ bool axpy_range_packed(const float* a, float s, const float* b, float* o, int lo, int hi) {
    const f2 ss = {s, s};
    int j = lo;
#pragma unroll
    for (; j + 1 < hi; j += 2) {
      const f2 r = __builtin_elementwise_fma(ss, f2{b[j], b[j + 1]}, f2{a[j], a[j + 1]});
      o[j] = r.x, o[j + 1] = r.y;
    }
    if (j < hi) o[j] = __builtin_fmaf(s, b[j], a[j]);
    return true;
  }
  template <typename... Args>
  static __device__ __forceinline__ bool axpy_range_packed(const double*, Args...) { return false; }
  template <typename... Args>
  static __device__ __forceinline__ bool axpy6_packed(const double*, Args...) { return false; }
  template <typename... Args>
  static __device__ __forceinline__ bool scale6_packed(const double*, Args...) { return false; }
  template <typename... Args>
  static __device__ __forceinline__ bool add6_packed(const double*, Args...) { return false; }
  template <typename... Args>
  static __device__ __forceinline__ bool dot6_packed(const double*, Args...) { return false; }
  static __device__ __forceinline__ bool mat3mul_packed(const double*, const double*, double*) { return false; }

  int lane_;    // lane within the group
  int base4_;   // (first wave lane of the group) * 4, for ds_bpermute byte addressing
  int env_;     // environment handled by this group
  int sub_;     // index of the environment inside its tile
  bool env_ok_;
  int N_;
  T* lds_;      // LDS scratch of this environment (row-distributed ABA), else unused
  int* flag_;   // two-wave workgroups: progress word of the inertia wave (behind the environments' areas), else null

  int wpe_;     // LDS words per environment (the areas of the environments of a wave follow each other)
  int blk_;     // tile of every batched array this wave works on (= blockIdx.x for single-wave workgroups)
  int wslot_;   // developer profiling build: stamp column block of this wave (the inertia wave stamps at +32)

  // `duo`: two-wave workgroups -- a workgroup of four waves holds TWO pairs (waves 0, 1: main and inertia wave of tile
  // 2 b; waves 2, 3: of tile 2 b + 1), each pair with its own LDS region and progress word.  Four-wave workgroups
  // because the dispatcher spreads 256 of them evenly over the chip (one wave per SIMD), while 512 two-wave
  // workgroups land unevenly (measured: tools/ubench/cu_share.hip).
  __device__ __forceinline__ DeviceLanes(int N, T* lds_base, int lds_words_per_env, bool duo = false) : N_(N) {
    const int wl = threadIdx.x & 63;
    lane_ = wl & (G - 1);
    base4_ = (wl & ~(G - 1)) << 2;
    const int pair = duo ? (int)(threadIdx.x >> 7) : 0;
    blk_ = duo ? (int)blockIdx.x * 2 + pair : (int)blockIdx.x;
    wslot_ = duo ? (int)((threadIdx.x >> 6) & 1) * 32 : 0;
    env_ = blk_ * (64 / G) + (wl / G);
    env_ok_ = env_ < N;
    sub_ = wl / G;
    T* const pair_base = lds_base + (size_t)pair * ((64 / G) * lds_words_per_env + kDuoFlagWords);
    wpe_ = lds_words_per_env;
    lds_ = pair_base + sub_ * lds_words_per_env;
    flag_ = duo ? reinterpret_cast<int*>(pair_base + (64 / G) * lds_words_per_env) : nullptr;
  }

  // ---- two-wave workgroups (jxs_core.h, run_inertia / aba_rows_main): one-way hand-over through the LDS.
  // The inertia wave publishes data with ordinary ds_write and then the number of finished levels with a
  // release store; the main wave polls that word (acquire) before it reads.  The DS queue of a CU executes the
  // operations of one wave in order, so a reader that sees the count sees the data.  The poll is bounded: a
  // missing publisher must never hang the GPU (the result is then wrong and the parity tests say so).
  __device__ __forceinline__ void flag_reset_and_barrier(bool publisher) const {
    if (publisher) {
      __hip_atomic_store(flag_, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  }
  // (wavefront-scope fences are compiler barriers only: no s_waitcnt is needed between LDS operations of one
  // wave, and a workgroup-scope release would cost the publisher an exposed LDS round trip per level)
  __device__ __forceinline__ void flag_post(int value) const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);
    __hip_atomic_store(flag_, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_sched_barrier(0);
  }
  // `seen`: the last value read (the publisher only counts up): no LDS round trip when it already covers `need`
  __device__ __forceinline__ void flag_wait(int need, int& seen) const {
    if (seen >= need) return;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int spin = 0; spin < (1 << 18); ++spin) {
      seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (seen >= need) break;
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_sched_barrier(0);
  }

  __device__ __forceinline__ VI lane() const { return lane_; }
  __device__ __forceinline__ VM all_true() const { return true; }
  // Scheduling fence: everything issued before it (a batch of shuffles) stays before, every
  // consumer after -- the batch is pipelined through the LDS crossbar and waited for once.
  // [round 4] An exec-masked LDS write `if (mask) lds_[a] = v` is a message to OTHER lanes, which the compiler cannot
  // know: it reasons per thread, merges neighbouring `if (mask)` regions, sinks the loads and arithmetic that feed the
  // store into them and orders everything around them freely.  The RungeKutta4 + RigidContacts kernel of the quadruped
  // (model-specialised, merged Delassus sweeps of jxs_rigid.inc) returned different bits from call to call that way.
  // Round 3 cured it with `s_nop 4` behind every masked write and blamed the hardware (a stale EXEC for DPP operands
  // behind a scalar write of EXEC).  tools/ubench/exec_dpp.hip measured that hazard away (profiles/
  // r04_exec_dpp_ubench.txt: 0 wrong lanes in 1e9 at 0..6 wait states; only the two documented VALU cases exist), and
  // an EMPTY volatile asm in the same place gives the same bits as the wait states did (profiles/
  // r04_masked_lds_write_bisect.md): what cured it was the compiler barrier.  So a masked write ends with one -- no
  // instruction, no wait state.
  static __device__ __forceinline__ void lds_publish() { asm volatile("" ::: "memory"); }
#ifndef JXS_FENCE_MASK  // developer knob: which instruction classes the scheduler may move across fence() (0: none)
#define JXS_FENCE_MASK 0
#endif
  __device__ __forceinline__ void fence() const { __builtin_amdgcn_sched_barrier(JXS_FENCE_MASK); }
  // Phase stamps for the developer profiling build; compiled out otherwise.
  template <class KA>
  __device__ __forceinline__ void stamp(const KA& A, int i) const {
#ifdef JXS_PHASE_TIMING
    __builtin_amdgcn_sched_barrier(0);
    if (A.dbg != nullptr && (threadIdx.x & 63) == 0) A.dbg[(size_t)blk_ * kDbgSlots + wslot_ + i] = (long long)__builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
#else
    (void)A;
    (void)i;
#endif
  }

  // developer profiling build: slot `i` = where this wave runs (HW_ID | XCC_ID << 16)
  template <class KA>
  __device__ __forceinline__ void stamp_hwid(const KA& A, int i) const {
#ifdef JXS_PHASE_TIMING
    const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    if (A.dbg != nullptr && (threadIdx.x & 63) == 0) A.dbg[(size_t)blk_ * kDbgSlots + wslot_ + i] = (long long)(id | (xcc << 16));
#else
    (void)A, (void)i;
#endif
  }
  // developer profiling build: slot `i` = max over the lanes / calls of `value` (iteration counts)
  template <class KA>
  __device__ __forceinline__ void debug_max(const KA& A, int i, int value) const {
#ifdef JXS_PHASE_TIMING
    if (A.dbg != nullptr) atomicMax(reinterpret_cast<unsigned long long*>(&A.dbg[(size_t)blk_ * kDbgSlots + i]), (unsigned long long)value);
#else
    (void)A, (void)i, (void)value;
#endif
  }
  // One count per ENVIRONMENT whose solve was discarded (jxs_solver_fault_counts): lane 0 of the group adds.
  __device__ __forceinline__ void count_fault(int* counters, int which, bool faulty) const {
    if (counters != nullptr && faulty && lane_ == 0 && env_ok_) atomicAdd(&counters[which], 1);
  }
  // developer profiling build: stamp `i` once `value` has arrived (the dummy use makes the compiler wait for it)
  template <class KA, class X>
  __device__ __forceinline__ void stamp_after(const KA& A, int i, X value) const {
#ifdef JXS_PHASE_TIMING
    asm volatile("" ::"v"(value));
    stamp(A, i);
#else
    (void)A, (void)i, (void)value;
#endif
  }

  __device__ __forceinline__ int src4(int src) const { return ((src & (G - 1)) << 2) + base4_; }
  __device__ __forceinline__ float shfl(float x, int src) const {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src4(src), __float_as_int(x)));
  }
  __device__ __forceinline__ int shfl(int x, int src) const { return __builtin_amdgcn_ds_bpermute(src4(src), x); }
  // true if the predicate holds in any lane of the wave (all environments of this tile)
  __device__ __forceinline__ bool any(bool m) const { return __builtin_amdgcn_ballot_w64(m) != 0ull; }
  // bit l = the predicate of lane l of this environment, in every lane of the environment
  __device__ __forceinline__ int env_bits(bool m) const {
    const unsigned long long b = __builtin_amdgcn_ballot_w64(m) >> (base4_ >> 2);
    return (int)((unsigned)b & (G >= 32 ? 0xffffffffu : ((1u << (G & 31)) - 1u)));
  }
  __device__ __forceinline__ double shfl(double x, int src) const {
    const int a = src4(src);
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_ds_bpermute(a, lo);
    hi = __builtin_amdgcn_ds_bpermute(a, hi);
    return __hiloint2double(hi, lo);
  }

  // DPP lane shifts (VALU operand modifiers, no LDS round trip).  They act on the whole wave, so
  // the first / last lane of a group sees a neighbour group's value: callers mask those lanes.
  template <int CTRL>
  static __device__ __forceinline__ float dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
  }
  template <int CTRL>
  static __device__ __forceinline__ double dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  }
  // value of lane `k` (0..15, a compile-time constant after unrolling) of this lane's 16-lane DPP row, in every lane of
  // the row: DPP row_newbcast (gfx90a and later)
  __device__ __forceinline__ V row_bcast(V x, int k) const {
    switch (k & 15) {
      case 0: return dpp<0x150>(x);
      case 1: return dpp<0x151>(x);
      case 2: return dpp<0x152>(x);
      case 3: return dpp<0x153>(x);
      case 4: return dpp<0x154>(x);
      case 5: return dpp<0x155>(x);
      case 6: return dpp<0x156>(x);
      case 7: return dpp<0x157>(x);
      case 8: return dpp<0x158>(x);
      case 9: return dpp<0x159>(x);
      case 10: return dpp<0x15A>(x);
      case 11: return dpp<0x15B>(x);
      case 12: return dpp<0x15C>(x);
      case 13: return dpp<0x15D>(x);
      case 14: return dpp<0x15E>(x);
      default: return dpp<0x15F>(x);
    }
  }
  // x[i] += m * (x[i] of lane K of this lane's 16-lane row), i < N: the elimination step of a Gauss-Jordan sweep whose
  // rows live in the lanes of a row.  fp32: N v_fmac_f32_dpp with the row broadcast folded into the source operand
  // (the compiler keeps v_mov_dpp + v_fma apart).  PRECONDITION (the assembler cannot see it): x[] was not written by
  // the two instructions in front of the call (VALU write -> DPP read needs two wait states; an s_nop 1 costs a lone
  // wave 9 ticks, tools/ubench/issue_rate.hip) -- in the Gauss-Jordan sweep the broadcast of the pivot, its reciprocal
  // and the multiplier stand between two steps.
  template <int K, int N>
  __device__ __forceinline__ void fmac_row_bcast(float* x, float m) const {
    if (K == 0) asm volatile("s_nop 1");  // (first step: the rows may just have been copied into place)
    static_assert(N >= 1 && N <= 6 && K >= 0 && K < 16, "fmac_row_bcast: one to six values, lane 0..15 of the row");
#define JXS_FB(i) "v_fmac_f32_dpp %" #i ", %" #i ", %[m] row_newbcast:%[k] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    if constexpr (N == 1)
      asm volatile(JXS_FB(0) : "+&v"(x[0]) : [m] "v"(m), [k] "n"(K));
    else if constexpr (N == 2)
      asm volatile(JXS_FB(0) JXS_FB(1) : "+&v"(x[0]), "+&v"(x[1]) : [m] "v"(m), [k] "n"(K));
    else if constexpr (N == 3)
      asm volatile(JXS_FB(0) JXS_FB(1) JXS_FB(2) : "+&v"(x[0]), "+&v"(x[1]), "+&v"(x[2]) : [m] "v"(m), [k] "n"(K));
    else if constexpr (N == 4)
      asm volatile(JXS_FB(0) JXS_FB(1) JXS_FB(2) JXS_FB(3)
                   : "+&v"(x[0]), "+&v"(x[1]), "+&v"(x[2]), "+&v"(x[3]) : [m] "v"(m), [k] "n"(K));
    else if constexpr (N == 5)
      asm volatile(JXS_FB(0) JXS_FB(1) JXS_FB(2) JXS_FB(3) JXS_FB(4)
                   : "+&v"(x[0]), "+&v"(x[1]), "+&v"(x[2]), "+&v"(x[3]), "+&v"(x[4]) : [m] "v"(m), [k] "n"(K));
    else
      asm volatile(JXS_FB(0) JXS_FB(1) JXS_FB(2) JXS_FB(3) JXS_FB(4) JXS_FB(5)
                   : "+&v"(x[0]), "+&v"(x[1]), "+&v"(x[2]), "+&v"(x[3]), "+&v"(x[4]), "+&v"(x[5]) : [m] "v"(m), [k] "n"(K));
#undef JXS_FB
  }
  template <int K, int N>
  __device__ __forceinline__ void fmac_row_bcast(double* x, double m) const {
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = x[i] + m * row_bcast(x[i], K);
  }
  __device__ __forceinline__ V from_next(V x) const { return dpp<0x130>(x); }  // wave_shl:1, lane i <- i+1
  __device__ __forceinline__ V from_prev(V x) const { return dpp<0x138>(x); }  // wave_shr:1, lane i <- i-1
  template <int OFF>
  __device__ __forceinline__ V row_from_next(V x) const { return dpp<0x100 + OFF>(x); }  // row_shl:OFF
  // w[k] += m * w[k]@(lane + OFF) within the 16-lane row (nothing beyond the row), k < 6: one step of the segmented
  // suffix sum of the point wrenches (link_wrench_sums) -- six v_fmac with a row_shl DPP operand where a v_mov_dpp, a
  // select and an addition per value stood.  `m`: 1.0 where the lane takes its partner's value, else 0.0.
  template <int OFF>
  __device__ __forceinline__ void fmac6_row_from_next(float* w, float m) const {
    static_assert(OFF >= 1 && OFF <= 15, "row shift");
#define JXS_S6(i) "v_fmac_f32_dpp %" #i ", %" #i ", %[m] row_shl:%[off] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t" JXS_S6(0) JXS_S6(1) JXS_S6(2) JXS_S6(3) JXS_S6(4) JXS_S6(5)
                 : "+&v"(w[0]), "+&v"(w[1]), "+&v"(w[2]), "+&v"(w[3]), "+&v"(w[4]), "+&v"(w[5])
                 : [m] "v"(m), [off] "n"(OFF));
#undef JXS_S6
  }
  template <int OFF>
  __device__ __forceinline__ void fmac6_row_from_next(double* w, double m) const {
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = w[k] + m * row_from_next<OFF>(w[k]);
  }
  // acc[i] += m * x[i]@(lane + OFF) inside the lane's 16-lane row (nothing beyond the row), i < N <= 9: a parent of the
  // link-per-lane sweeps gathers a child that sits OFF lanes up (KParams::child_shift) -- v_fmac with a row_shl DPP
  // operand instead of ds_bpermute + select + add per value
  static constexpr bool kHasRowShl = sizeof(T) == 4;
  template <int OFF, int N>
  __device__ __forceinline__ void fmac_row_shl_c(float* a, const float* x, float m) const {
    static_assert(N == 3 || N == 6 || N == 9, "three, six or nine values");
#define JXS_RS(i, j) "v_fmac_f32_dpp %" #i ", %" #j ", %[m] row_shl:%[off] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    if constexpr (N == 3)
      asm volatile("s_nop 1\n\t" JXS_RS(0, 3) JXS_RS(1, 4) JXS_RS(2, 5)
                   : "+&v"(a[0]), "+&v"(a[1]), "+&v"(a[2]) : "v"(x[0]), "v"(x[1]), "v"(x[2]), [m] "v"(m), [off] "n"(OFF));
    else if constexpr (N == 6)
      asm volatile("s_nop 1\n\t" JXS_RS(0, 6) JXS_RS(1, 7) JXS_RS(2, 8) JXS_RS(3, 9) JXS_RS(4, 10) JXS_RS(5, 11)
                   : "+&v"(a[0]), "+&v"(a[1]), "+&v"(a[2]), "+&v"(a[3]), "+&v"(a[4]), "+&v"(a[5])
                   : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), [m] "v"(m), [off] "n"(OFF));
    else
      asm volatile("s_nop 1\n\t" JXS_RS(0, 9) JXS_RS(1, 10) JXS_RS(2, 11) JXS_RS(3, 12) JXS_RS(4, 13) JXS_RS(5, 14) JXS_RS(6, 15)
                   JXS_RS(7, 16) JXS_RS(8, 17)
                   : "+&v"(a[0]), "+&v"(a[1]), "+&v"(a[2]), "+&v"(a[3]), "+&v"(a[4]), "+&v"(a[5]), "+&v"(a[6]), "+&v"(a[7]), "+&v"(a[8])
                   : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), [m] "v"(m),
                     [off] "n"(OFF));
#undef JXS_RS
  }
  template <int N>
  __device__ __forceinline__ void fmac_row_shl(float* a, const float* x, float m, int off) const {
    switch (off) {  // (wave-uniform; a constant in the model-specialised kernels)
      case 1: fmac_row_shl_c<1, N>(a, x, m); break;
      case 2: fmac_row_shl_c<2, N>(a, x, m); break;
      case 3: fmac_row_shl_c<3, N>(a, x, m); break;
      case 4: fmac_row_shl_c<4, N>(a, x, m); break;
      case 5: fmac_row_shl_c<5, N>(a, x, m); break;
      case 6: fmac_row_shl_c<6, N>(a, x, m); break;
      case 7: fmac_row_shl_c<7, N>(a, x, m); break;
      case 8: fmac_row_shl_c<8, N>(a, x, m); break;
      case 9: fmac_row_shl_c<9, N>(a, x, m); break;
      case 10: fmac_row_shl_c<10, N>(a, x, m); break;
      case 11: fmac_row_shl_c<11, N>(a, x, m); break;
      case 12: fmac_row_shl_c<12, N>(a, x, m); break;
      case 13: fmac_row_shl_c<13, N>(a, x, m); break;
      case 14: fmac_row_shl_c<14, N>(a, x, m); break;
      default: fmac_row_shl_c<15, N>(a, x, m); break;
    }
  }
  template <int N>
  __device__ __forceinline__ void fmac_row_shl(double*, const double*, double, int) const {}  // (never called: kHasRowShl)
  // acc[k] += x[k](lane+1) * m for 9 values: nine v_fmac_f32_dpp in one asm block.  hipcc does not
  // fuse mov_dpp + fma itself; inside an asm block it does not see the "VALU write -> DPP read"
  // hazard either, hence the leading s_nop 1 (2 wait states) -- operands are not rewritten inside.
  // [round 6] Every in/out operand of the multi-instruction blocks of this file is EARLY-CLOBBER ("+&v"): with a plain
  // "+v" the compiler may give an input that it knows to hold the same VALUE as the input half of an in/out operand
  // (two zeros, say) the same register -- and the first instruction of the block then overwrites the input of the
  // second.  Found by the ISA lint in the MODE_DYN_RIGID kernels (v_fmac_f32_dpp v8, v8, v2 ... v_fmac_f32_dpp v10, v8, v2).
  // `active`: the lanes that take part -- a lane that does not is not written, and reads as 0 from its neighbour
  // (bound_ctrl).  The callers switch the LAST lane of every environment off: its lane + 1 is the base link of the
  // NEXT environment, and 0 x (a non-finite value of a diverged neighbour) is not 0.
  __device__ __forceinline__ void fmac9_from_next(float* a, const float* x, float m, bool active) const {
    if (active) asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %9, %18 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %10, %18 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %11, %18 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %12, %18 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %4, %13, %18 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %5, %14, %18 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %6, %15, %18 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %7, %16, %18 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %8, %17, %18 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+&v"(a[0]), "+&v"(a[1]), "+&v"(a[2]), "+&v"(a[3]), "+&v"(a[4]), "+&v"(a[5]), "+&v"(a[6]), "+&v"(a[7]), "+&v"(a[8])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]), "v"(m));
  }
  // six values (a spatial force): the sweeps of the rigid contact models hand the first child's share to its parent
  __device__ __forceinline__ void fmac6_from_next(float* a, const float* x, float m, bool active) const {
    if (active) asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %6, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %1, %7, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %2, %8, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %3, %9, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %4, %10, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fmac_f32_dpp %5, %11, %12 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+&v"(a[0]), "+&v"(a[1]), "+&v"(a[2]), "+&v"(a[3]), "+&v"(a[4]), "+&v"(a[5])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(m));
  }
  __device__ __forceinline__ void fmac6_from_next(double* a, const double* x, double m, bool active) const {
    if (active) {
#pragma unroll
      for (int k = 0; k < 6; ++k) a[k] = a[k] + m * from_next(x[k]);
    }
  }
  __device__ __forceinline__ void fmac9_from_next(double* a, const double* x, double m, bool active) const {
    if (active) {
#pragma unroll
      for (int k = 0; k < 9; ++k) a[k] = a[k] + m * from_next(x[k]);
    }
  }

  // sum over the 8 lanes of a slot (row-distributed ABA), result in all 8: three DPP adds
  __device__ __forceinline__ double allreduce8(double x) const {
    x = x + dpp<0xB1>(x);   // quad_perm:[1,0,3,2]
    x = x + dpp<0x4E>(x);   // quad_perm:[2,3,0,1]
    x = x + dpp<0x141>(x);  // row_half_mirror
    return x;
  }
  // fp32: three v_add_f32_dpp written out -- the compiler turns the first stage into v_mov_dpp + v_fmac (recomputing the
  // product that feeds it), one dependent instruction more on the base-to-leaves chain of pass 3
  __device__ __forceinline__ float allreduce8(float x) const {
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "+&v"(x));
    return x;
  }
  // seven independent 8-lane reductions advanced stage by stage: consecutive DPP instructions never read a
  // register written by their predecessor.  fp32: v_add_f32_dpp written out (the compiler otherwise turns the
  // first stage into v_mov_dpp + v_fmac, recomputing the product that feeds it: three instructions per value
  // instead of two); the leading s_nop covers the VALU-write -> DPP-read hazard the assembler cannot see.
  __device__ __forceinline__ void allreduce8x7(float* x) const {
#define JXS_DPP7(CTRL)                                                                                       \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                           \
  "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                           \
  "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                           \
  "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                           \
  "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                           \
  "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                           \
  "v_add_f32_dpp %6, %6, %6 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    // one block: the stages follow each other without further wait states (a value is read six instructions
    // after it was written); only the first DPP needs the s_nop behind the VALU that produced its source
    asm volatile("s_nop 1\n\t" JXS_DPP7("quad_perm:[1,0,3,2]") JXS_DPP7("quad_perm:[2,3,0,1]") JXS_DPP7("row_half_mirror")
                 : "+&v"(x[0]), "+&v"(x[1]), "+&v"(x[2]), "+&v"(x[3]), "+&v"(x[4]), "+&v"(x[5]), "+&v"(x[6]));
#undef JXS_DPP7
  }
  // two independent 8-lane reductions, stage by stage (d = S.U and S.pA of a tree level)
  __device__ __forceinline__ void allreduce8x2(float* x) const {
#define JXS_DPP2(CTRL)                                                              \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    // (two values: each stage reads what its predecessor wrote one instruction earlier -- one wait state more)
    asm volatile("s_nop 1\n\t" JXS_DPP2("quad_perm:[1,0,3,2]") "s_nop 0\n\t" JXS_DPP2("quad_perm:[2,3,0,1]") "s_nop 0\n\t" JXS_DPP2("row_half_mirror")
                 : "+&v"(x[0]), "+&v"(x[1]));
#undef JXS_DPP2
  }
  __device__ __forceinline__ void allreduce8x2(double* x) const {
    x[0] = allreduce8(x[0]), x[1] = allreduce8(x[1]);
  }
  // three independent 8-lane reductions (d = S.U, S.pA and U.c of a tree level): a value is read three instructions
  // after it was written, no wait states between the stages
  __device__ __forceinline__ void allreduce8x3(float* x) const {
#define JXS_DPP3(CTRL)                                                              \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
  "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t" JXS_DPP3("quad_perm:[1,0,3,2]") JXS_DPP3("quad_perm:[2,3,0,1]") JXS_DPP3("row_half_mirror")
                 : "+&v"(x[0]), "+&v"(x[1]), "+&v"(x[2]));
#undef JXS_DPP3
  }
  __device__ __forceinline__ void allreduce8x3(double* x) const {
    x[0] = allreduce8(x[0]), x[1] = allreduce8(x[1]), x[2] = allreduce8(x[2]);
  }
  // Seven values per lane, lane = 8 * slot + row:  x[j] += c1 * x[j]@(linear row (jj + 2) % 3) + c2 * x[j]@(linear row
  // (jj + 1) % 3) of the same slot, on the angular row lanes 3 + jj (c1 = c2 = 0 on every other lane; the linear rows
  // are sources only and keep their values) -- the re-reference of a link's rows to another anchor (aba_rows).  fp32:
  // row 3 finds its two sources in its own quad (quad_perm), rows 4 and 5 in a copy shifted by four lanes (row_shr:4);
  // every source is the DPP operand of a bank-masked v_fmac: 35 vector instructions where fourteen ds_bpermute (19 ticks
  // of issue each for a lone wave) and fourteen multiply-adds stood.  The stages run value by value, so that no DPP
  // reads a register written by the two instructions in front of it.  (`s1`, `s2`: the source lanes, for the generic form)
  __device__ __forceinline__ void ang_from_lin7(float* x, float c1, float c2, int, int) const {
    float X0, X1, X2, X3, X4, X5, X6;
#define JXS_A7(OP) OP(0, 7) OP(1, 8) OP(2, 9) OP(3, 10) OP(4, 11) OP(5, 12) OP(6, 13)
#define JXS_A7_SHR(i, X) "v_mov_b32_dpp %" #X ", %" #i " row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define JXS_A7_A1(i, X) "v_fmac_f32_dpp %" #i ", %" #i ", %[c1] quad_perm:[0,1,2,2] row_mask:0xf bank_mask:0x5\n\t"
#define JXS_A7_A2(i, X) "v_fmac_f32_dpp %" #i ", %" #i ", %[c2] quad_perm:[0,1,2,1] row_mask:0xf bank_mask:0x5\n\t"
#define JXS_A7_B1(i, X) "v_fmac_f32_dpp %" #i ", %" #X ", %[c1] quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xa\n\t"
#define JXS_A7_B2(i, X) "v_fmac_f32_dpp %" #i ", %" #X ", %[c2] quad_perm:[2,0,2,3] row_mask:0xf bank_mask:0xa\n\t"
    asm volatile("s_nop 1\n\t" JXS_A7(JXS_A7_SHR) JXS_A7(JXS_A7_A1) JXS_A7(JXS_A7_A2) JXS_A7(JXS_A7_B1) JXS_A7(JXS_A7_B2)
                 : "+&v"(x[0]), "+&v"(x[1]), "+&v"(x[2]), "+&v"(x[3]), "+&v"(x[4]), "+&v"(x[5]), "+&v"(x[6]), "=&v"(X0), "=&v"(X1), "=&v"(X2),
                   "=&v"(X3), "=&v"(X4), "=&v"(X5), "=&v"(X6)
                 : [c1] "v"(c1), [c2] "v"(c2));
#undef JXS_A7
#undef JXS_A7_SHR
#undef JXS_A7_A1
#undef JXS_A7_A2
#undef JXS_A7_B1
#undef JXS_A7_B2
  }
  __device__ __forceinline__ void ang_from_lin7(double* x, double c1, double c2, int s1, int s2) const {
    double g1[7], g2[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) g1[j] = shfl(x[j], s1), g2[j] = shfl(x[j], s2);
    fence();
#pragma unroll
    for (int j = 0; j < 7; ++j) x[j] = x[j] + c1 * g2[j] + c2 * g1[j];
  }
  // acc[j] += m * x[j]@(lane + 8), j < 7, within the lane's 16-lane row (lanes 8..15 of a row add nothing): a parent slot
  // pulls the rows of a child that sits in the slot next to it (aba_rows) -- seven v_fmac with a row_shl:8 DPP operand
  // instead of seven ds_bpermute, a wait and seven additions
  static constexpr bool kHasRowShift = sizeof(T) == 4;
  __device__ __forceinline__ void fmac7_from_next_slot(float* acc, const float* x, float m) const {
#define JXS_P8(i, j) "v_fmac_f32_dpp %" #i ", %" #j ", %[m] row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t" JXS_P8(0, 7) JXS_P8(1, 8) JXS_P8(2, 9) JXS_P8(3, 10) JXS_P8(4, 11) JXS_P8(5, 12) JXS_P8(6, 13)
                 : "+&v"(acc[0]), "+&v"(acc[1]), "+&v"(acc[2]), "+&v"(acc[3]), "+&v"(acc[4]), "+&v"(acc[5]), "+&v"(acc[6])
                 : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), [m] "v"(m));
#undef JXS_P8
  }
  __device__ __forceinline__ void fmac7_from_next_slot(double*, const double*, double) const {}  // (never called: kHasRowShift)
  // Rank-one update of a 6x6 matrix whose rows sit in the lanes 0..5 of an 8-lane slot:  m[j] += s * u_j, j < 6, with
  // u_j = the value `u` of lane j of the slot -- the all-gather of u is never materialised.  fp32: the lanes of the
  // slot's low quad reach u_0..u_3 by a quad broadcast of u and u_4, u_5 by a quad broadcast of the half-mirrored u
  // (lane i <-> 7 - i), the high quad the other way round; every broadcast is the DPP operand of a v_fmac restricted
  // to its quads by the bank mask: 13 instructions, where six 8-lane reductions of the scaled rows took 21 + 3.
  __device__ __forceinline__ void rank1_rows(float* m, float u, float s) const {
    float mir;
#define JXS_R1(i, SRC, QP, BANK) "v_fmac_f32_dpp %" #i ", %[" SRC "], %[s] quad_perm:[" QP "] row_mask:0xf bank_mask:" BANK "\n\t"
    // (no leading s_nop: `u` is older than the reductions and the reciprocal that produce `s`)
    asm volatile("v_mov_b32_dpp %[mir], %[u] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 JXS_R1(0, "u", "0,0,0,0", "0x5") JXS_R1(1, "u", "1,1,1,1", "0x5") JXS_R1(2, "u", "2,2,2,2", "0x5")
                 JXS_R1(3, "u", "3,3,3,3", "0x5") JXS_R1(4, "u", "0,0,0,0", "0xa") JXS_R1(5, "u", "1,1,1,1", "0xa")
                 JXS_R1(0, "mir", "3,3,3,3", "0xa") JXS_R1(1, "mir", "2,2,2,2", "0xa") JXS_R1(2, "mir", "1,1,1,1", "0xa")
                 JXS_R1(3, "mir", "0,0,0,0", "0xa") JXS_R1(4, "mir", "3,3,3,3", "0x5") JXS_R1(5, "mir", "2,2,2,2", "0x5")
                 : "+&v"(m[0]), "+&v"(m[1]), "+&v"(m[2]), "+&v"(m[3]), "+&v"(m[4]), "+&v"(m[5]), [mir] "=&v"(mir)
                 : [u] "v"(u), [s] "v"(s));
#undef JXS_R1
  }
  __device__ __forceinline__ void rank1_rows(double* m, double u, double s) const {
    const int slot0 = (lane_ & ~7);
    double g[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) g[j] = shfl(u, slot0 + j);
    fence();
#pragma unroll
    for (int j = 0; j < 6; ++j) m[j] = m[j] + s * g[j];
  }
  // six reductions (the inertia wave of a two-wave workgroup: U = MA S without the bias entry)
  __device__ __forceinline__ void allreduce8x6(float* x) const {
#define JXS_DPP6(CTRL)                                                                                       \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                           \
  "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                           \
  "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                           \
  "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                           \
  "v_add_f32_dpp %4, %4, %4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                           \
  "v_add_f32_dpp %5, %5, %5 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t" JXS_DPP6("quad_perm:[1,0,3,2]") JXS_DPP6("quad_perm:[2,3,0,1]") JXS_DPP6("row_half_mirror")
                 : "+&v"(x[0]), "+&v"(x[1]), "+&v"(x[2]), "+&v"(x[3]), "+&v"(x[4]), "+&v"(x[5]));
#undef JXS_DPP6
  }
  __device__ __forceinline__ void allreduce8x6(double* x) const {
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = x[k] + dpp<0xB1>(x[k]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = x[k] + dpp<0x4E>(x[k]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 6; ++k) x[k] = x[k] + dpp<0x141>(x[k]);
  }
  __device__ __forceinline__ void allreduce8x7(double* x) const {
#pragma unroll
    for (int k = 0; k < 7; ++k) x[k] = x[k] + dpp<0xB1>(x[k]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 7; ++k) x[k] = x[k] + dpp<0x4E>(x[k]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 7; ++k) x[k] = x[k] + dpp<0x141>(x[k]);
  }
  // Reductions over the G lanes of one environment, result in every lane: DPP butterflies inside a
  // 16-lane row (quad_perm, row_half_mirror, row_mirror), ds_bpermute only across rows (G = 32, 64).
  template <class Op>
  __device__ __forceinline__ V env_reduce(V x, Op op) const {
    if (G >= 2) x = op(x, dpp<0xB1>(x));   // quad_perm:[1,0,3,2]
    if (G >= 4) x = op(x, dpp<0x4E>(x));   // quad_perm:[2,3,0,1]
    if (G >= 8) x = op(x, dpp<0x141>(x));  // row_half_mirror
    if (G >= 16) x = op(x, dpp<0x140>(x)); // row_mirror
    if (G >= 32) x = op(x, shfl(x, lane_ ^ 16));
    if (G >= 64) x = op(x, shfl(x, lane_ ^ 32));
    return x;
  }
  __device__ __forceinline__ V env_sum(V x) const { return env_reduce(x, [](V a, V b) { return a + b; }); }
  __device__ __forceinline__ V env_max(V x) const { return env_reduce(x, [](V a, V b) { return vmax(a, b); }); }
  __device__ __forceinline__ V env_min(V x) const { return env_reduce(x, [](V a, V b) { return vmin(a, b); }); }
  // Value of lane `src` (wave-uniform) of this environment in all its lanes.  G >= 8: one v_readlane
  // per environment of the wave (scalar lane select, no LDS crossbar round trip, no branch) and a
  // select on the environment index; G = 4 (16 environments per wave): ds_bpermute.
  __device__ __forceinline__ int env_bcast_bits(int v, int src) const {
    if (G < 8) return __builtin_amdgcn_ds_bpermute(src4(src), v);
    constexpr int E = 64 / G;  // environments per wave: 1, 2, 4 or 8
    int r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) r[e] = __builtin_amdgcn_readlane(v, e * G + src);
    int out = r[0];
#pragma unroll
    for (int e = 1; e < E; ++e) out = (sub_ == e) ? r[e] : out;
    return out;
  }
  __device__ __forceinline__ float env_bcast16(float x, int src) const {
    return __int_as_float(env_bcast_bits(__float_as_int(x), src));
  }
  __device__ __forceinline__ double env_bcast16(double x, int src) const {
    const int lo = env_bcast_bits(__double2loint(x), src), hi = env_bcast_bits(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
  }
  // keep a wave-uniform kernel-argument value in an SGPR from here on
#ifdef JXS_SPEC_ASSIGN  // (constants must stay visible to the optimiser)
  static __device__ __forceinline__ unsigned pin(unsigned x) { return x; }
  static __device__ __forceinline__ int pin(int x) { return x; }
#else
  static __device__ __forceinline__ unsigned pin(unsigned x) {
    asm volatile("" : "+s"(x));
    return x;
  }
  static __device__ __forceinline__ int pin(int x) {
    asm volatile("" : "+s"(x));
    return x;
  }
#endif
  // a value that is the same in every lane by construction, handed to the compiler as a scalar (loops
  // over it become SALU loops instead of exec-masked ones)
  static __device__ __forceinline__ unsigned uniform(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }
  static __device__ __forceinline__ unsigned long long uniform(unsigned long long x) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32));
    return ((unsigned long long)hi << 32) | lo;
  }
  // LDS scratch of this group's environment (the workgroup is one wave: program order is the
  // only synchronisation needed between a ds_write and a later ds_read of another lane)
  __device__ __forceinline__ void lds_write(int addr, T v) const { lds_[addr] = v; }
  __device__ __forceinline__ void lds_write(int addr, T v, bool mask) const {
    if (mask) lds_[addr] = v;
    lds_publish();
  }
  __device__ __forceinline__ V lds_read(int addr) const { return lds_[addr]; }
  // N consecutive words with 128-bit LDS instructions; `addr` is a multiple of 16 bytes (4 floats / 2 doubles).
  // A ds instruction costs a lone wave the same issue slot whether it moves 4 or 16 bytes (tools/ubench/
  // issue_rate.hip); what it does NOT buy is latency: a write -> read -> wait round trip is ~90 cycles exposed,
  // where a batch of ds_bpermute pipelines (the forward-kinematics rounds through LDS were slower).
  template <int N>
  __device__ __forceinline__ void lds_writev(int addr, const T* v) const {
    constexpr int W = 16 / (int)sizeof(T);
    typedef T vec __attribute__((ext_vector_type(W)));
    T* p = lds_ + addr;
#pragma unroll
    for (int i = 0; i + W <= N; i += W) {
      vec x;
#pragma unroll
      for (int e = 0; e < W; ++e) x[e] = v[i + e];
      *reinterpret_cast<vec*>(p + i) = x;
    }
    constexpr int done = N / W * W;
    if constexpr (N - done >= 2) {
      typedef T vec2 __attribute__((ext_vector_type(2)));
      *reinterpret_cast<vec2*>(p + done) = vec2{v[done], v[done + 1]};
    }
    if constexpr ((N - done) % 2 == 1) p[N - 1] = v[N - 1];
  }
  // the LDS writes issued by `f` for the lanes inside `m` only: ONE exec-masked region around them
  template <class F>
  __device__ __forceinline__ void lds_masked(bool m, F&& f) const {
    if (m) f();
    lds_publish();
  }
  template <int N>
  __device__ __forceinline__ void lds_writev_if(int addr, const T* v, bool mask) const {
    if (mask) lds_writev<N>(addr, v);
    lds_publish();
  }
  // Masked writes WITHOUT an exec-masked region: lanes outside `mask` write to `sink` (words nobody reads) instead.  No
  // branch, no join block -- the construct hipcc's spill placement mishandles under register pressure (isa_lint.py
  // masked_join) cannot arise.  Used where a kernel is at the limit of the register file (jxs_rigid.inc ls_*).
  template <int N>
  __device__ __forceinline__ void lds_writev_sel(int addr, const T* v, bool mask, int sink) const {
    lds_writev<N>(mask ? addr : sink, v);
    lds_publish();
  }
  __device__ __forceinline__ void lds_write_sel(int addr, T v, bool mask, int sink) const {
    lds_[mask ? addr : sink] = v;
    lds_publish();
  }
  template <int N>
  __device__ __forceinline__ void lds_readv(int addr, T* v) const {
    constexpr int W = 16 / (int)sizeof(T);
    typedef T vec __attribute__((ext_vector_type(W)));
    const T* p = lds_ + addr;
#pragma unroll
    for (int i = 0; i + W <= N; i += W) {
      const vec x = *reinterpret_cast<const vec*>(p + i);
#pragma unroll
      for (int e = 0; e < W; ++e) v[i + e] = x[e];
    }
    constexpr int done = N / W * W;
    if constexpr (N - done >= 2) {
      typedef T vec2 __attribute__((ext_vector_type(2)));
      const vec2 x = *reinterpret_cast<const vec2*>(p + done);
      v[done] = x[0], v[done + 1] = x[1];
    }
    if constexpr ((N - done) % 2 == 1) v[N - 1] = p[N - 1];
  }
  // ---- MFMA tiles of the contact solvers' blocked Cholesky (jxs_rigid.inc rigid_cholesky) -------------------------
  // The trailing matrix of the factorisation lives in v_mfma_f32_16x16x4_f32 accumulator tiles -- lower-triangular
  // tile pairs (ti >= tj) of every environment of the wave -- and takes the rank-3 update of a block column
  // A22 -= L21 L21^T as ONE matrix instruction per tile (K = 3 padded to 4) instead of 27 multiply-adds and 18 LDS
  // reads per lane and previous block column (tools/ubench/mfma_trailing.hip: 10.5x on that work, bit-identical:
  // an f32 MFMA is an exact chain of fused multiply-adds over k).  Lane layout, chosen so that the panel can be
  // written straight into the packed lower triangle in the LDS (rows are contiguous there): with the operands
  // swapped (A = rows of tile column tj, B = rows of tile row ti) lane (i16, k4) of the accumulator holds
  // matrix ROW 16 ti + i16 and COLUMNS 16 tj + 4 k4 + v, v = 0..3.
  static constexpr bool kHasMfma = sizeof(T) == 4;
  typedef float v4f __attribute__((ext_vector_type(4)));
  template <int NTMAX>
  struct ChTiles {
    static constexpr int E = 64 / G;
    static constexpr int NQ = NTMAX * (NTMAX + 1) / 2;
    v4f C[E][NQ > 0 ? NQ : 1];  // (NTMAX = 0: the placeholder of builds without the matrix-core path)
    float* base[E];  // the environments' LDS areas
    int nt, n, i16, k4;
    static __device__ __forceinline__ int tri(int i) { return (i * (i + 1)) / 2; }
    __device__ __forceinline__ ChTiles(const DeviceLanes& ln, int nt_, int n_) : nt(nt_), n(n_) {
      const int wl = threadIdx.x & 63;
      i16 = wl & 15, k4 = wl >> 4;
#pragma unroll
      for (int e = 0; e < E; ++e) base[e] = reinterpret_cast<float*>(ln.lds_) + (e - ln.sub_) * ln.wpe_;
    }
    // C = H (lower triangle, zero elsewhere)
    __device__ __forceinline__ void load() {
#pragma unroll
      for (int ti = 0; ti < NTMAX; ++ti) {
        const int row = 16 * ti + i16;
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) {
          if (16 * tj < n && 16 * ti < n) {
#pragma unroll
            for (int e = 0; e < E; ++e)
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                const int col = 16 * tj + 4 * k4 + v;
                const bool ok = row < n && col <= row;
                C[e][ti * (ti + 1) / 2 + tj][v] = ok ? base[e][nt + tri(ok ? row : 0) + (ok ? col : 0)] : 0.0f;
              }
          } else {
#pragma unroll
            for (int e = 0; e < E; ++e) C[e][ti * (ti + 1) / 2 + tj] = v4f{0, 0, 0, 0};
          }
        }
      }
    }
    // columns 3 jc .. 3 jc + 2 of the trailing matrix (rows from the diagonal down) -> their places in the LDS triangle
    __device__ __forceinline__ void extract(int jc) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int col = 3 * jc + c, tjc = col >> 4, kq = (col >> 2) & 3, v = col & 3;
#pragma unroll
        for (int tj = 0; tj < NTMAX; ++tj) {
          if (tj != tjc) continue;
#pragma unroll
          for (int ti = tj; ti < NTMAX; ++ti) {
            if (16 * ti >= n) continue;
            const int row = 16 * ti + i16;
            const bool mine = (k4 == kq) && row >= col && row < n;
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const v4f t = C[e][ti * (ti + 1) / 2 + tj];
              const float val = v == 0 ? t.x : v == 1 ? t.y : v == 2 ? t.z : t.w;
              if (mine) base[e][nt + tri(row) + col] = val;
            }
          }
        }
      }
    }
    // A22 -= L21 L21^T with the factor block column jc (just written to the LDS)
    __device__ __forceinline__ void update(int jc) {
      const int done = 3 * (jc + 1);  // rows / columns below `done` are finished
      float op[E][NTMAX];
#pragma unroll
      for (int ti = 0; ti < NTMAX; ++ti) {
        const bool need = 16 * (ti + 1) > done && 16 * ti < n;
        const int row = 16 * ti + i16;
        const bool ok = need && k4 < 3 && row >= done && row < n;
#pragma unroll
        for (int e = 0; e < E; ++e) op[e][ti] = ok ? base[e][nt + tri(ok ? row : 0) + (ok ? 3 * jc + k4 : 0)] : 0.0f;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ti = 0; ti < NTMAX; ++ti)
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) {
          if (!(16 * (tj + 1) > done && 16 * ti < n)) continue;  // (wave-uniform) finished tile column / beyond the matrix
#pragma unroll
          for (int e = 0; e < E; ++e)
            C[e][ti * (ti + 1) / 2 + tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(-op[e][tj], op[e][ti], C[e][ti * (ti + 1) / 2 + tj], 0, 0, 0);
        }
    }
  };
  // Between LDS writes and reads of ANOTHER lane's data: the hardware keeps the DS queue of a wave in
  // order, but the compiler reasons per thread (it may forward a masked store to the following load, or
  // order the two sides of a lane-divergent branch either way) -- this fence pins program order.
  __device__ __forceinline__ void lds_sync() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
  // developer dumps: out[env * stride + idx] = v
  __device__ __forceinline__ void dbg_store(T* out, int stride, int idx, T v, bool mask) const {
    if (mask && env_ok_) out[(size_t)env_ * stride + idx] = v;
  }

  // per-lane model constants: lane-major records (jxs_params.h), 16-byte aligned -- the loads of
  // consecutive fields are merged into dwordx4 / dwordx2 loads by the compiler
  template <typename U>
  static __device__ __forceinline__ const U* al16(const U* p) { return static_cast<const U*>(__builtin_assume_aligned(p, 16)); }
  __device__ __forceinline__ V lconstf(const T* tbl, int field) const { return al16(tbl)[lane_ * kLtfStride + field]; }
  // packed integer tables (jxs_params.h: lti_get / rti_get): the dword loads of one record merge into 128-bit loads
  // (indexed from the uniform table pointer: scalar-base addressing, no 64-bit vector address per record)
  __device__ __forceinline__ VI lconsti(const int* tbl, int field) const {
    return lti_unpack((unsigned)al16(tbl)[lane_ * kLtiPackWords + lti_word(field)], field);
  }
  __device__ __forceinline__ VI rconsti(const int* tbl, int field) const {
    return rti_unpack((unsigned)al16(tbl)[lane_ * kRtiPackWords + rti_word(field)], field);
  }
  __device__ __forceinline__ VI hconsti(const int* head, int chunk) const { return head[chunk * G + lane_]; }
  // one element of a table in device memory per lane, by a per-lane index (height-field samples: L2-resident)
  __device__ __forceinline__ V tgather(const T* tbl, int idx) const { return tbl[idx]; }
  // per-slot point tables (slot-major, stride 4)
  __device__ __forceinline__ V ploadf(const T* tbl, int field, int slot) const { return al16(tbl)[slot * kPtStride + field]; }
  __device__ __forceinline__ VI ploadi(const int* tbl, int field, int slot) const { return al16(tbl)[slot * kPtStride + field]; }
  // Batched arrays are tile-interleaved: [N/T][rows][T] with T = 64/G environments per tile = the
  // environments of ONE wave.  A wave therefore touches one contiguous
  // rows*T*sizeof(T) span per array and a load instruction whose lanes read consecutive rows is
  // fully coalesced.  Loads are unconditional (callers clamp the row and mask the value): no
  // exec-mask branch, and every load can be issued up front.  Arrays are allocated in whole tiles,
  // so the environments beyond N of the last tile are readable (their stores are masked).
  static constexpr int TILE = 64 / G;
  // [round 3] address = (wave-uniform start of this wave's tile, SGPRs) + (32-bit lane offset): the loads and stores
  // then use the scalar-base addressing mode and need no 64-bit vector arithmetic per access (the prologue and the
  // epilogue of the step kernel each carried ~3 such instructions per row: ~70 of 1950 vector instructions).
  template <typename U>
  __device__ __forceinline__ U* tile_base(U* base, int nrows) const {
    const unsigned b = (unsigned)__builtin_amdgcn_readfirstlane(blk_);  // (uniform by construction; tell the compiler)
    return base + (size_t)b * (size_t)((unsigned)nrows * (unsigned)TILE);
  }
  __device__ __forceinline__ unsigned lane_off(int row) const { return (unsigned)row * (unsigned)TILE + (unsigned)sub_; }
  __device__ __forceinline__ V gload(const T* base, int row, int nrows) const { return tile_base(base, nrows)[lane_off(row)]; }
  __device__ __forceinline__ V gload_u(const T* base, int row, int nrows) const { return tile_base(base, nrows)[lane_off(row)]; }
  __device__ __forceinline__ void gstore(T* base, int row, T val, bool mask, int nrows) const {
    if (mask && env_ok_) tile_base(base, nrows)[lane_off(row)] = val;
  }
};

}  // namespace jxs
