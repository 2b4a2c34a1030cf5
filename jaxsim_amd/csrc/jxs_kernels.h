// Kernel entry point + launchers of one (dtype, mode): compiled once per pair in its own translation unit
// (jxs_inst.hip with -DJXS_INST_T / -DJXS_INST_MODE, see build.sh) so that the 16 units build in parallel;
// jxs_api.hip only sees the declaration of jxs_launch_g.
#pragma once
#include <hip/hip_runtime.h>

#include "jxs_lanes_device.h"
// lanes before the core: the core's unqualified calls on scalar lane values bind here
#include <cstdlib>

#include "jxs_core.h"

namespace jxs_launch {

// Kernel arguments.  The first 16 dwords are PRELOADED into SGPRs by the command processor
// (-mllvm -amdgpu-kernarg-preload-count=16; gfx940+) and arrive with the wave: the state pointers, the
// inputs of `step`, the batch size, and ONE pointer to the device model block (jxs_params.h) that holds
// the wave-uniform parameters and every table.  `step` never reads the kernarg segment itself: measured
// with arrival stamps, a scalar load from it costs a lone wave ~1600 cycles (it is host-visible memory),
// while the model block is ordinary device memory (L2-resident after the first wave).  The arguments of
// the other entry points (accelerations in / out, kinematics out) stay in the tail struct.
template <typename T>
struct KTail {
  const T* in_a;
  T* out_a;
  T* out_H;
  T* out_V;
  T* out_tau;
  int id_zero_vel;
  long long* dbg;
  int* faults;
  int flags;
  T fparam;
};

// OCC2 (rigid contact modes only): compiled for two waves per SIMD (at most 256 registers; a few values go
// to scratch).  Measured on the quadruped with one point per foot: -2 % at 4096 environments (one wave per
// SIMD either way), +26 % at 16384, +58 % at 65536 -- the launcher picks it when the grid has more waves than
// the chip has SIMDs and the LDS footprint of the contact problem lets a second wave in.
//
// COMMON (soft-contact step / rollout only): the feature switches of the most common kind of model -- floating
// base, no joint-successor transforms, row-distributed ABA, flat terrain, p = q = 1/2, anchored chains, one
// contact chunk -- as constants; the launcher picks it when the model's flags say exactly that.  It is the
// part of a model-specialised build (jxs_spec.hip) that needs no knowledge of the tree: 9.74 -> 8.55 us per
// step for the 23-DoF humanoid where the full specialisation reaches 7.68 (tools/spec_split_experiment.py).
enum KernelVariant : int { KV_GENERIC = 0, KV_OCC2 = 1, KV_COMMON = 2 };
#ifndef JXS_MIN_WAVES
#define JXS_MIN_WAVES 1  // experiment knob (JAXSIM_AMD_SPEC_EXTRA_FLAGS=-DJXS_MIN_WAVES=4): register budget for that many waves per SIMD
#endif
template <typename T, int G, int MODE, int VARIANT = KV_GENERIC>
__global__ __launch_bounds__(64, VARIANT == KV_OCC2 ? 2 : JXS_MIN_WAVES) void jxs_kernel(const T* pre_state_in, T* pre_state_out, const unsigned char* __restrict__ pre_mblk,
                                                 const T* pre_tau, const T* pre_link_f, int pre_N, int pre_n_rows,
                                                 int pre_n, int pre_force_repr, int pre_n_steps,
                                                 const KTail<T> tail) {
  // (14 dwords are preloaded in practice: the last two ints are only needed late -- with link forces / in
  // the fused rollout -- and may come from the kernarg segment)
  // wave-uniform parameters: scalar loads from the model block (uniform address, never written by the
  // kernel: `__restrict__` lets the compiler prove it and use s_load)
  jxs::KParams<T> P = *reinterpret_cast<const jxs::KParams<T>*>(pre_mblk);
  jxs::KArgs<T> A{};
  A.state_in = pre_state_in, A.state_out = pre_state_out, A.tau = pre_tau, A.link_f = pre_link_f;
#ifdef JXS_ASSUME_NO_INPUTS  // analysis aid (tools/spec_isa.py): the instruction stream without the optional inputs
  A.tau = nullptr, A.link_f = nullptr;
#endif
  A.N = pre_N, A.force_repr = pre_force_repr, A.n_steps = pre_n_steps;
  A.ltf = reinterpret_cast<const T*>(pre_mblk + jxs::mblk_off_ltf<T>());
  A.lti = reinterpret_cast<const int*>(pre_mblk + jxs::mblk_off_lti<T>(G));
  A.rti = reinterpret_cast<const int*>(pre_mblk + jxs::mblk_off_rti<T>(G));
  A.chunks = pre_mblk + jxs::mblk_off_chunks<T>(G);
  A.hf = reinterpret_cast<const T*>(pre_mblk + P.hf_off);
  A.in_a = tail.in_a, A.out_a = tail.out_a, A.out_H = tail.out_H, A.out_V = tail.out_V, A.out_tau = tail.out_tau;
  A.id_zero_vel = tail.id_zero_vel, A.dbg = tail.dbg, A.faults = tail.faults, A.flags = tail.flags, A.fparam = tail.fparam;
  if constexpr (VARIANT == KV_OCC2) A.flags |= 1;  // two waves per SIMD: 256 registers, no room for MFMA accumulator tiles
  // the state-block rows of SURVEY section 8(a) row D, derived from the preloaded joint count instead of loaded
  P.n_rows = pre_n_rows, P.n = pre_n;
  P.row_pos = 0, P.row_quat = 3, P.row_s = 7, P.row_vlin = 7 + pre_n, P.row_vang = 10 + pre_n, P.row_sd = 13 + pre_n;
  P.row_m = 13 + 2 * pre_n;
  if constexpr (VARIANT == KV_COMMON) {
    P.floating = 1, P.any_suc = 0, P.seg_dpp_ok = 1, P.row_mode = 1, P.flat = 1, P.pq_half = 1, P.anchored = 1, P.rigid = 0;
    P.rk4fast = 0, P.n_chunks = 1, P.hf = 0;
  }
#ifdef JXS_SPEC_ASSIGN
  // model-specialised build (jxs_spec.hip): the integer model flags are compile-time constants from here on
  JXS_SPEC_ASSIGN;
  constexpr bool kFlagsKnown = true;
#else
  constexpr bool kFlagsKnown = VARIANT == KV_COMMON;
#endif
  // (launch_one allocates the LDS area of the row layout for exactly these modes when P.row_mode is set; only where
  // that flag is a compile-time constant, so that the first loads of the generic kernel need no scalar load)
#ifdef JXS_SPEC_ASSIGN
  A.spec_consts = 1;
#else
  A.spec_consts = 0;
#endif
  A.has_lds = (kFlagsKnown && P.row_mode && (MODE == jxs::MODE_STEP || MODE == jxs::MODE_ROLLOUT || MODE == jxs::MODE_FD || MODE == jxs::MODE_STEP_RK4 || MODE == jxs::MODE_DYN)) ? 1 : 0;
  extern __shared__ __align__(16) unsigned char jxs_smem[];
  const jxs::DeviceLanes<T, G> ln(A.N, reinterpret_cast<T*>(jxs_smem),
                                  (MODE == jxs::MODE_STEP_RIGID || MODE == jxs::MODE_STEP_RK4_RIGID || MODE == jxs::MODE_DYN_RIGID) ? jxs::rigid_lds_words_per_env(P.n_cp, P.rigid, P.ct_tree, P.n_chunks, G)
                                  : MODE == jxs::MODE_STEP_RK4 ? jxs::rk4_lds_words_per_env(G, P.n_chunks)
                                                               : jxs::lds_rows_words(G, P.nL));
  jxs::Core<jxs::DeviceLanes<T, G>> core(P, A, ln);
  core.template run<MODE>();
}

// Two-wave workgroups (jxs_core.h, "Two-wave workgroups"): a main wave and an inertia wave work on the same
// 64 / G environments; a workgroup holds two such pairs (see DeviceLanes: even placement).  Picked by the launcher for the soft-contact step of row-layout models while the grid
// leaves SIMDs idle (one wave per workgroup fills at most blocks of the chip's 1024 SIMDs).
template <typename T, int G, int MODE>
__global__ __launch_bounds__(256, 1) void jxs_kernel_duo(const T* pre_state_in, T* pre_state_out, const unsigned char* __restrict__ pre_mblk,
                                                       const T* pre_tau, const T* pre_link_f, int pre_N, int pre_n_rows,
                                                       int pre_n, int pre_force_repr, int pre_n_steps,
                                                       const KTail<T> tail) {
  jxs::KParams<T> P = *reinterpret_cast<const jxs::KParams<T>*>(pre_mblk);
  jxs::KArgs<T> A{};
  A.state_in = pre_state_in, A.state_out = pre_state_out, A.tau = pre_tau, A.link_f = pre_link_f;
#ifdef JXS_ASSUME_NO_INPUTS  // analysis aid (tools/spec_isa.py): the instruction stream without the optional inputs
  A.tau = nullptr, A.link_f = nullptr;
#endif
  A.N = pre_N, A.force_repr = pre_force_repr, A.n_steps = pre_n_steps;
  A.ltf = reinterpret_cast<const T*>(pre_mblk + jxs::mblk_off_ltf<T>());
  A.lti = reinterpret_cast<const int*>(pre_mblk + jxs::mblk_off_lti<T>(G));
  A.rti = reinterpret_cast<const int*>(pre_mblk + jxs::mblk_off_rti<T>(G));
  A.chunks = pre_mblk + jxs::mblk_off_chunks<T>(G);
  A.hf = reinterpret_cast<const T*>(pre_mblk + P.hf_off);
  A.in_a = tail.in_a, A.out_a = tail.out_a, A.out_H = tail.out_H, A.out_V = tail.out_V, A.out_tau = tail.out_tau;
  A.id_zero_vel = tail.id_zero_vel, A.dbg = tail.dbg, A.faults = tail.faults, A.flags = tail.flags, A.fparam = tail.fparam;
  P.n_rows = pre_n_rows, P.n = pre_n;
  P.row_pos = 0, P.row_quat = 3, P.row_s = 7, P.row_vlin = 7 + pre_n, P.row_vang = 10 + pre_n, P.row_sd = 13 + pre_n;
  P.row_m = 13 + 2 * pre_n;
  // the launcher picks this kernel only for models with exactly these features
  P.row_mode = 1, P.rigid = 0, P.rk4fast = 0;
  A.has_lds = 1;
#ifdef JXS_SPEC_ASSIGN
  JXS_SPEC_ASSIGN;
#endif
  extern __shared__ __align__(16) unsigned char jxs_smem[];
  // four waves = two pairs (main, inertia) on consecutive tiles; the second pair of the last workgroup may be empty
  const jxs::DeviceLanes<T, G> ln(A.N, reinterpret_cast<T*>(jxs_smem), jxs::duo_words_per_env(G), true);
  const bool inertia_wave = (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) & 1) != 0;
  ln.flag_reset_and_barrier(inertia_wave);
  if (__builtin_amdgcn_readfirstlane(ln.blk_) * (64 / G) >= A.N) return;
  jxs::Core<jxs::DeviceLanes<T, G>> core(P, A, ln);
  if (inertia_wave)
    core.run_inertia();
  else
    core.template run<MODE, jxs::ROLE_MAIN>();
}
template <typename T>
inline size_t duo_lds_bytes(int G) { return 2 * sizeof(T) * ((size_t)(64 / G) * jxs::duo_words_per_env(G) + jxs::kDuoFlagWords); }

// `mblk`: device model block of the model (KParams | tables), `P`: its host copy (launch geometry only)
template <typename T, int G, int MODE>
hipError_t launch_one(const jxs::KParams<T>& P, const unsigned char* mblk, const jxs::KArgs<T>& A, hipStream_t s) {
  const int envs_per_wave = 64 / G;  // = the tile of every batched array: block b owns tile b
  const int blocks = (A.N + envs_per_wave - 1) / envs_per_wave;
  const bool rows = P.row_mode && (MODE == jxs::MODE_STEP || MODE == jxs::MODE_ROLLOUT || MODE == jxs::MODE_FD ||
                                   MODE == jxs::MODE_STEP_RK4 || MODE == jxs::MODE_DYN);
  size_t lds_bytes = rows ? sizeof(T) * (size_t)envs_per_wave * jxs::lds_rows_words(G, P.nL) : 0;
  // RungeKutta4 with several point chunks keeps its per-slot stage data in the LDS (jxs_params.h rk4_lds_words_per_env)
  // (... and its row layout sits in an area of the G-wide upper bound, behind which the chunk data start: the kernel's
  // words per environment are rk4_lds_words_per_env in both cases)
  if (MODE == jxs::MODE_STEP_RK4 && (rows || P.n_chunks > 1)) lds_bytes = sizeof(T) * (size_t)envs_per_wave * jxs::rk4_lds_words_per_env(G, P.n_chunks);
  // (developer knobs arrive in A.knobs: the library reads the environment once, jxs_api.hip debug_knobs -- no getenv on the
  // launch path, no race with a Python thread that edits os.environ)
#ifdef JXS_EXP_LDS_BYTES  // developer TIMING experiment only (results are garbage): allocate this many bytes whatever the kernel uses
  if (rows) lds_bytes = JXS_EXP_LDS_BYTES;
#endif
  const KTail<T> tail{A.in_a, A.out_a, A.out_H, A.out_V, A.out_tau, A.id_zero_vel, A.dbg, A.faults,
                      A.flags | ((A.knobs & jxs::KNOB_NO_MFMA) ? 1 : 0), A.fparam};  // KNOB_NO_MFMA: A/B of the vector path of the contact solvers' Cholesky
  if (MODE == jxs::MODE_STEP_RIGID || MODE == jxs::MODE_STEP_RK4_RIGID || MODE == jxs::MODE_DYN_RIGID) {
    lds_bytes = sizeof(T) * (size_t)envs_per_wave * jxs::rigid_lds_words_per_env(P.n_cp, P.rigid, P.ct_tree, P.n_chunks, G);
    if (lds_bytes > 64 * 1024) {  // beyond the default dynamic-LDS window (gfx950 has 160 KiB per CU)
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&jxs_kernel<T, G, MODE>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return e;
    }
    // two waves per SIMD pay off once there are more waves than SIMDs (256 CUs x 4) and eight of them fit
    // the LDS of a CU (160 KiB)
    // [round 4] semi-implicit Euler only: the RungeKutta4 variant squeezed into 256 registers spills to scratch, and hipcc
    // (ROCm 7.2) placed those spills in front of the EXEC restore of a join block (jaxsim_amd/isa_lint.py, rule
    // masked_join -- the build fails on it); the one-wave kernel of that mode has the registers it needs.
    constexpr bool kHasOcc2 = MODE == jxs::MODE_STEP_RIGID && sizeof(T) == 4;
    if (kHasOcc2 && blocks > 1024 && lds_bytes * 8 <= 160 * 1024 && P.n_cp <= 8) {
      hipLaunchKernelGGL((jxs_kernel<T, G, MODE, kHasOcc2 ? KV_OCC2 : KV_GENERIC>), dim3(blocks), dim3(64), lds_bytes, s, A.state_in, A.state_out, mblk,
                         A.tau, A.link_f, A.N, P.n_rows, P.n, A.force_repr, A.n_steps, tail);
      return hipGetLastError();
    }
  }
  // Two-wave workgroups (jxs_core.h): OPT-IN (JXS_DUO=1) after the round-3 measurement -- at 1024 environments the
  // variant runs at parity with the single-wave kernel (7.70 vs 7.65 us, profiles/r03_two_wave_experiment.md): the
  // inertia recursion stays the critical path and the waves of a CU share its LDS and vector-memory pipelines.
  // JXS_DUO_MAX_BLOCKS bounds the grids it is used for.
#ifdef JXS_WITH_DUO  // [round 4] compiled into the library only (csrc/build.sh): a model-specialised object no longer carries
                     // the kernel of an experiment that lost (a second kernel per object: compile time on first use)
  if constexpr (MODE == jxs::MODE_STEP && G >= 8) {
    const int duo_env = (A.knobs & jxs::KNOB_DUO) ? 1 : 0;
    const int duo_max_blocks = A.duo_max_blocks > 0 ? A.duo_max_blocks : (1 << 30);
    const bool fits = P.row_mode == 1 && P.rigid == 0 && P.n_chunks <= 1 && duo_lds_bytes<T>(G) <= (size_t)160 * 1024;
    if (fits && duo_env > 0 && blocks <= duo_max_blocks) {
      const size_t bytes = duo_lds_bytes<T>(G);
      static bool attr_set[64] = {};  // per device: a function attribute belongs to the device's copy of the kernel
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&jxs_kernel_duo<T, G, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
      }
      hipLaunchKernelGGL((jxs_kernel_duo<T, G, MODE>), dim3((blocks + 1) / 2), dim3(256), bytes, s, A.state_in, A.state_out, mblk, A.tau,
                         A.link_f, A.N, P.n_rows, P.n, A.force_repr, A.n_steps, tail);
      return hipGetLastError();
    }
  }
#endif
#ifndef JXS_SPEC_ASSIGN  // (a model-specialised build has these constants anyway)
  constexpr bool kHasCommon = (MODE == jxs::MODE_STEP || MODE == jxs::MODE_ROLLOUT) && G >= 8;
  const bool common_off = (A.knobs & jxs::KNOB_NO_COMMON_VARIANT) != 0;  // developer knob: A/B against KV_GENERIC
  if (kHasCommon && !common_off && P.floating == 1 && P.any_suc == 0 && P.seg_dpp_ok == 1 && P.row_mode == 1 && P.flat == 1 && P.hf == 0 && P.pq_half == 1 &&
      P.anchored == 1 && P.rigid == 0 && P.rk4fast == 0 && P.n_chunks == 1) {
    hipLaunchKernelGGL((jxs_kernel<T, G, MODE, kHasCommon ? KV_COMMON : KV_GENERIC>), dim3(blocks), dim3(64), lds_bytes, s, A.state_in,
                       A.state_out, mblk, A.tau, A.link_f, A.N, P.n_rows, P.n, A.force_repr, A.n_steps, tail);
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL((jxs_kernel<T, G, MODE>), dim3(blocks), dim3(64), lds_bytes, s, A.state_in, A.state_out, mblk, A.tau,
                     A.link_f, A.N, P.n_rows, P.n, A.force_repr, A.n_steps, tail);
  return hipGetLastError();
}

template <typename T, int MODE>
hipError_t launch_g(int G, const jxs::KParams<T>& P, const unsigned char* mblk, const jxs::KArgs<T>& A, hipStream_t s) {
  switch (G) {
    case 4: return launch_one<T, 4, MODE>(P, mblk, A, s);
    case 8: return launch_one<T, 8, MODE>(P, mblk, A, s);
    case 16: return launch_one<T, 16, MODE>(P, mblk, A, s);
    case 32: return launch_one<T, 32, MODE>(P, mblk, A, s);
    default: return launch_one<T, 64, MODE>(P, mblk, A, s);
  }
}

}  // namespace jxs_launch
