// Kernel entry point + launchers of one (dtype, mode): compiled once per pair in its own translation unit
// (jxs_inst.hip with -DJXS_INST_T / -DJXS_INST_MODE, see build.sh) so that the 16 units build in parallel;
// jxs_api.hip only sees the declaration of jxs_launch_g.
#pragma once
#include <hip/hip_runtime.h>

#include "jxs_lanes_device.h"
// lanes before the core: the core's unqualified calls on scalar lane values bind here
#include "jxs_core.h"

namespace jxs_launch {

// The first 16 dwords of the kernel arguments are PRELOADED into SGPRs by the command processor
// (-mllvm -amdgpu-kernarg-preload-count=16; gfx940+): the pointers and row counts every first-batch
// load address needs arrive with the wave instead of after a scalar-load round trip.  The structs that
// follow carry the same values (and everything else); the preloaded copies simply replace them.
template <typename T, int G, int MODE>
__global__ __launch_bounds__(64) void jxs_kernel(const T* pre_state_in, const T* pre_ltf, const int* pre_lti,
                                                 const T* pre_ptf, const int* pre_pti, const int* pre_head,
                                                 int pre_n_rows, int pre_n, int pre_n_slots, int pre_N,
                                                 const jxs::KParams<T> P_, const jxs::KArgs<T> A_) {
  jxs::KParams<T> P = P_;
  jxs::KArgs<T> A = A_;
  A.state_in = pre_state_in, A.ltf = pre_ltf, A.lti = pre_lti, A.ptf = pre_ptf, A.pti = pre_pti, A.head = pre_head;
  A.N = pre_N;
  P.n_rows = pre_n_rows, P.n = pre_n, P.n_slots = pre_n_slots;
  P.row_pos = 0, P.row_quat = 3, P.row_s = 7, P.row_vlin = 7 + pre_n, P.row_vang = 10 + pre_n, P.row_sd = 13 + pre_n;
  P.row_m = 13 + 2 * pre_n;  // the state-block rows of SURVEY section 8(a) row D, derived instead of loaded
  extern __shared__ __align__(16) unsigned char jxs_smem[];
  const jxs::DeviceLanes<T, G> ln(A.N, reinterpret_cast<T*>(jxs_smem),
                                  (MODE == jxs::MODE_STEP_RIGID || MODE == jxs::MODE_STEP_RK4_RIGID) ? jxs::rigid_lds_words_per_env(P.n_cp, P.rigid)
                                                               : jxs::lds_words_per_env(G));
  jxs::Core<jxs::DeviceLanes<T, G>> core(P, A, ln);
  core.template run<MODE>();
}

template <typename T, int G, int MODE>
hipError_t launch_one(const jxs::KParams<T>& P, const jxs::KArgs<T>& A, hipStream_t s) {
  const int envs_per_wave = 64 / G;  // = the tile of every batched array: block b owns tile b
  const int blocks = (A.N + envs_per_wave - 1) / envs_per_wave;
  const bool rows = P.row_mode && (MODE == jxs::MODE_STEP || MODE == jxs::MODE_ROLLOUT || MODE == jxs::MODE_FD ||
                                   MODE == jxs::MODE_STEP_RK4);
  size_t lds_bytes = rows ? sizeof(T) * (size_t)envs_per_wave * jxs::lds_words_per_env(G) : 0;
  if (MODE == jxs::MODE_STEP_RIGID || MODE == jxs::MODE_STEP_RK4_RIGID) {
    lds_bytes = sizeof(T) * (size_t)envs_per_wave * jxs::rigid_lds_words_per_env(P.n_cp, P.rigid);
    if (lds_bytes > 64 * 1024) {  // beyond the default dynamic-LDS window (gfx950 has 160 KiB per CU)
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&jxs_kernel<T, G, MODE>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return e;
    }
  }
  hipLaunchKernelGGL((jxs_kernel<T, G, MODE>), dim3(blocks), dim3(64), lds_bytes, s, A.state_in, A.ltf, A.lti, A.ptf, A.pti,
                     A.head, P.n_rows, P.n, P.n_slots, A.N, P, A);
  return hipGetLastError();
}

template <typename T, int MODE>
hipError_t launch_g(int G, const jxs::KParams<T>& P, const jxs::KArgs<T>& A, hipStream_t s) {
  switch (G) {
    case 4: return launch_one<T, 4, MODE>(P, A, s);
    case 8: return launch_one<T, 8, MODE>(P, A, s);
    case 16: return launch_one<T, 16, MODE>(P, A, s);
    case 32: return launch_one<T, 32, MODE>(P, A, s);
    default: return launch_one<T, 64, MODE>(P, A, s);
  }
}

}  // namespace jxs_launch
