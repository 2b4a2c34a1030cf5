// Device lane backend: one link per lane, G lanes per environment, 64/G environments per
// wavefront.  Cross-lane traffic along the kinematic tree is ds_bpermute (wavefront shuffle);
// nothing goes through LDS allocations or __syncthreads -- a block is a single wave.
#pragma once
#include <hip/hip_runtime.h>

#include "jxs_params.h"

namespace jxs {

// ---- scalar "vector" primitives (V = T on the device) --------------------------------------
__device__ __forceinline__ float vsel(bool m, float a, float b) { return m ? a : b; }
__device__ __forceinline__ double vsel(bool m, double a, double b) { return m ? a : b; }
__device__ __forceinline__ int vsel(bool m, int a, int b) { return m ? a : b; }
__device__ __forceinline__ float vsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double vsqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float vabs(float x) { return fabsf(x); }
__device__ __forceinline__ double vabs(double x) { return fabs(x); }
__device__ __forceinline__ float vmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double vmin(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ double vmax(double a, double b) { return fmax(a, b); }
__device__ __forceinline__ float vpow(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double vpow(double a, double b) { return pow(a, b); }
__device__ __forceinline__ void vsincos(float x, float& s, float& c) { sincosf(x, &s, &c); }
__device__ __forceinline__ void vsincos(double x, double& s, double& c) { sincos(x, &s, &c); }
__device__ __forceinline__ float vsin(float x) { return sinf(x); }
__device__ __forceinline__ double vsin(double x) { return sin(x); }

template <typename T_, int G_>
struct DeviceLanes {
  using T = T_;
  using V = T_;
  using VI = int;
  using VM = bool;
  static constexpr int G = G_;

  int lane_;    // lane within the group
  int base4_;   // (first wave lane of the group) * 4, for ds_bpermute byte addressing
  int env_;     // environment handled by this group
  bool env_ok_;
  int N_;

  __device__ __forceinline__ DeviceLanes(int N) : N_(N) {
    const int wl = threadIdx.x & 63;
    lane_ = wl & (G - 1);
    base4_ = (wl & ~(G - 1)) << 2;
    env_ = blockIdx.x * (64 / G) + (wl / G);
    env_ok_ = env_ < N;
  }

  __device__ __forceinline__ VI lane() const { return lane_; }
  __device__ __forceinline__ VM all_true() const { return true; }

  __device__ __forceinline__ int src4(int src) const { return ((src & (G - 1)) << 2) + base4_; }
  __device__ __forceinline__ float shfl(float x, int src) const {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src4(src), __float_as_int(x)));
  }
  __device__ __forceinline__ double shfl(double x, int src) const {
    const int a = src4(src);
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_ds_bpermute(a, lo);
    hi = __builtin_amdgcn_ds_bpermute(a, hi);
    return __hiloint2double(hi, lo);
  }

  // per-lane model constants
  __device__ __forceinline__ V lconstf(const T* tbl, int field) const { return tbl[field * G + lane_]; }
  __device__ __forceinline__ VI lconsti(const int* tbl, int field) const { return tbl[field * G + lane_]; }
  // per-slot point tables
  __device__ __forceinline__ V ploadf(const T* tbl, int field, int n_slots, int slot) const {
    return tbl[field * n_slots + slot];
  }
  __device__ __forceinline__ VI ploadi(const int* tbl, int field, int n_slots, int slot) const {
    return tbl[field * n_slots + slot];
  }
  // [row][N] arrays at this group's environment
  __device__ __forceinline__ V gload(const T* base, int row, bool mask) const {
    return (mask && env_ok_) ? base[(size_t)row * N_ + env_] : T(0);
  }
  __device__ __forceinline__ V gload_u(const T* base, int row) const {
    return env_ok_ ? base[(size_t)row * N_ + env_] : T(0);
  }
  __device__ __forceinline__ void gstore(T* base, int row, T val, bool mask) const {
    if (mask && env_ok_) base[(size_t)row * N_ + env_] = val;
  }
};

}  // namespace jxs
