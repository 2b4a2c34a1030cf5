// Developer micro-benchmark: issue cost of DPP / plain VALU / readlane / bpermute for ONE resident wave.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/dpp_cost.hip -o /tmp/dpp_cost && /tmp/dpp_cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
#define REP8(X) X X X X X X X X
__global__ void k(float* out, long long* t, int mode) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = out[threadIdx.x + 64 * i];
  long long t0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < 64; ++it) {
    if (mode == 0) { REP8(for (int i = 0; i < 8; ++i) a[i] = a[i] * 1.0001f + 0.5f;) }           // 64 independent-ish FMAs
    if (mode == 1) { REP8(for (int i = 0; i < 8; ++i) a[i] = a[i] + dpp<0xB1>(a[i]);) }          // quad_perm
    if (mode == 2) { REP8(for (int i = 0; i < 8; ++i) a[i] = a[i] + dpp<0x141>(a[i]);) }         // row_half_mirror
    if (mode == 3) { REP8(for (int i = 0; i < 8; ++i) a[i] = a[i] + dpp<0x130>(a[i]);) }         // wave_shl:1
    if (mode == 4) { REP8(for (int i = 0; i < 8; ++i) a[i] = a[i] + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[i]), 5));) }
    if (mode == 5) { REP8(for (int i = 0; i < 8; ++i) a[i] = a[i] + __int_as_float(__builtin_amdgcn_ds_bpermute((threadIdx.x ^ 1) << 2, __float_as_int(a[i])));) }
    if (mode == 6) { REP8(for (int i = 0; i < 8; ++i) a[i] = a[i] + dpp<0x140>(a[i]);) }         // row_mirror
  }
  __builtin_amdgcn_sched_barrier(0);
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) t[mode] = t1 - t0;
}
int main() {
  float* d; long long* t; hipMalloc(&d, 64 * 8 * 4); hipMalloc(&t, 8 * 8); hipMemset(d, 0, 64 * 8 * 4);
  const char* names[] = {"v_fma", "dpp quad_perm add", "dpp row_half_mirror add", "dpp wave_shl add", "v_readlane + add", "ds_bpermute + add", "dpp row_mirror add"};
  for (int m = 0; m < 7; ++m) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t, m); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t, m); }
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
  for (int m = 0; m < 7; ++m) printf("%-26s %6.2f cycles per op (4096 ops, one wave)\n", names[m], (double)h[m] / 4096.0);
  return 0;
}
