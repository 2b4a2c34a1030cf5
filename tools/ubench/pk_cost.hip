// Developer micro-benchmark: packed fp32 math vs scalar for ONE resident wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
__global__ void k(float* out, long long* t, int mode) {
  f2 a[8]; float b[8];
  for (int i = 0; i < 8; ++i) { a[i].x = out[threadIdx.x + 64 * i]; a[i].y = a[i].x + 1.f; b[i] = a[i].x; }
  const f2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  long long t0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < 64; ++it) {
    if (mode == 0) { REP8(for (int i = 0; i < 8; ++i) b[i] = b[i] * 1.0001f + 0.5f;) }
    if (mode == 1) { REP8(for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);) }
    if (mode == 2) { REP8(for (int i = 0; i < 8; ++i) a[i] = a[i] * m;) }
    if (mode == 3) { REP8(for (int i = 0; i < 8; ++i) a[i] = a[i] + c;) }
    if (mode == 4) { REP8(for (int i = 0; i < 8; ++i) b[i] = (b[i] > 1.0f) ? b[i] * 0.5f : b[i] + 0.25f;) }  // cmp + cndmask + 2 ops
  }
  __builtin_amdgcn_sched_barrier(0);
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y + b[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) t[mode] = t1 - t0;
}
int main() {
  float* d; long long* t; hipMalloc(&d, 64 * 8 * 4); hipMalloc(&t, 8 * 8); hipMemset(d, 0, 64 * 8 * 4);
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "select pattern"};
  for (int m = 0; m < 5; ++m) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t, m); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t, m); }
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
  for (int m = 0; m < 5; ++m) printf("%-16s %6.2f cycles per source op (4096 ops, one wave)\n", names[m], (double)h[m] / 4096.0);
  return 0;
}
