// Developer micro-benchmark: on which SIMDs do the waves of a small workgroup land?  (HW_ID register)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = hw;
  // keep the waves resident for a while so that the dispatcher cannot reuse slots
  float a = threadIdx.x;
  for (int i = 0; i < 20000; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) out[0] = 0;
}
int main() {
  for (int waves = 1; waves <= 4; waves *= 2) {
    const int blocks = 512;
    unsigned* d; hipMalloc(&d, blocks * waves * 4);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 0, 0, d);
    hipDeviceSynchronize();
    unsigned h[2048]; hipMemcpy(h, d, blocks * waves * 4, hipMemcpyDeviceToHost);
    int same = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 1; w < waves; ++w)
        if (((h[b * waves + w] >> 4) & 3) == ((h[b * waves] >> 4) & 3)) ++same;
    printf("workgroups of %d wave(s): first 4 workgroups (simd ids):", waves);
    for (int b = 0; b < 4; ++b) { printf(" ["); for (int w = 0; w < waves; ++w) printf("%u", (h[b * waves + w] >> 4) & 3); printf("]"); }
    printf("  waves sharing the SIMD of wave 0: %d of %d\n", same, blocks * (waves - 1));
    hipFree(d);
  }
  return 0;
}
