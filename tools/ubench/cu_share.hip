// Developer micro-benchmark (round 3): which per-CU resources do the waves of small grids share?
// For grids of B workgroups x W waves: where the waves land (XCC / SE / CU / SIMD from HW_ID, XCC_ID) and what ONE
// wave pays per instruction of a stream of  (a) v_fma_f32  (b) ds_bpermute_b32  (c) ds_read_b128
// (d) global_load_dwordx4 from a 32 KB table (cache hits)  while the other waves of the grid run the same stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#define X16(S) S S S S S S S S S S S S S S S S
template <int KIND>
__global__ void k(const float* tab, float* out, long long* t, unsigned* hw) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  lds[tid] = tab[tid];
  float a = tab[tid & 1023], b = a + 1.f;
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 q = {a, b, a, b};
  int addr = ((tid & 63) * 16) & 8191;
  const float* gp = tab + ((tid & 63) * 4);
  __syncthreads();
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID [3:0]
  long long t0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < 64; ++it) {
    if (KIND == 0) asm volatile(X16("v_fma_f32 %0, %0, %2, %3\n") : "+v"(a), "+v"(b) : "v"(1.0001f), "v"(0.5f));
    if (KIND == 1) { asm volatile(X16("ds_bpermute_b32 %1, %2, %0\n") : "+v"(a), "+v"(b) : "v"(addr) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    if (KIND == 2) { asm volatile(X16("ds_read_b128 %0, %1\n") : "+v"(q) : "v"(addr) : "memory"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    if (KIND == 3) { asm volatile(X16("global_load_dwordx4 %0, %1, off\n") : "+v"(q) : "v"(gp) : "memory"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    if (KIND == 4) { asm volatile(X16("global_load_dword %0, %1, off\n") : "+v"(a) : "v"(gp) : "memory"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  }
  __builtin_amdgcn_sched_barrier(0);
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + tid] = a + b + q.x + q.w;
  if ((tid & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + (tid >> 6);
    t[w] = t1 - t0;
    hw[w] = id | (xcc << 16);
  }
}
int main() {
  float *tab, *out; long long* t; unsigned* hw;
  hipMalloc(&tab, 65536); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&t, 4096 * 8); hipMalloc(&hw, 4096 * 4);
  hipMemset(tab, 0, 65536);
  const char* names[] = {"v_fma_f32", "ds_bpermute_b32", "ds_read_b128", "global_load_dwordx4 (cached)", "global_load_dword (cached)"};
  const int cfgs[][2] = {{256, 1}, {512, 1}, {1024, 1}, {256, 2}, {512, 2}, {256, 4}, {512, 4}, {2048, 1}};
  for (auto& c : cfgs) {
    const int B = c[0], W = c[1], n = B * W;
    for (int kind = 0; kind < 5; ++kind) {
      for (int rep = 0; rep < 2; ++rep) {
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(B), dim3(64 * W), 8192 * 4, 0, tab, out, t, hw);
        if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(B), dim3(64 * W), 8192 * 4, 0, tab, out, t, hw);
        if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(B), dim3(64 * W), 8192 * 4, 0, tab, out, t, hw);
        if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(B), dim3(64 * W), 8192 * 4, 0, tab, out, t, hw);
        if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(B), dim3(64 * W), 8192 * 4, 0, tab, out, t, hw);
      }
      hipDeviceSynchronize();
      std::vector<long long> ht(n); std::vector<unsigned> hh(n);
      hipMemcpy(ht.data(), t, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hh.data(), hw, n * 4, hipMemcpyDeviceToHost);
      double mean = 0; long long mx = 0;
      for (int i = 0; i < n; ++i) { mean += ht[i]; if (ht[i] > mx) mx = ht[i]; }
      mean /= n;
      std::map<unsigned, int> percu, persimd;
      for (int i = 0; i < n; ++i) {
        const unsigned h = hh[i];
        const unsigned cu = ((h >> 16) & 15) << 12 | ((h >> 13) & 7) << 8 | ((h >> 12) & 1) << 4 | ((h >> 8) & 15);  // xcc, se, sh, cu
        percu[cu]++; persimd[cu << 2 | ((h >> 4) & 3)]++;
      }
      int mcu = 0, msimd = 0;
      for (auto& p : percu) if (p.second > mcu) mcu = p.second;
      for (auto& p : persimd) if (p.second > msimd) msimd = p.second;
      if (kind == 0) printf("grid %4d x %d wave(s): %zu CUs used, max %d waves per CU, %zu SIMDs used, max %d waves per SIMD\n", B, W, percu.size(), mcu, persimd.size(), msimd);
      printf("    %-30s %7.2f ticks per instruction per wave (slowest wave %7.2f)\n", names[kind], mean / 1024.0, (double)mx / 1024.0);
    }
  }
  return 0;
}
