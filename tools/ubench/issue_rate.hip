// Developer micro-benchmark: what ONE resident wave pays per instruction, by instruction kind.
// Question behind it: the step kernel runs one wave per SIMD at N = 1024 -- is it bound by the number of
// VALU instructions, by dependent-issue latency, or by the total instruction count (SALU, waits, nops too)?
//   part 1: v_fma_f32 chains with ILP 1..8, alone and with a second wave on the SIMD
//   part 2: 2048 x {v_fma_f32 + X} for X in {nothing, s_nop 0, s_waitcnt (satisfied), s_add_u32, v_cndmask,
//           v_pk_fma_f32, v_add_f32 dpp, v_rcp_f32, ds_bpermute (no wait), ds_read_b32 (no wait),
//           ds_read_b128 (no wait), s_cbranch (not taken)}: the difference to "nothing" is the cost of X
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ void k(float* out, long long* t, int slot) {
  float b[ILP];
  for (int i = 0; i < ILP; ++i) b[i] = out[threadIdx.x + 64 * i];
  long long t0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < 4096 / ILP / 16; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(1.0001f), "v"(0.5f));
  }
  __builtin_amdgcn_sched_barrier(0);
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < ILP; ++i) s += b[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) t[slot] = t1 - t0;
}
#define X16(S) S S S S S S S S S S S S S S S S
template <int KIND>
__global__ void kx(float* out, long long* t, int slot, int stride = 16) {
  __shared__ float lds[4096];
  lds[threadIdx.x] = out[threadIdx.x];
  float a = out[threadIdx.x], b = a + 1.f, c = b + 1.f, d = c + 1.f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef float f4 __attribute__((ext_vector_type(4)));
  f2 p = {a, b};
  f4 q = {a, b, c, d};
  int addr = (threadIdx.x * stride) & 16383, sreg = 0;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < 128; ++it) {
#define FMA "v_fma_f32 %0, %0, %6, %7\n"
#define BODY(X) asm volatile(X16(FMA X) : "+v"(a), "+v"(b), "+v"(p), "+v"(q), "+s"(sreg), "+v"(c) : "v"(1.0001f), "v"(0.5f), "v"(addr) : "vcc", "scc", "memory")
    if (KIND == 0) BODY("");
    if (KIND == 1) BODY("s_nop 0\n");
    if (KIND == 2) BODY("s_waitcnt lgkmcnt(0)\n");
    if (KIND == 3) BODY("s_add_u32 %4, %4, 1\n");
    if (KIND == 4) BODY("v_cndmask_b32 %1, %1, %0, vcc\n");
    if (KIND == 5) BODY("v_pk_fma_f32 %2, %2, %2, %2\n");
    if (KIND == 6) BODY("v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n");
    if (KIND == 7) BODY("v_rcp_f32 %1, %1\n");
    if (KIND == 8) { BODY("ds_bpermute_b32 %5, %8, %1\n"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    if (KIND == 9) { BODY("ds_read_b32 %5, %8\n"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    if (KIND == 10) { BODY("ds_read_b128 %3, %8\n"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    if (KIND == 11) BODY("s_cmp_eq_u32 %4, -1\ns_cbranch_scc1 1f\n1:\n");
    if (KIND == 12) BODY("s_nop 1\n");
    if (KIND == 13) BODY("v_mov_b32 %1, %0\n");
    if (KIND == 14) BODY("ds_bpermute_b32 %0, %8, %0\ns_waitcnt lgkmcnt(0)\n");
    if (KIND == 15) BODY("ds_write_b32 %8, %0\nds_read_b32 %0, %8\ns_waitcnt lgkmcnt(0)\n");
    if (KIND == 16) BODY("ds_write_b128 %8, %3\nds_read_b128 %3, %8\ns_waitcnt lgkmcnt(0)\n");
    if (KIND == 17) BODY("ds_read_b32 %0, %8\ns_waitcnt lgkmcnt(0)\n");
    if (KIND == 18) BODY("ds_read_b128 %3, %8\ns_waitcnt lgkmcnt(0)\n");
    if (KIND == 19) BODY("ds_bpermute_b32 %0, %8, %0\nds_bpermute_b32 %1, %8, %1\nds_bpermute_b32 %5, %8, %5\ns_waitcnt lgkmcnt(0)\n");
  }
  __builtin_amdgcn_sched_barrier(0);
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = a + b + p.x + p.y + q.x + q.w + c + sreg;
  if (threadIdx.x == 0) t[slot] = t1 - t0;
}
int main() {
  float* d; long long* t; hipMalloc(&d, 4096 * 8 * 4); hipMalloc(&t, 64 * 8); hipMemset(d, 0, 4096 * 8 * 4);
  int slot = 0;
  for (int bd : {64, 320}) {  // 320 threads: five waves, SIMD 0 holds two of them
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k<1>, dim3(1), dim3(bd), 0, 0, d, t, slot + 0);
      hipLaunchKernelGGL(k<2>, dim3(1), dim3(bd), 0, 0, d, t, slot + 1);
      hipLaunchKernelGGL(k<4>, dim3(1), dim3(bd), 0, 0, d, t, slot + 2);
      hipLaunchKernelGGL(k<8>, dim3(1), dim3(bd), 0, 0, d, t, slot + 3);
    }
    slot += 4;
  }
  for (int rep = 0; rep < 2; ++rep) {
#define L(K) hipLaunchKernelGGL(kx<K>, dim3(1), dim3(64), 0, 0, d, t, 8 + K);
    L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19)
  }
  const int strides[] = {16, 32, 48, 64, 80, 84, 96, 128};
  for (int rep = 0; rep < 2; ++rep)
    for (int i = 0; i < 8; ++i) {
      hipLaunchKernelGGL(kx<16>, dim3(1), dim3(64), 0, 0, d, t, 32 + i, strides[i]);
      hipLaunchKernelGGL(kx<10>, dim3(1), dim3(64), 0, 0, d, t, 40 + i, strides[i]);
    }
  hipDeviceSynchronize();
  long long h[64]; hipMemcpy(h, t, 64 * 8, hipMemcpyDeviceToHost);
  const int ilp[] = {1, 2, 4, 8};
  for (int s = 0; s < 8; ++s)
    printf("%s  ILP %d: %5.2f ticks per v_fma_f32\n", s < 4 ? "1 wave on the SIMD " : "2 waves on the SIMD", ilp[s % 4], (double)h[s] / 4096.0);
  const char* names[] = {"(v_fma_f32 alone)", "s_nop 0", "s_waitcnt (satisfied)", "s_add_u32", "v_cndmask_b32", "v_pk_fma_f32", "v_add_f32 dpp",
                         "v_rcp_f32", "ds_bpermute_b32 (16 in flight)", "ds_read_b32 (16 in flight)", "ds_read_b128 (16 in flight)", "s_cmp + s_cbranch (not taken)",
                         "s_nop 1", "v_mov_b32", "ds_bpermute + wait (dependent)", "ds_write_b32 + ds_read_b32 + wait", "ds_write_b128 + ds_read_b128 + wait",
                         "ds_read_b32 + wait", "ds_read_b128 + wait", "3 x ds_bpermute + wait"};
  const double base = (double)h[8] / 2048.0;
  printf("dependent v_fma_f32: %.2f ticks each\n", base);
  for (int kx_ = 1; kx_ < 20; ++kx_) printf("  + %-32s %6.2f ticks\n", names[kx_], (double)h[8 + kx_] / 2048.0 - base);
  for (int i = 0; i < 8; ++i)
    printf("lane stride %3d B: ds_write_b128 + ds_read_b128 + wait %6.2f ticks, ds_read_b128 (16 in flight) %6.2f ticks\n", strides[i],
           (double)h[32 + i] / 2048.0 - base, (double)h[40 + i] / 2048.0 - base);
  return 0;
}
