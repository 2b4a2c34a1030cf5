// Developer micro-benchmark: latency of DEPENDENT cross-lane reductions for ONE resident wave (the base-to-leaves chain of
// ABA pass 3 is one 8-lane reduction per tree level, each depending on the previous level's result).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/dpp_chain.hip -o /tmp/dpp_chain && /tmp/dpp_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))
__global__ void k(float* out, long long* t, int mode) {
  float x = out[threadIdx.x], y = out[threadIdx.x + 64], z = 0.f;
  long long t0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < 16; ++it) {
    if (mode == 0) {  // 3 dependent v_add_f32_dpp, s_nop 1 in front of each
      REP16(asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));)
    }
    if (mode == 1) {  // the same with plain dependent v_add_f32 (no DPP, no nops): the floor of a 3-instruction chain
      REP16(asm volatile("v_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0\n\tv_add_f32 %0, %0, %0" : "+v"(x));)
    }
    if (mode == 2) {  // DPP stages with two independent VALU instructions as the wait states instead of s_nop 1
      REP16(asm volatile("v_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_add_f32 %1, %1, %1\n\tv_add_f32 %2, %2, %2\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "+v"(x), "+v"(y), "+v"(z));)
    }
    if (mode == 3) {  // s_nop 0 twice instead of s_nop 1
      REP16(asm volatile("s_nop 0\n\ts_nop 0\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 0\n\ts_nop 0\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 0\n\ts_nop 0\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x));)
    }
    if (mode == 4) {  // one ds_swizzle butterfly stage + add, dependent (LDS crossbar)
      REP16(asm volatile("ds_swizzle_b32 %1, %0 offset:swizzle(SWAP,1)\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %1\n\t"
                         "ds_swizzle_b32 %1, %0 offset:swizzle(SWAP,2)\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %1\n\t"
                         "ds_swizzle_b32 %1, %0 offset:swizzle(SWAP,4)\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %1" : "+v"(x), "+v"(y));)
    }
    if (mode == 5) {  // plain chain of 3 with s_nop 1 in front of each (what the nops alone cost)
      REP16(asm volatile("s_nop 1\n\tv_add_f32 %0, %0, %0\n\ts_nop 1\n\tv_add_f32 %0, %0, %0\n\ts_nop 1\n\tv_add_f32 %0, %0, %0" : "+v"(x));)
    }
    if (mode == 6) {  // v_mul + 3 DPP stages + v_fma: one level of pass 3
      REP16(asm volatile("v_mul_f32 %0, %0, %1\n\t"
                         "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                         "v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));)
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x + y + z;
  if (threadIdx.x == 0) t[mode] = t1 - t0;
}
int main() {
  float* d; long long* t; hipMalloc(&d, 64 * 8 * 4); hipMalloc(&t, 8 * 8); hipMemset(d, 0, 64 * 8 * 4);
  const char* names[] = {"3 x (s_nop 1 + v_add_f32_dpp), dependent", "3 x v_add_f32, dependent", "3 x (2 independent VALU + v_add_f32_dpp)",
                         "3 x (2 s_nop 0 + v_add_f32_dpp)", "3 x (ds_swizzle + wait + v_add_f32)", "3 x (s_nop 1 + v_add_f32)",
                         "v_mul + 3 x (s_nop 1 + dpp add) + v_fma"};
  for (int m = 0; m < 7; ++m) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t, m); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t, m); }
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
  for (int m = 0; m < 7; ++m) printf("%-48s %7.2f ticks per group (256 groups, one wave)\n", names[m], (double)h[m] / 256.0);
  return 0;
}
