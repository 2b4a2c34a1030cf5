// Developer micro-benchmark: does a DPP operand see a stale EXEC mask behind a SCALAR write of EXEC?  [round 4]
//
// Background: in round 3 the RungeKutta4 + RigidContacts kernel returned non-finite
// environments on some calls; the cure that worked was `s_nop 4` in front of the DPP blocks that sit in / behind an
// exec-masked region, and the diagnosis was "a DPP instruction a few cycles behind a scalar write of EXEC decides
// which source lanes are switched off by the OLD mask".  The ISA manual and LLVM's hazard recogniser only know
//   (a) VALU writes EXEC  -> DPP op           : 5 wait states
//   (b) VALU writes VGPR  -> DPP reads it     : 2 wait states
// This program measures every candidate with k = 0..6 wait states between producer and DPP consumer and counts the
// lanes whose result differs from the architectural one (new mask, new value).  (a) and (b) are the positive controls:
// if they show errors for small k and the scalar cases show none, the scalar case does not exist.
//
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/exec_dpp.hip -o tools/ubench/exec_dpp && tools/ubench/exec_dpp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define NOPS_0 ""
#define NOPS_1 "s_nop 0\n\t"
#define NOPS_2 "s_nop 1\n\t"
#define NOPS_3 "s_nop 2\n\t"
#define NOPS_4 "s_nop 3\n\t"
#define NOPS_5 "s_nop 4\n\t"
#define NOPS_6 "s_nop 5\n\t"

// four independent VALU instructions in front of the producer: the vector pipeline is busy when EXEC changes
#define FILL "v_fma_f32 %[f0], %[f0], %[f0], %[f1]\n\tv_fma_f32 %[f1], %[f1], %[f1], %[f2]\n\tv_fma_f32 %[f2], %[f2], %[f2], %[f3]\n\tv_fma_f32 %[f3], %[f3], %[f3], %[f0]\n\t"
#define DPP_SHL "v_mov_b32_dpp %[out], %[v] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define DPP_ROW "v_mov_b32_dpp %[out], %[v] row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define DPP_QUAD "v_mov_b32_dpp %[out], %[v] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define SETTLE "s_nop 7\n\t"

struct Res {
  unsigned long long bad;   // lanes x iterations with a wrong result
  unsigned first_lane, first_got, first_want, pad;
};

// one atomic per WAVE (contended per-lane atomics on one address made the first version of this program crawl)
__device__ __forceinline__ void report(Res* r, unsigned nbad, unsigned lane, unsigned got, unsigned want, bool isbad) {
  unsigned tot = nbad;
  for (int off = 32; off; off >>= 1) tot += __shfl_xor(tot, off);
  const unsigned long long bad_lanes = __ballot(isbad);
  if (tot && lane == 0) atomicAdd(&r->bad, (unsigned long long)tot);
  if (bad_lanes && lane == (unsigned)__ffsll((long long)bad_lanes) - 1u && atomicCAS(&r->pad, 0u, 1u) == 0u)
    r->first_lane = lane, r->first_got = got, r->first_want = want;
}

// Every test: `want` is computed on the host-visible rule "source lane switched off or beyond the wave -> 0 (bound_ctrl),
// destination lane switched off -> keeps the sentinel".
#define PROLOGUE                                                                                     \
  const unsigned lane = threadIdx.x & 63;                                                           \
  const unsigned long long M = 0x5555555555555555ull; /* even lanes on */                           \
  float f0 = lane, f1 = 1.5f, f2 = 0.25f, f3 = 3.0f;                                                  \
  unsigned nbad = 0, fgot = 0, fwant = 0;                                                            \
  bool any = false;                                                                                  \
  for (int it = 0; it < iters; ++it) {                                                              \
    const unsigned v = ((unsigned)it << 8 | lane) + 1u;                                             \
    const unsigned sentinel = 0xDEAD0000u | lane;                                                   \
    unsigned out = sentinel;                                                                         \
    unsigned long long save;
#define EPILOGUE                                                                                     \
    if (out != want) {                                                                               \
      ++nbad;                                                                                        \
      if (!any) any = true, fgot = out, fwant = want;                                                \
    }                                                                                                \
  }                                                                                                  \
  report(res, nbad, lane, fgot, fwant, any);                                                         \
  if (f0 + f1 + f2 + f3 == 12345.f) res->pad = 7;

#define FOPS [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3)

#define KERNELS(k)                                                                                                      \
  /* T1: lanes switch OFF.  full -> s_mov_b64 exec, M -> k -> DPP.  Even lanes read an odd (off) lane: 0. */            \
  __global__ void t1_##k(Res* res, int iters) {                                                                         \
    PROLOGUE                                                                                                            \
    asm volatile("s_mov_b64 %[save], exec\n\t" FILL "s_mov_b64 exec, %[m]\n\t" NOPS_##k DPP_SHL SETTLE                  \
                 "s_mov_b64 exec, %[save]\n\t" SETTLE                                                                   \
                 : [out] "+v"(out), [save] "=&s"(save), FOPS : [v] "v"(v), [m] "s"(M));                               \
    const unsigned want = (lane & 1) ? sentinel : 0u;                                                                   \
    EPILOGUE                                                                                                            \
  }                                                                                                                     \
  /* T2: lanes switch ON.  exec = M (settled) -> masked VALU -> s_or_b64 exec, exec, save -> k -> DPP.  All read v. */  \
  __global__ void t2_##k(Res* res, int iters) {                                                                         \
    PROLOGUE                                                                                                            \
    asm volatile("s_mov_b64 %[save], exec\n\ts_mov_b64 exec, %[m]\n\t" SETTLE FILL "s_or_b64 exec, exec, %[save]\n\t"   \
                 NOPS_##k DPP_SHL SETTLE                                                                                \
                 : [out] "+v"(out), [save] "=&s"(save), FOPS : [v] "v"(v), [m] "s"(M));                               \
    const unsigned want = lane == 63 ? 0u : v + 1u;                                                                     \
    EPILOGUE                                                                                                            \
  }                                                                                                                     \
  /* T3: the round-3 sequence.  s_or_b64 exec (end of a region) ; s_and_saveexec_b64 (next region) -> k -> DPP */       \
  __global__ void t3_##k(Res* res, int iters) {                                                                         \
    PROLOGUE                                                                                                            \
    unsigned long long sv2;                                                                                             \
    asm volatile("s_mov_b64 %[save], exec\n\ts_mov_b64 exec, %[m2]\n\t" SETTLE FILL "s_or_b64 exec, exec, %[save]\n\t"  \
                 "s_and_saveexec_b64 %[sv2], %[m]\n\t" NOPS_##k DPP_SHL SETTLE "s_or_b64 exec, exec, %[sv2]\n\t" SETTLE  \
                 : [out] "+v"(out), [save] "=&s"(save), [sv2] "=&s"(sv2), FOPS                                         \
                 : [v] "v"(v), [m] "s"(M), [m2] "s"(~M));                                                              \
    const unsigned want = (lane & 1) ? sentinel : 0u;                                                                   \
    EPILOGUE                                                                                                            \
  }                                                                                                                     \
  /* T4 (control a): VALU writes EXEC (v_cmpx) -> k -> DPP.  Documented: 5 wait states. */                              \
  __global__ void t4_##k(Res* res, int iters) {                                                                         \
    PROLOGUE                                                                                                            \
    const unsigned sel = (lane & 1) ? 0u : 1u;                                                                          \
    asm volatile("s_mov_b64 %[save], exec\n\t" FILL "v_cmpx_ne_u32_e32 vcc, 0, %[sel]\n\t" NOPS_##k DPP_SHL SETTLE      \
                 "s_mov_b64 exec, %[save]\n\t" SETTLE                                                                   \
                 : [out] "+v"(out), [save] "=&s"(save), FOPS : [v] "v"(v), [sel] "v"(sel) : "vcc");                    \
    const unsigned want = (lane & 1) ? sentinel : 0u;                                                                   \
    EPILOGUE                                                                                                            \
  }                                                                                                                     \
  /* T5 (control b): VALU writes the DPP source -> k -> DPP reads it.  Documented: 2 wait states. */                    \
  __global__ void t5_##k(Res* res, int iters) {                                                                         \
    PROLOGUE                                                                                                            \
    unsigned x = 0x11110000u | lane;                                                                                    \
    asm volatile(FILL "v_mov_b32 %[x], %[v]\n\t" NOPS_##k "v_mov_b32_dpp %[out], %[x] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" SETTLE \
                 : [out] "+v"(out), [x] "+v"(x), FOPS : [v] "v"(v));                                                   \
    save = 0;                                                                                                           \
    const unsigned want = lane == 63 ? 0u : v + 1u;                                                                     \
    EPILOGUE                                                                                                            \
  }                                                                                                                     \
  /* T6: exec-masked LDS write, region ends (s_or_b64 exec), -> k -> DPP */                                              \
  __global__ void t6_##k(Res* res, int iters) {                                                                         \
    __shared__ unsigned lds[256];                                                                                       \
    lds[threadIdx.x & 255] = 0;                                                                                         \
    PROLOGUE                                                                                                            \
    const unsigned addr = (threadIdx.x & 255) * 4u;                                                                     \
    asm volatile("s_mov_b64 %[save], exec\n\ts_mov_b64 exec, %[m]\n\t" SETTLE "ds_write_b32 %[addr], %[v]\n\t"          \
                 "s_or_b64 exec, exec, %[save]\n\t" NOPS_##k DPP_SHL SETTLE "s_waitcnt lgkmcnt(0)\n\t"                  \
                 : [out] "+v"(out), [save] "=&s"(save), FOPS : [v] "v"(v), [m] "s"(M), [addr] "v"(addr) : "memory");   \
    const unsigned want = lane == 63 ? 0u : v + 1u;                                                                     \
    EPILOGUE                                                                                                            \
    if (lds[(threadIdx.x + 1) & 255] == 0x12345u) res->pad = 9;                                                                                                           \
  }                                                                                                                     \
  /* T7: the DPP source is written by a VALU inside the masked region; s_or_b64 exec (1 state) -> k -> DPP.           \
     Even lanes hold the new value, odd lanes the old one; needs 2 states in total if the scalar counts as one. */      \
  __global__ void t7_##k(Res* res, int iters) {                                                                         \
    PROLOGUE                                                                                                            \
    unsigned x = 0x22220000u | lane;                                                                                    \
    asm volatile("s_mov_b64 %[save], exec\n\ts_mov_b64 exec, %[m]\n\t" SETTLE FILL "v_mov_b32 %[x], %[v]\n\t"           \
                 "s_or_b64 exec, exec, %[save]\n\t" NOPS_##k                                                            \
                 "v_mov_b32_dpp %[out], %[x] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" SETTLE             \
                 : [out] "+v"(out), [save] "=&s"(save), [x] "+v"(x), FOPS : [v] "v"(v), [m] "s"(M));                  \
    const unsigned want = lane == 63 ? 0u : ((lane & 1) ? v + 1u : (0x22220000u | (lane + 1)));                         \
    EPILOGUE                                                                                                            \
  }                                                                                                                     \
  /* T10 (control c): VALU writes the NON-shuffled operand (src1) of a DPP instruction -> k -> DPP.  LLVM pads 2. */      \
  __global__ void t10_##k(Res* res, int iters) {                                                                        \
    PROLOGUE                                                                                                            \
    unsigned x = 0x33330000u | lane;                                                                                    \
    asm volatile(FILL "v_mov_b32 %[x], %[v]\n\t" NOPS_##k "v_add_u32_dpp %[out], %[v], %[x] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" SETTLE \
                 : [out] "+v"(out), [x] "+v"(x), FOPS : [v] "v"(v));                                                   \
    save = 0;                                                                                                           \
    const unsigned want = lane == 63 ? v : v + 1u + v;                                                                  \
    EPILOGUE                                                                                                            \
  }                                                                                                                     \
  /* T11 (control d): VALU writes the ACCUMULATOR of v_fmac_f32_dpp (a plain operand too) -> k -> DPP */                \
  __global__ void t11_##k(Res* res, int iters) {                                                                        \
    PROLOGUE                                                                                                            \
    float acc = 0.5f;                                                                                                   \
    const float vf = (float)(lane + 1), one = 1.0f, base = (float)((it & 255) + 3);                                      \
    asm volatile(FILL "v_mov_b32 %[acc], %[base]\n\t" NOPS_##k "v_fmac_f32_dpp %[acc], %[vf], %[one] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" SETTLE \
                 : [acc] "+v"(acc), FOPS : [vf] "v"(vf), [one] "v"(one), [base] "v"(base));                            \
    save = 0;                                                                                                           \
    out = __float_as_uint(acc);                                                                                         \
    const unsigned want = __float_as_uint(base + (lane == 63 ? 0.0f : (float)(lane + 2)));                              \
    EPILOGUE                                                                                                            \
  }                                                                                                                     \
  /* T8: T1 with row_shl:1 (16-lane rows) */                                                                            \
  __global__ void t8_##k(Res* res, int iters) {                                                                         \
    PROLOGUE                                                                                                            \
    asm volatile("s_mov_b64 %[save], exec\n\t" FILL "s_and_b64 exec, exec, %[m]\n\t" NOPS_##k DPP_ROW SETTLE            \
                 "s_mov_b64 exec, %[save]\n\t" SETTLE                                                                   \
                 : [out] "+v"(out), [save] "=&s"(save), FOPS : [v] "v"(v), [m] "s"(M));                               \
    const unsigned want = (lane & 1) ? sentinel : 0u;                                                                   \
    EPILOGUE                                                                                                            \
  }                                                                                                                     \
  /* T9: T2 with quad_perm (lanes switch on; the quad partner of an even lane is the odd lane that was off) */          \
  __global__ void t9_##k(Res* res, int iters) {                                                                         \
    PROLOGUE                                                                                                            \
    asm volatile("s_mov_b64 %[save], exec\n\ts_mov_b64 exec, %[m]\n\t" SETTLE FILL "s_mov_b64 exec, %[save]\n\t"        \
                 NOPS_##k DPP_QUAD SETTLE                                                                               \
                 : [out] "+v"(out), [save] "=&s"(save), FOPS : [v] "v"(v), [m] "s"(M));                               \
    const unsigned want = (((unsigned)it << 8 | (lane ^ 1)) + 1u);                                                      \
    EPILOGUE                                                                                                            \
  }

KERNELS(0)
KERNELS(1)
KERNELS(2)
KERNELS(3)
KERNELS(4)
KERNELS(5)
KERNELS(6)

typedef void (*Kern)(Res*, int);
#define ROW(t) {t##_0, t##_1, t##_2, t##_3, t##_4, t##_5, t##_6}
static Kern table[11][7] = {ROW(t1), ROW(t2), ROW(t3), ROW(t4), ROW(t5), ROW(t6), ROW(t7), ROW(t8), ROW(t9), ROW(t10), ROW(t11)};
static const char* names[11] = {
    "T1 s_mov_b64 exec (lanes off)          -> DPP wave_shl", "T2 s_or_b64 exec (lanes on)            -> DPP wave_shl",
    "T3 s_or_b64 ; s_and_saveexec_b64       -> DPP wave_shl", "T4 CONTROL v_cmpx writes EXEC (doc: 5) -> DPP wave_shl",
    "T5 CONTROL v_mov writes source (doc: 2)-> DPP wave_shl", "T6 masked ds_write ; s_or_b64 exec     -> DPP wave_shl",
    "T7 masked v_mov of source ; s_or exec  -> DPP wave_shl", "T8 s_and_b64 exec (lanes off)          -> DPP row_shl ",
    "T9 s_mov_b64 exec (lanes on)           -> DPP quad_perm",
    "T10 CONTROL v_mov writes src1 (LLVM: 2)-> DPP wave_shl",
    "T11 CONTROL v_mov writes fmac's acc    -> DPP wave_shl"};

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4096;
  setvbuf(stdout, nullptr, _IONBF, 0);
  Res* d;
  hipMalloc(&d, sizeof(Res));
  // waves per SIMD ~ blocks / 1024 for one-wave blocks on 256 CUs x 4 SIMDs; 256-thread blocks put 4 waves on a CU at once
  struct Geo { int blocks, threads; const char* what; } geos[] = {{1024, 64, "1024 x 1 wave"}, {4096, 64, "4096 x 1 wave"}, {2048, 256, "2048 x 4 waves"}};
  for (const Geo& g : geos) {
    printf("== %s, %d iterations per lane (%.2e DPP results per cell) ==\n", g.what, iters, (double)g.blocks * g.threads * iters);
    printf("%-56s %10s %10s %10s %10s %10s %10s %10s\n", "wrong lanes at k wait states:", "k=0", "k=1", "k=2", "k=3", "k=4", "k=5", "k=6");
    for (int t = 0; t < 11; ++t) {
      printf("%-56s", names[t]);
      Res firsts[7];
      for (int k = 0; k < 7; ++k) {
        hipMemset(d, 0, sizeof(Res));
        hipLaunchKernelGGL(table[t][k], dim3(g.blocks), dim3(g.threads), 0, 0, d, iters);
        if (hipDeviceSynchronize() != hipSuccess) { printf(" launch failed\n"); return 1; }
        hipMemcpy(&firsts[k], d, sizeof(Res), hipMemcpyDeviceToHost);
        printf(" %10llu", firsts[k].bad);
      }
      printf("\n");
      for (int k = 0; k < 7; ++k)
        if (firsts[k].bad) { printf("      first at k=%d: lane %u got 0x%08x want 0x%08x\n", k, firsts[k].first_lane, firsts[k].first_got, firsts[k].first_want); break; }
    }
  }
  return 0;
}
