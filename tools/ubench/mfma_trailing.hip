// Developer micro-benchmark: the accumulation work of the blocked Cholesky of the
// rigid-contact solver (48 x 48, sixteen points, four environments per wave) done
//   (A) as in jxs_rigid.inc::rigid_cholesky: left-looking, every lane (= one point) accumulates the 3x3 products
//       of its three rows with the rows of the current block column over all previous block columns
//       (18 LDS reads + 27 FMAs per previous block column), and
//   (B) right-looking with MFMA: after block column j the trailing matrix takes A22 -= L21 L21^T as
//       v_mfma_f32_16x16x4_f32 tiles (K = 3 padded to 4) held in registers, six lower-triangular 16x16 tiles
//       per environment, operands = the panel read from LDS once per tile row.
// ONE resident wave, random factor entries, both variants compute the same Schur updates of the last block
// (cross-checked); cycles by s_memtime.  The dependent work of a real factorisation (diagonal 3x3 blocks,
// broadcasts, extraction of the next panel from the tiles) is NOT included in either number.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int NB = 16, NX = 48, NT = NX * (NX + 1) / 2, ENVS = 4;
__device__ __forceinline__ int tri(int i) { return (i * (i + 1)) / 2; }

__global__ void k_valu(const float* Lin, float* out, long long* t) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 15, env = threadIdx.x >> 4;
  float* L = lds + env * NT;
  for (int i = threadIdx.x & 15; i < NT; i += 16) L[i] = Lin[env * NT + i];
  __syncthreads();
  float total[3][3] = {};
  const long long t0 = __builtin_readcyclecounter();
  for (int jc = 1; jc < NB; ++jc) {
    float acc[3][3] = {};
    for (int kb = 0; kb < jc; ++kb) {
      float pr[3][3], ow[3][3];
      for (int u = 0; u < 3; ++u)
        for (int c = 0; c < 3; ++c) {
          pr[c][u] = L[tri(3 * jc + c) + 3 * kb + u];
          ow[c][u] = (lane >= jc) ? L[tri(3 * lane + c) + 3 * kb + u] : 0.0f;
        }
      for (int u = 0; u < 3; ++u)
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) acc[r][c] += ow[r][u] * pr[c][u];
    }
    if (jc == NB - 1)
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) total[r][c] = acc[r][c];
    else  // keep the work alive
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) total[r][c] += 1e-30f * acc[r][c];
  }
  const long long t1 = __builtin_readcyclecounter();
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out[(threadIdx.x * 3 + r) * 3 + c] = total[r][c];
  if (threadIdx.x == 0) t[0] = t1 - t0;
}

__global__ void k_mfma(const float* Lin, float* out, long long* t) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < ENVS * NT; i += 64) lds[i] = Lin[i];
  __syncthreads();
  const int l = threadIdx.x, i16 = l & 15, k4 = l >> 4;
  // C tiles (ti >= tj) of every environment, accumulators start at zero: C = sum_j L21_j L21_j^T
  v4f C[ENVS][6];
  for (int e = 0; e < ENVS; ++e)
    for (int q = 0; q < 6; ++q) C[e][q] = v4f{0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int jc = 0; jc < NB - 1; ++jc) {
    for (int e = 0; e < ENVS; ++e) {
      const float* L = lds + e * NT;
      float op[3];
      // operand of tile row ti: lane (i16, k4) holds L[16 ti + i16][3 jc + k4] (zero for k4 = 3 and rows of done blocks)
      for (int ti = 0; ti < 3; ++ti) {
        const int row = 16 * ti + i16;
        op[ti] = (k4 < 3 && row >= 3 * (jc + 1)) ? L[tri(row) + 3 * jc + k4] : 0.0f;
      }
      int q = 0;
      for (int ti = 0; ti < 3; ++ti)
        for (int tj = 0; tj <= ti; ++tj, ++q)
          if (16 * (ti + 1) > 3 * (jc + 1))  // tiles above the current block column are finished
            C[e][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(op[ti], op[tj], C[e][q], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  // C tile (2,2) holds rows 32..47 x cols 32..47: lane (col = i16, rows 4 k4 + i) -> out[env][row][col]
  for (int e = 0; e < ENVS; ++e)
    for (int i = 0; i < 4; ++i) out[(e * 16 + 4 * k4 + i) * 16 + i16] = C[e][5][i];
  if (threadIdx.x == 0) t[1] = t1 - t0;
}

int main() {
  const int n = ENVS * NT;
  float* h = (float*)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  float *dL, *o1, *o2;
  long long* t;
  hipMalloc(&dL, n * 4), hipMalloc(&o1, 64 * 9 * 4), hipMalloc(&o2, ENVS * 256 * 4), hipMalloc(&t, 16);
  hipMemcpy(dL, h, n * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_valu, dim3(1), dim3(64), n * 4, 0, dL, o1, t);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), n * 4, 0, dL, o2, t);
  }
  hipDeviceSynchronize();
  long long ht[2];
  float a[64 * 9], b[ENVS * 256];
  hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost), hipMemcpy(a, o1, sizeof(a), hipMemcpyDeviceToHost), hipMemcpy(b, o2, sizeof(b), hipMemcpyDeviceToHost);
  // cross-check: the accumulated products of the LAST block column (jc = 15) for the owner lane 15 of every
  // environment = rows 45..47 x cols 45..47 of the MFMA result
  double err = 0;
  for (int e = 0; e < ENVS; ++e)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        const double va = a[((e * 16 + 15) * 3 + r) * 3 + c], vb = b[(e * 16 + 13 + r) * 16 + 13 + c];
        err = fmax(err, fabs(va - vb));
      }
  printf("accumulation work of one 48x48 factorisation, 4 environments, one wave:\n");
  printf("  (A) VALU, left-looking (rigid_cholesky):  %8lld cycles\n", ht[0]);
  printf("  (B) MFMA 16x16x4 f32, right-looking:      %8lld cycles\n", ht[1]);
  printf("  cross-check of the last diagonal block: max |A - B| = %.2e\n", err);
  return 0;
}
