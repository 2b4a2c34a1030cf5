// [round 4] accuracy of v_rcp_f64 / v_rsq_f64 seeds and Newton steps against IEEE division / sqrt (fp64 kernels: the
// compiler's correctly rounded 1.0 / x is ~14 instructions, sqrt ~20).  hipcc --offload-arch=gfx950 -O3 -o rcp64 rcp64.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
__global__ void k(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  out[i] = r;
  r = fma(fma(-v, r, 1.0), r, r);
  out[n + i] = r;
  r = fma(fma(-v, r, 1.0), r, r);
  out[2 * n + i] = r;
  double q = __builtin_amdgcn_rsq(v);
  out[3 * n + i] = q;
  q = q * fma(-0.5 * v * q, q, 1.5);
  out[4 * n + i] = q;
  q = q * fma(-0.5 * v * q, q, 1.5);
  out[5 * n + i] = q;
  q = q * fma(-0.5 * v * q, q, 1.5);
  out[6 * n + i] = q;
}
int main() {
  const int n = 1 << 20;
  double* hx = (double*)malloc(n * 8);
  double* ho = (double*)malloc(7 * n * 8);
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = ldexp(1.0 + (double)rand() / RAND_MAX, (rand() % 120) - 60);
  double *dx, *dout;
  hipMalloc(&dx, n * 8), hipMalloc(&dout, 7 * n * 8);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(ho, dout, 7 * n * 8, hipMemcpyDeviceToHost);
  const char* names[7] = {"rcp seed", "rcp + 1 Newton", "rcp + 2 Newton", "rsq seed", "rsq + 1 Newton", "rsq + 2 Newton", "rsq + 3 Newton"};
  for (int j = 0; j < 7; ++j) {
    double worst = 0;
    for (int i = 0; i < n; ++i) {
      const double ref = j < 3 ? 1.0 / hx[i] : 1.0 / sqrt(hx[i]);
      const double e = fabs(ho[j * n + i] - ref) / fabs(ref);
      if (e > worst) worst = e;
    }
    printf("%-16s worst relative error %.3e (%.1f ulp)\n", names[j], worst, worst / 1.11e-16);
  }
  return 0;
}
