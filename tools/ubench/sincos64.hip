// [round 4] accuracy of a short fp64 sincos (3-term Cody-Waite reduction, degree-13 / degree-14 kernels) against the
// host's libm on 2^20 random joint angles -- the fp64 kernels call sincos once per joint and step, and the device
// library's version is ~150 instructions with a Payne-Hanek branch.  hipcc --offload-arch=gfx950 -O3 -o sincos64 sincos64.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ void vsincos(double x, double& s, double& c) {
  const double kf = __builtin_rint(x * 6.36619772367581382433e-01);
  double r = __builtin_fma(kf, -1.57079632673412561417e+00, x);
  r = __builtin_fma(kf, -6.07710050630396597660e-11, r);
  r = __builtin_fma(kf, -2.02226624879595063154e-21, r);
  const double r2 = r * r;
  double ps = __builtin_fma(r2, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(r2, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(r2, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(r2, ps, 8.33333333332248946124e-03);
  ps = __builtin_fma(r2, ps, -1.66666666666666324348e-01);
  const double sr = __builtin_fma(r * r2, ps, r);
  double pc = __builtin_fma(r2, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(r2, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(r2, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(r2, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(r2, pc, 4.16666666666666019037e-02);
  const double cr = __builtin_fma(r2 * r2, pc, __builtin_fma(r2, -0.5, 1.0));
  const int k = (int)kf;
  const bool swap = (k & 1) != 0;
  const double s0 = swap ? cr : sr, c0 = swap ? sr : cr;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}
__global__ void kern(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s, c;
  vsincos(x[i], s, c);
  out[i] = s, out[n + i] = c;
}
int main() {
  const int n = 1 << 20;
  double *hx = (double*)malloc(n * 8), *ho = (double*)malloc(2 * n * 8);
  srand(2);
  const double ranges[3] = {3.2, 100.0, 1.0e4};
  for (int rg = 0; rg < 3; ++rg) {
    for (int i = 0; i < n; ++i) hx[i] = ranges[rg] * (2.0 * rand() / RAND_MAX - 1.0);
    double *dx, *dout;
    hipMalloc(&dx, n * 8), hipMalloc(&dout, 2 * n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(ho, dout, 2 * n * 8, hipMemcpyDeviceToHost);
    double ws = 0, wc = 0;
    for (int i = 0; i < n; ++i) {
      ws = fmax(ws, fabs(ho[i] - sin(hx[i])));
      wc = fmax(wc, fabs(ho[n + i] - cos(hx[i])));
    }
    printf("|x| < %-7g worst absolute error sin %.3e cos %.3e (in units of 2^-53: %.2f / %.2f)\n", ranges[rg], ws, wc, ws / 1.11e-16, wc / 1.11e-16);
    hipFree(dx), hipFree(dout);
  }
  return 0;
}
