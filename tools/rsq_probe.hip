// Probe: accuracy of v_rsq_f64 and of candidate Newton sequences (diagnostic, not shipped).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  const double d = x[i];
  double y = __builtin_amdgcn_rsq(d);
  o[i] = y;
  // two Newton steps: y <- y + y*(0.5 - 0.5*d*y*y)   as  e = fma(-(d*y), y, 1); y = fma(0.5*y, e, y)
  double t = d * y, hy = 0.5 * y;
  double e = fma(-t, y, 1.0);
  y = fma(hy, e, y);
  o[n + i] = y;
  t = d * y; hy = 0.5 * y;
  e = fma(-t, y, 1.0);
  y = fma(hy, e, y);
  o[2 * n + i] = y;
  o[3 * n + i] = d * y;   // sqrt candidate
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(4 * n);
  srand(7);
  for (int i = 0; i < n; ++i) x[i] = exp((rand() / (double)RAND_MAX) * 40.0 - 20.0);
  double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  for (int i = 0; i < n; ++i) {
    long double r = 1.0L / sqrtl((long double)x[i]), s = sqrtl((long double)x[i]);
    e0 = fmax(e0, (double)fabsl((o[i] - r) / r));
    e1 = fmax(e1, (double)fabsl((o[n + i] - r) / r));
    e2 = fmax(e2, (double)fabsl((o[2 * n + i] - r) / r));
    e3 = fmax(e3, (double)fabsl((o[3 * n + i] - s) / s));
  }
  printf("v_rsq_f64 max rel err %.3e (2^%.1f); after 1 NR %.3e ; after 2 NR %.3e (%.2f ulp); sqrt=d*y %.3e (%.2f ulp)\n", e0, log2(e0), e1, e2,
         e2 / 1.11e-16, e3, e3 / 1.11e-16);
  return 0;
}
