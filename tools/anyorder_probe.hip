// Does hipExtAnyOrderLaunch let a kernel start while its predecessor IN THE SAME STREAM is still running (gfx950, ROCm 7.2)?
// Kernel A spins ~100 us and logs its end; kernel B logs its start.  hipcc --offload-arch=gfx950 -O2 -o tools/anyorder_probe tools/anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long* out, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  out[0] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  out[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void stamp(unsigned long long* out) { out[2] = __builtin_amdgcn_s_memrealtime(); }
int main() {
  unsigned long long* d; hipMalloc(&d, 64); unsigned long long h[3];
  hipStream_t s; hipStreamCreate(&s);
  for (int flag = 0; flag < 2; ++flag) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(d, 0, 64);
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, 0, d, 10000ull);   // 100 us at 100 MHz
      hipExtLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, flag, d);
      hipStreamSynchronize(s);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      printf("flags=%d: A ran %.1f us; B started %.1f us after A STARTED, %.1f us after A ENDED\n", flag, (h[1] - h[0]) / 100.0,
             ((double)h[2] - (double)h[0]) / 100.0, ((double)h[2] - (double)h[1]) / 100.0);
    }
  }
  return 0;
}
