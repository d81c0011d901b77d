// peaks.hip -- measured ceilings on this MI355X for the two rooflines the path is priced
// against: fp64 MFMA (v_mfma_f64_16x16x4_f64) issue rate, fp64 VALU FMA rate, and the
// HBM write/copy rate of a 16-B-per-lane streaming kernel.   Build: see __graft_entry__.build().
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

// (256, 2): accumulators in VGPRs; with the default register budget hipcc uses AGPR accumulators and the
// f64 MFMA then issues every ~100 cycles instead of 64 (mfma_loop_agpr below keeps that form for the record)
template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_loop(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop_agpr(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}

__global__ __launch_bounds__(256) void fma_loop(double* out, int iters, double a0) {
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x * 1e-3;
  const double m = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = fma(x[i], m, c);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 12345.678) out[0] = s;
}

// co-issue probe: half the waves of a 512-thread block run the MFMA loop, the other half the VALU FMA loop
template <int MFMA_WAVES>
__global__ __launch_bounds__(512) void coissue_loop(double* out, int iters, double a0) {
  const int wave = threadIdx.x >> 6;
  double s = 0.0;
  if (wave < MFMA_WAVES) {
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-3, b = 2.0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = a0 + i + threadIdx.x * 1e-3;
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters * 14; ++it) {   // ~ the same wall time as the MFMA half
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = fma(x[i], m, c);
    }
    for (int i = 0; i < 8; ++i) s += x[i];
  }
  if (s == 12345.678) out[0] = s;
}

__global__ __launch_bounds__(256) void fill_kernel(v2d* dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v2d v = {1.0, 2.0};
  for (; i < n; i += stride) dst[i] = v;
}

__global__ __launch_bounds__(256) void copy_kernel(v2d* dst, const v2d* src, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}

// streaming read (16 B per lane, 4 loads in flight per thread per iteration), result kept live through a sum
__global__ __launch_bounds__(256) void read_kernel(const v2d* src, size_t n, double* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const v2d a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
  }
  for (; i < n; i += stride) acc += src[i].x + src[i].y;
  if (acc == 12345.678) out[0] = acc;
}

// write-only variants: 4 stores in flight per thread; non-temporal stores
__global__ __launch_bounds__(256) void fill4_kernel(v2d* dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const v2d v = {1.0, 2.0};
  for (; i + 3 * stride < n; i += 4 * stride) { dst[i] = v; dst[i + stride] = v; dst[i + 2 * stride] = v; dst[i + 3 * stride] = v; }
  for (; i < n; i += stride) dst[i] = v;
}
__global__ __launch_bounds__(256) void fill_nt_kernel(v2d* dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const v2d v = {1.0, 2.0};
  for (; i < n; i += stride) __builtin_nontemporal_store(v, &dst[i]);
}
// contiguous 1 KiB per wave-instruction, each workgroup owning one contiguous slab (the gram kernel's store shape: row segments)
__global__ __launch_bounds__(256) void fill_slab_kernel(v2d* dst, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
  const v2d v = {1.0, 2.0};
  for (size_t i = b0 + threadIdx.x; i < b1; i += 256) dst[i] = v;
}

// in-kernel clock under an MFMA-dense loop: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime)
__global__ __launch_bounds__(256) void clock_probe(double* out, unsigned long long* clk, int iters) {
  v4d acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-3, b = 2.0 - threadIdx.x * 1e-3;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 12345.678) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <typename F>
static float time_ms(F&& f, int reps) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) f();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s %s CUs=%d clock=%d MHz\n", prop.name, prop.gcnArchName, cus, prop.clockRate / 1000);
  double* out; CHECK(hipMalloc(&out, 64));
  const int iters = 4096;
  for (int wpc = 4; wpc <= 32; wpc *= 2) {   // waves per CU: 4 = one per SIMD, 8 = two per SIMD, ...
    dim3 grid(cus * wpc / 4), block(256);
    float ms8 = time_ms([&] { hipLaunchKernelGGL(mfma_loop<8>, grid, block, 0, 0, out, iters, 1.0, 2.0); }, 5);
    printf("mfma_f64_16x16x4 waves/CU=%d: 8 acc %.2f TFLOP/s\n", wpc, 2048.0 * iters * 8 * (double)grid.x * 4 / ms8 * 1e-9);
    float msa = time_ms([&] { hipLaunchKernelGGL(mfma_loop_agpr<4>, grid, block, 0, 0, out, iters, 1.0, 2.0); }, 5);
    printf("mfma_f64_16x16x4 waves/CU=%d: 4 acc in AGPRs (default register budget) %.2f TFLOP/s\n", wpc, 2048.0 * iters * 4 * (double)grid.x * 4 / msa * 1e-9);
    float ms1 = time_ms([&] { hipLaunchKernelGGL(mfma_loop<1>, grid, block, 0, 0, out, iters, 1.0, 2.0); }, 5);
    float ms4 = time_ms([&] { hipLaunchKernelGGL(mfma_loop<4>, grid, block, 0, 0, out, iters, 1.0, 2.0); }, 5);
    double fl1 = 2048.0 * iters * 1 * (double)grid.x * 4, fl4 = 2048.0 * iters * 4 * (double)grid.x * 4;
    printf("mfma_f64_16x16x4 waves/CU=%d: 1 acc %.2f TFLOP/s (%.1f cyc/MFMA @2.4GHz), 4 acc %.2f TFLOP/s (%.1f cyc/MFMA/SIMD)\n", wpc,
           fl1 / ms1 * 1e-9, ms1 * 1e-3 * 2.4e9 / iters / (wpc / 4), fl4 / ms4 * 1e-9, ms4 * 1e-3 * 2.4e9 / (iters * 4) / (wpc / 4));
  }
  {
    unsigned long long* clk; CHECK(hipMalloc(&clk, 16));
    dim3 grid(cus * 2), block(256);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(clock_probe, grid, block, 0, 0, out, clk, 1 << 16);
    CHECK(hipDeviceSynchronize());
    unsigned long long hc[2]; CHECK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
    printf("in-kernel clock under fp64 MFMA load: %.0f MHz; cycles per MFMA per SIMD (2 waves/SIMD, 4 acc): %.1f\n",
           (double)hc[0] / (double)hc[1] * 100.0, (double)hc[0] / ((double)(1 << 16) * 4 * 2));
  }
  {
    // 4 MFMA waves + 4 VALU waves per block, 2 blocks per CU
    dim3 grid(cus * 2), block(512);
    const int it2 = 2048;
    float ms = time_ms([&] { hipLaunchKernelGGL(coissue_loop<4>, grid, block, 0, 0, out, it2, 1.0); }, 5);
    const double fl_mfma = 2048.0 * it2 * 4 * (double)grid.x * 4, fl_valu = 2.0 * 8 * 64 * (double)it2 * 14 * (double)grid.x * 4;
    printf("co-issue (4 MFMA waves + 4 VALU waves per block): %.2f TFLOP/s total (MFMA part %.2f + VALU part %.2f if both spanned the whole kernel), %.3f ms\n",
           (fl_mfma + fl_valu) / ms * 1e-9, fl_mfma / ms * 1e-9, fl_valu / ms * 1e-9, ms);
    float ms_m = time_ms([&] { hipLaunchKernelGGL(coissue_loop<8>, grid, block, 0, 0, out, it2, 1.0); }, 5);
    float ms_v = time_ms([&] { hipLaunchKernelGGL(coissue_loop<0>, grid, block, 0, 0, out, it2, 1.0); }, 5);
    printf("  same kernel, all 8 waves MFMA: %.3f ms (%.2f TFLOP/s); all 8 waves VALU: %.3f ms (%.2f TFLOP/s)\n", ms_m,
           2.0 * fl_mfma / ms_m * 1e-9, ms_v, 2.0 * fl_valu / ms_v * 1e-9);
  }
  {
    dim3 grid(cus * 8), block(256);   // 8 waves per SIMD worth of blocks
    float ms = time_ms([&] { hipLaunchKernelGGL(fma_loop, grid, block, 0, 0, out, iters, 1.0); }, 5);
    double fl = 2.0 * 8 * iters * (double)grid.x * 256;
    printf("v_fma_f64: %.2f TFLOP/s\n", fl / ms * 1e-9);
  }
  {
    const size_t bytes = (size_t)1 << 31;  // 2 GiB
    v2d *a, *b; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
    const size_t n = bytes / sizeof(v2d);
    dim3 grid(cus * 8), block(256);
    float msf = time_ms([&] { hipLaunchKernelGGL(fill_kernel, grid, block, 0, 0, a, n); }, 5);
    float msc = time_ms([&] { hipLaunchKernelGGL(copy_kernel, grid, block, 0, 0, b, a, n); }, 5);
    printf("HBM fill (write only): %.2f TB/s ; copy (read+write): %.2f TB/s\n", bytes / msf * 1e-9, 2.0 * bytes / msc * 1e-9);
    // The guide's 6.0-6.3 TB/s figures (MI355X_MICROARCH.md:296,332,351) are a streaming READ of a 1.2 GB table and plain
    // stores into 75 / 302 MB tables (the smaller one inside the 256 MB Infinity Cache).  Reads, the store variants and a
    // cache-sized footprint, to see which ceiling a 2 GiB write stream -- the gram build's traffic -- actually has here:
    for (int wpc : {8, 16, 32}) {
      dim3 g2(cus * wpc / 4);
      float msr = time_ms([&] { hipLaunchKernelGGL(read_kernel, g2, block, 0, 0, a, n, out); }, 5);
      float ms4 = time_ms([&] { hipLaunchKernelGGL(fill4_kernel, g2, block, 0, 0, a, n); }, 5);
      float msn = time_ms([&] { hipLaunchKernelGGL(fill_nt_kernel, g2, block, 0, 0, a, n); }, 5);
      float mss = time_ms([&] { hipLaunchKernelGGL(fill_slab_kernel, g2, block, 0, 0, a, n); }, 5);
      printf("2 GiB, %2d waves/CU: read %.2f TB/s ; fill x4 in flight %.2f ; fill non-temporal %.2f ; fill slab-per-workgroup %.2f TB/s\n", wpc,
             bytes / msr * 1e-9, bytes / ms4 * 1e-9, bytes / msn * 1e-9, bytes / mss * 1e-9);
    }
    for (size_t mb : {64, 128, 512}) {
      const size_t nb = (mb << 20) / sizeof(v2d);
      float msr = time_ms([&] { hipLaunchKernelGGL(read_kernel, grid, block, 0, 0, a, nb, out); }, 20);
      float msw = time_ms([&] { hipLaunchKernelGGL(fill_kernel, grid, block, 0, 0, a, nb); }, 20);
      printf("%4zu MiB footprint (Infinity Cache is 256 MiB), repeated: read %.2f TB/s ; fill %.2f TB/s\n", mb, (mb << 20) / msr * 1e-9, (mb << 20) / msw * 1e-9);
    }
    CHECK(hipFree(a)); CHECK(hipFree(b));
  }
  return 0;
}
