// Diagnostic build of the diagonal-block kernel with s_memtime stamps per phase (never shipped:
// the product library is built without GPX_POTRF_STAMPS).  Prints where potrf_diag spends its cycles.
#define GPX_POTRF_STAMPS 1
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include "../gpax_amd/csrc/potrf.hip"

void gpx_set_error(gpx_handle*, const std::string& m) { printf("error: %s\n", m.c_str()); }

template <int NB>
static void run(int nwaves_label) {
  const int NT = NB / 16;
  std::vector<double> a(NB * NB);
  srand(1);
  std::vector<double> x(NB * 4);
  for (auto& v : x) v = rand() / (double)RAND_MAX * 2 - 1;
  for (int i = 0; i < NB; ++i)
    for (int j = 0; j < NB; ++j) {
      double r2 = 0; for (int k = 0; k < 4; ++k) { double d = x[i * 4 + k] - x[j * 4 + k]; r2 += d * d; }
      a[i * NB + j] = exp(-0.5 * r2) + (i == j ? 0.01 : 0.0);
    }
  double *dA, *dL, *dW; int* dinfo; unsigned long long* dst;
  hipMalloc(&dA, NB * NB * 8); hipMalloc(&dL, NB * NB * 8); hipMalloc(&dW, NB * NB * 8); hipMalloc(&dinfo, 4);
  hipMalloc(&dst, 512 * 8); hipMemset(dst, 0, 512 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof(dst));
  hipMemcpy(dA, a.data(), NB * NB * 8, hipMemcpyHostToDevice);
  hipMemset(dinfo, 0, 4);
  PotrfLaunch pl;
  pl.nb = NB; pl.A = dA; pl.lda = NB; pl.L = dL; pl.ldl = NB; pl.W = dW; pl.info = dinfo; pl.col0 = 0;
  for (int rep = 0; rep < 3; ++rep) launch_potrf_diag(nullptr, 0, pl);
  hipDeviceSynchronize();
  unsigned long long st[512]; hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost);
  printf("NB=%d (dataflow): barrier after %llu cycles, first leaf starts after %llu\n", NB, st[1] - st[0], st[2] - st[0]);
  for (int jb = 0; jb < NT; ++jb)
    printf("  stage %d: leaf %llu cycles, hand-over to the next leaf %lld\n", jb, st[3 + 2 * jb] - st[2 + 2 * jb],
           jb + 1 < NT ? (long long)(st[2 + 2 * (jb + 1)] - st[3 + 2 * jb]) : (long long)(st[2 + 2 * NT] - st[3 + 2 * jb]));
  for (int w = 1; w < NT; ++w) {
    printf("  wave %d:", w);
    for (int jb = 0; jb < w; ++jb) {
      const unsigned long long* e = st + 64 + w * 32 + jb * 4;
      printf("  [%d] %lld pre %lld solve %lld upd %lld |", jb, (long long)(e[0] - st[0]), e[3] ? (long long)(e[3] - e[0]) : 0ll,
             (long long)(e[1] - (e[3] ? e[3] : e[0])), e[2] ? (long long)(e[2] - e[1]) : 0ll);
    }
    printf("\n");
  }
  for (int w = 1; w < NT; ++w) {
    const unsigned long long* e = st + 64 + w * 32 + (w - 1) * 4;
    printf("  F1 wave %d: start %lld, groups final at +%lld +%lld +%lld +%lld, end +%lld\n", w, (long long)(e[0] - st[0]),
           (long long)(st[320 + w * 8] - e[0]), (long long)(st[320 + w * 8 + 1] - e[0]), (long long)(st[320 + w * 8 + 2] - e[0]),
           (long long)(st[320 + w * 8 + 3] - e[0]), (long long)(e[1] - e[0]));
  }
  printf("  leaf starts:");
  for (int jb = 0; jb < NT; ++jb) printf(" %llu..%llu", st[2 + 2 * jb] - st[0], st[3 + 2 * jb] - st[0]);
  printf("\n");
  printf("  whole kernel %llu cycles (%.1f us @2.4GHz); last wave done %llu\n", st[3 + 2 * NT] - st[0], (st[3 + 2 * NT] - st[0]) / 2400.0,
         st[2 + 2 * NT] - st[0]);
  // correctness spot check
  std::vector<double> l(NB * NB), w(NB * NB);
  hipMemcpy(l.data(), dL, NB * NB * 8, hipMemcpyDeviceToHost);
  hipMemcpy(w.data(), dW, NB * NB * 8, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0;
  for (int i = 0; i < NB; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0, t = 0;
      for (int k = 0; k <= j; ++k) s += l[i * NB + k] * l[j * NB + k];
      for (int k = j; k <= i; ++k) t += l[i * NB + k] * w[k * NB + j];
      e1 = fmax(e1, fabs(s - a[i * NB + j]));
      e2 = fmax(e2, fabs(t - (i == j ? 1.0 : 0.0)));
    }
  printf("  max|LL^T-A| = %.2e  max|L W - I| = %.2e\n", e1, e2);
}


// ---- probe: the cost of one streaming solve when every column is already published (one wave alone, or 8 waves at once)
__global__ __launch_bounds__(512) void stream_probe(double* out, unsigned long long* clk, int mode) {
  __shared__ double dt[DT_STRIDE];
  __shared__ double tile[8][256];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fk = lane >> 4;
  for (int t = tid; t < DT_STRIDE; t += blockDim.x) dt[t] = 0.0;
  __syncthreads();
  for (int t = tid; t < 256; t += blockDim.x) { const int c = t / 16, r = t % 16; dt[t] = (r == c) ? 2.0 : (r > c ? 0.01 * (r - c) : 0.0); }
  if (tid < 16) dt[DT_RINV + tid] = 0.5;
  for (int t = lane; t < 256; t += 64) tile[w][t] = 1.0 + 0.001 * t;
  __syncthreads();
  if (tid == 0) reinterpret_cast<int*>(dt + DT_EPOCH)[0] = 16;
  __syncthreads();
  int info = 0;
  double* Xt = tile[w];
  if (mode >= 2 && w != 0) {
    // partner load: waves on wave 0's SIMD (w == 4) or elsewhere run back-to-back MFMAs (mode 2: independent, 3: dependent)
    if ((mode & 4) ? (w == 1) : (w == 4)) {
      v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      double x = tile[w][lane];
      const unsigned long long m0 = __builtin_amdgcn_s_memtime();
      for (int it = 0; it < 600; ++it) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a0, 0, 0, 0);
        if ((mode & 3) == 2) {
          a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a3, 0, 0, 0);
        }
      }
      out[tid] = a0[0] + a1[0] + a2[0] + a3[0];
      const unsigned long long m1 = __builtin_amdgcn_s_memtime();
      if (lane == 0) clk[7] = (m1 - m0) / (((mode & 3) == 2) ? 2400 : 600);
    }
    return;
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < 16; ++rep) {
    double u[1][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) u[0][q] = Xt[eix(fk + 4 * q, fr)];
    if (mode == 0) {
      stream_solve<1, false>(dt, u, lane, &info, [&](auto S, double rk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Xt[eix(fk + 4 * q, fr)] = u[0][q] * rk;
      });
    } else {
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      double xp = 0.0;
      stream_solve<1, true>(dt, u, lane, &info, [&](auto S, double rk) {
        constexpr int s = decltype(S)::value;
        if constexpr (s > 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xp, xp, acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) Xt[eix(fk + 4 * q, fr)] = u[0][q] * rk;
        xp = Xt[eix(fr, 4 * s + fk)];
      });
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xp, xp, acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) Xt[eix(fk + 4 * q, fr)] += 1e-30 * acc[q];
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) clk[w] = (t1 - t0) / 16;
  out[tid] = Xt[lane];
}
void run_probe() {
  double* out; unsigned long long* clk;
  hipMalloc(&out, 512 * 8); hipMalloc(&clk, 8 * 8);
  for (int mode = 2; mode <= 7; ++mode) {
    if ((mode & 3) < 2) continue;
    hipLaunchKernelGGL(stream_probe, dim3(1), dim3(512), 0, 0, out, clk, mode);
    hipLaunchKernelGGL(stream_probe, dim3(1), dim3(512), 0, 0, out, clk, mode);
    hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    printf("stream_solve probe (F1 form) with a wave of %s MFMAs on %s SIMD: %llu cycles per solve\n", (mode & 3) == 2 ? "independent" : "dependent",
           (mode & 4) ? "another" : "the same", h[0]);
    printf("    (that wave: %llu cycles per MFMA)\n", h[7]);
  }
  for (int nw = 1; nw <= 8; nw *= 2)
    for (int mode = 0; mode < 2; ++mode) {
      hipLaunchKernelGGL(stream_probe, dim3(1), dim3(64 * nw), 0, 0, out, clk, mode);
      hipLaunchKernelGGL(stream_probe, dim3(1), dim3(64 * nw), 0, 0, out, clk, mode);
      hipDeviceSynchronize();
      unsigned long long h[8]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
      printf("stream_solve probe: %d wave(s), %s: %llu cycles per solve (wave 0)\n", nw, mode ? "F1 form (MFMA per group)" : "plain", h[0]);
    }
}
int main() { run<64>(4); run<128>(8); run_probe(); return 0; }
