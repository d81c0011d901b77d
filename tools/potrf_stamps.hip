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
#if GPX_POTRF_FLOW
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
#else
  printf("NB=%d: load %llu cycles, first leaf %llu\n", NB, st[1] - st[0], st[2] - st[1]);
  printf("  first leaf: wave0 (potrf) done after %llu cycles, wave1 (inverse) after %llu\n", st[40] - st[1], st[41] - st[1]);
  unsigned long long pa = 0, sh = 0;
  for (int jb = 0; jb < NT; ++jb) {
    unsigned long long a_ = st[3 + 2 * jb] - st[2 + 2 * jb], b_ = st[2 + 2 * (jb + 1)] - st[3 + 2 * jb];
    printf("  jb=%d phaseA %llu  shadow slot (leaf || update+inverse) %llu\n", jb, a_, b_);
    pa += a_; sh += b_;
  }
  printf("  totals: phaseA %llu slots %llu store %llu ; whole kernel %llu cycles (%.1f us @2.4GHz)\n", pa, sh,
         st[3 + 2 * NT] - st[2 + 2 * NT], st[3 + 2 * NT] - st[0], (st[3 + 2 * NT] - st[0]) / 2400.0);
#endif
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

int main() { run<64>(4); run<128>(8); return 0; }
