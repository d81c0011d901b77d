// Standalone timing of the GEMM kernels at the shapes of the factorisation (diagnostic, not shipped).
#define GPX_GEMM_STAMPS 1
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../gpax_amd/csrc/gemm.hip"
void gpx_set_error(gpx_handle*, const std::string& m) { printf("error: %s\n", m.c_str()); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  const int64_t npad = 4096, rows = 4096 + 1024, nb = 128, ld = npad;
  double *A, *L; CK(hipMalloc(&A, rows * ld * 8));
  if (getenv("ALIAS")) L = A; else CK(hipMalloc(&L, rows * ld * 8));   // ALIAS=1: panel read from the matrix itself (in-place footprint, 168 MB instead of 336)
  std::vector<double> h(rows * ld);
  srand(3); for (auto& v : h) v = rand() / (double)RAND_MAX - 0.5;
  CK(hipMemcpy(A, h.data(), rows * ld * 8, hipMemcpyHostToDevice)); if (L != A) CK(hipMemcpy(L, h.data(), rows * ld * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned long long* dst; CK(hipMalloc(&dst, 64 * 8)); CK(hipMemset(dst, 0, 64 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), &dst, sizeof(dst)));
  for (int k : {0, 4, 8, 16, 24}) {
    const int64_t c0 = k * nb, r0 = c0 + nb, r1 = r0 + nb, mr = rows - r1, ncols = npad - r1;
    GemmLaunch t;
    t.A = L + r1 * ld + c0; t.lda = ld; t.B = L + r1 * ld + c0; t.ldb = ld; t.C = A + r1 * ld + r1; t.ldc = ld;
    t.M = mr; t.N = ncols; t.K = nb; t.alpha = -1.0; t.beta = 1.0; t.lower = 1; t.row_off = r1; t.col_off = r1; t.b_lower_tri = 0;
    const double entries = (double)ncols * (ncols + 1) / 2.0 + (double)(mr - ncols) * ncols;
    for (int variant = 0; variant < 1; ++variant) {
      t.small_tile = 0;
      for (int w = 0; w < 3; ++w) launch_gemm_nt(nullptr, 0, t);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) launch_gemm_nt(nullptr, 0, t); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
      printf("trailing k=%2d M=%lld N=%lld variant %d: %.1f us  %.1f TFLOP/s\n", k, (long long)mr, (long long)ncols, variant, ms * 1e3, 2.0 * entries * nb / ms * 1e-9);
      if (variant == 0) {
        unsigned long long st[16]; CK(hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost));
        printf("   one workgroup: first load+store %llu | ", st[1] - st[0]);
        for (int c = 0; c < 4; ++c) printf("chunk %d: barrier %llu, mfma %llu, (next store) %llu | ", c, st[2 + 3 * c] - st[1 + 3 * c], st[3 + 3 * c] - st[2 + 3 * c], (c < 3 ? st[4 + 3 * c] : st[13]) - st[3 + 3 * c]);
        printf(" total to epilogue %llu, epilogue (C wait + stores issued) %llu cycles\n", st[13] - st[0], st[14] - st[13]);
      }
    }
    // panel-solve shape
    GemmLaunch g; g.A = A + r0 * ld + c0; g.lda = ld; g.B = L; g.ldb = nb; g.C = L + r0 * ld + c0; g.ldc = ld;
    g.M = rows - r0; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0; g.lower = 0; g.row_off = 0; g.col_off = 0; g.b_lower_tri = 0; g.small_tile = 1;
    for (int w = 0; w < 3; ++w) launch_gemm_nt(nullptr, 0, g);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) launch_gemm_nt(nullptr, 0, g); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("panel    k=%2d M=%lld: %.1f us  %.1f TFLOP/s\n", k, (long long)g.M, ms * 1e3, 2.0 * g.M * nb * nb / ms * 1e-9);
  }
  return 0;
}
