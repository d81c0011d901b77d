// Standalone timing of the bulk-update GEMM (lower trapezoid, C -= A B^T) at a given shape: ./gemm_shape M N K
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../gpax_amd/csrc/gemm.hip"
void gpx_set_error(gpx_handle*, const std::string& m) { printf("error: %s\n", m.c_str()); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  const int64_t M = atoll(argv[1]), N = atoll(argv[2]);
  const int64_t ld = N + (getenv("LDPAD") ? atoll(getenv("LDPAD")) : 1024);
  double *A, *L; CK(hipMalloc(&A, M * ld * 8)); CK(hipMalloc(&L, M * ld * 8));
  CK(hipMemset(A, 0, M * ld * 8)); CK(hipMemset(L, 0, M * ld * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  for (int a = 3; a < argc; ++a) {
    const int64_t K = atoll(argv[a]);
    GemmLaunch t;
    t.A = L; t.lda = ld; t.B = L; t.ldb = ld; t.C = A; t.ldc = ld;
    t.M = M; t.N = N; t.K = K; t.alpha = -1.0; t.beta = getenv("BETA0") ? 0.0 : 1.0; t.lower = 1; t.row_off = 0; t.col_off = 0; t.b_lower_tri = 0;   // BETA0=1: no C read (counter calibration)
    t.small_tile = 0;
    double entries = (double)N * (N + 1) / 2.0 + (double)(M - N) * N;
    if (getenv("RECT")) {   // RECT=1: full M x N rectangle on the 64x64 kernel, RECT=2: on the one-pass 32x32 kernel (the look-ahead column updates)
      t.lower = 0; entries = (double)M * N; t.small_tile = atoi(getenv("RECT")) == 2;
    }
    auto go = [&]() { launch_gemm_nt(nullptr, 0, t); };
    for (int w = 0; w < 2; ++w) go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) go(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("M=%lld N=%lld K=%lld: %.1f us  %.1f TFLOP/s\n", (long long)M, (long long)N, (long long)K, ms * 1e3, 2.0 * entries * K / ms * 1e-9);
    if (getenv("GAPTEST")) {   // what sits between consecutive launches on one stream: nothing / a satisfied event wait / an attached stop event
      hipStream_t st, other; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      CK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo)); CK(hipStreamCreateWithPriority(&other, hipStreamNonBlocking, hi));
      hipEvent_t done[16], evs[16];
      for (int i = 0; i < 16; ++i) { CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming | hipEventDisableSystemFence)); CK(hipEventCreateWithFlags(&evs[i], hipEventDisableTiming | hipEventDisableSystemFence)); CK(hipEventRecord(done[i], other)); }
      CK(hipDeviceSynchronize());
      for (int mode = 0; mode < 4; ++mode) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 10; ++r) {
          if (mode == 1 || mode == 3) CK(hipStreamWaitEvent(st, done[r], 0));
          GemmLaunch u = t; u.stop_event = (mode >= 2) ? evs[r] : nullptr;
          launch_gemm_nt(nullptr, st, u);
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float m2; CK(hipEventElapsedTime(&m2, e0, e1));
        printf("   10 launches on a low-priority stream, %s: %.1f us per launch\n",
               mode == 0 ? "nothing between" : mode == 1 ? "a satisfied event wait before each" : mode == 2 ? "a stop event attached to each" : "both", m2 * 100.0);
      }
    }
  }
  return 0;
}
