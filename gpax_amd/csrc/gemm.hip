// gemm.hip -- fp64 MFMA "NT" GEMM for gfx950:  C[M,N] = beta*C + alpha * A[M,K] * B[N,K]^T
//
// The one GEMM shape the factorisation needs (both operands K-contiguous, row-major):
//   panel solve      L_ik = A_ik * inv(L_kk)^T          (replaces the TRSM of cho_factor, gpax/gp.py:296)
//   trailing update  A_ij -= L_ik * L_jk^T              (SYRK/GEMM of the right-looking Cholesky)
//   posterior        S_j -= X_k * L_jk^T, cov -= V^T V  (cho_solve + matmul of gpax/gp.py:297-299)
//
// Built on v_mfma_f64_16x16x4_f64 (__builtin_amdgcn_mfma_f64_16x16x4f64):
//   A operand: lane l holds A[i = l&15][k = l>>4];  B operand: lane l holds B[k = l>>4][j = l&15];
//   C/D: 4 doubles per lane, col = l&15, row = (l>>4) + 4*reg   (the f64 map, NOT the f32 one).
// One f64 MFMA is 2048 flop and occupies the SIMD's matrix pipe for 64 cycles, so the
// kernel is MFMA-bound with very little operand traffic: a 2x2 register block per wave
// needs only 4 ds_read_b64 per 4 MFMAs (256 cycles).  Tiles are staged global -> registers
// -> LDS (one register set, written after the barrier; the next tile's global loads are in
// flight during the MFMAs).  LDS rows are padded to 34 doubles: row stride = 68 dwords = 4 (mod
// 64) makes the 16-row x 2-k fragment read of each 32-lane half hit 64 distinct banks.

#include "gpx_internal.h"

struct GemmArgs {
  const double* A; const double* B; double* C;
  int64_t lda, ldb, ldc;
  int64_t M, N, K;
  double alpha, beta;
  int64_t row_off, col_off;
  int lower, b_lower_tri;
};

#define GEMM_BK 32
#define GEMM_LDT (GEMM_BK + 2)

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g) {
  constexpr int TM = BM / 32, TN = BN / 32;        // 16x16 MFMA tiles per wave (waves as 2 x 2)
  constexpr int NA = BM * (GEMM_BK / 2) / 256;      // 16-B chunks per thread, A tile
  constexpr int NBc = BN * (GEMM_BK / 2) / 256;
  __shared__ __attribute__((aligned(16))) double As[BM * GEMM_LDT];
  __shared__ __attribute__((aligned(16))) double Bs[BN * GEMM_LDT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  if (g.lower && (g.row_off + m0 + BM - 1 < g.col_off + n0)) return;  // tile strictly above the diagonal

  int64_t kend = g.K;
  if (g.b_lower_tri) {  // B[n][k] == 0 for k > n: K beyond the tile's last row is dead
    int64_t kk = (n0 + BN + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    if (kk < kend) kend = kk;
  }

  v4d acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

  v2d ra[NA], rb[NBc];
  const double* Ag = g.A + m0 * g.lda;
  const double* Bg = g.B + n0 * g.ldb;

  auto load_tiles = [&](int64_t k0) {
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int c = tid + it * 256;
      ra[it] = *reinterpret_cast<const v2d*>(Ag + (int64_t)(c >> 4) * g.lda + k0 + 2 * (c & 15));
    }
#pragma unroll
    for (int it = 0; it < NBc; ++it) {
      const int c = tid + it * 256;
      rb[it] = *reinterpret_cast<const v2d*>(Bg + (int64_t)(c >> 4) * g.ldb + k0 + 2 * (c & 15));
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int c = tid + it * 256;
      *reinterpret_cast<v2d*>(&As[(c >> 4) * GEMM_LDT + 2 * (c & 15)]) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < NBc; ++it) {
      const int c = tid + it * 256;
      *reinterpret_cast<v2d*>(&Bs[(c >> 4) * GEMM_LDT + 2 * (c & 15)]) = rb[it];
    }
  };

  const int fr = lane & 15, fk = lane >> 4;
  const double* Ap = &As[(wm * (BM / 2) + fr) * GEMM_LDT + fk];
  const double* Bp = &Bs[(wn * (BN / 2) + fr) * GEMM_LDT + fk];

  load_tiles(0);
  for (int64_t k0 = 0; k0 < kend; k0 += GEMM_BK) {
    store_tiles();
    __syncthreads();
    if (k0 + GEMM_BK < kend) load_tiles(k0 + GEMM_BK);  // in flight during the MFMAs below
#pragma unroll
    for (int s = 0; s < GEMM_BK / 4; ++s) {
      double a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Ap[i * 16 * GEMM_LDT + 4 * s];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bp[j * 16 * GEMM_LDT + 4 * s];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue: lane (col = l&15, row group = l>>4), reg r -> row = (l>>4) + 4r
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + wn * (BN / 2) + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * (BM / 2) + i * 16 + fk + 4 * r;
        double* cp = g.C + row * g.ldc + col;
        double v = g.alpha * acc[i][j][r];
        if (g.beta != 0.0) v += g.beta * (*cp);
        *cp = v;
      }
    }
}

// ---- latency variant for K = nb (64 or 128): one pass, one barrier ---------------------------
// The panel solve and the look-ahead update sit on the critical path of the factorisation and
// are small (M x nb x nb), so what matters is the latency of ONE tile, not throughput: the whole
// K range of the 32-row A and B tiles is fetched with every 16-B load of the workgroup in flight
// at once (coalesced 256-B row segments), staged through LDS once, and each wave then runs its
// K/4 MFMAs back to back on two accumulators (a dependent f64 MFMA issues every ~190 cycles, an
// independent one every ~100).  One global round trip and one barrier per tile instead of K/32.
// (A register-only variant with per-lane contiguous K chunks was tried and measured slower:
// its loads touch 64 different cache lines per instruction.)
template <int K>
__global__ __launch_bounds__(256) void gemm_nt_onepass_kernel(GemmArgs g) {
  constexpr int LD = K + 2;                 // row stride 2 (mod 32) doubles: conflict-free fragment reads
  constexpr int NCH = 32 * (K / 2) / 256;   // 16-B chunks per thread and operand
  __shared__ __attribute__((aligned(16))) double As[32 * LD];
  __shared__ __attribute__((aligned(16))) double Bs[32 * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * 32, n0 = (int64_t)blockIdx.x * 32;
  if (g.lower && (g.row_off + m0 + 31 < g.col_off + n0)) return;
  const double* Ag = g.A + m0 * g.lda;
  const double* Bg = g.B + n0 * g.ldb;
  v2d ra[NCH], rb[NCH];
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int c = tid + it * 256;
    ra[it] = *reinterpret_cast<const v2d*>(Ag + (int64_t)(c / (K / 2)) * g.lda + 2 * (c % (K / 2)));
    rb[it] = *reinterpret_cast<const v2d*>(Bg + (int64_t)(c / (K / 2)) * g.ldb + 2 * (c % (K / 2)));
  }
  const int fr = lane & 15, fk = lane >> 4;
  double* cp = g.C + (m0 + wm * 16 + fk) * g.ldc + n0 + wn * 16 + fr;   // row = fk + 4r, col = fr
  v4d cin = {0.0, 0.0, 0.0, 0.0};
  if (g.beta != 0.0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) cin[r] = cp[(int64_t)(4 * r) * g.ldc];
  }
#pragma unroll
  for (int it = 0; it < NCH; ++it) {
    const int c = tid + it * 256;
    *reinterpret_cast<v2d*>(&As[(c / (K / 2)) * LD + 2 * (c % (K / 2))]) = ra[it];
    *reinterpret_cast<v2d*>(&Bs[(c / (K / 2)) * LD + 2 * (c % (K / 2))]) = rb[it];
  }
  __syncthreads();
  const double* Ap = &As[(wm * 16 + fr) * LD + fk];
  const double* Bp = &Bs[(wn * 16 + fr) * LD + fk];
  v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < K / 4; s += 2) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[4 * s], Bp[4 * s], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[4 * s + 4], Bp[4 * s + 4], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double v = g.alpha * (acc0[r] + acc1[r]);
    if (g.beta != 0.0) v += g.beta * cin[r];
    cp[(int64_t)(4 * r) * g.ldc] = v;
  }
}

int launch_gemm_nt(gpx_handle* h, hipStream_t s, const GemmLaunch& L) {
  if (L.M <= 0 || L.N <= 0) return GPX_OK;
  const int bm = L.small_tile ? 32 : 64;
  if (L.M % bm || L.N % bm || L.K % GEMM_BK || L.K <= 0 || (L.lda & 1) || (L.ldb & 1) ||
      (reinterpret_cast<uintptr_t>(L.A) & 15) || (reinterpret_cast<uintptr_t>(L.B) & 15)) {
    gpx_set_error(h, "gemm_nt: shape/alignment contract violated (M,N % tile, K % 32, even ld, 16-B base)");
    return GPX_ERR_ARG;
  }
  GemmArgs g;
  g.A = L.A; g.B = L.B; g.C = L.C; g.lda = L.lda; g.ldb = L.ldb; g.ldc = L.ldc;
  g.M = L.M; g.N = L.N; g.K = L.K; g.alpha = L.alpha; g.beta = L.beta;
  g.row_off = L.row_off; g.col_off = L.col_off; g.lower = L.lower; g.b_lower_tri = L.b_lower_tri;
  dim3 grid((unsigned)(L.N / bm), (unsigned)(L.M / bm));
  if (L.small_tile && L.K == 128) hipLaunchKernelGGL((gemm_nt_onepass_kernel<128>), grid, dim3(256), 0, s, g);
  else if (L.small_tile && L.K == 64) hipLaunchKernelGGL((gemm_nt_onepass_kernel<64>), grid, dim3(256), 0, s, g);
  else if (L.small_tile) hipLaunchKernelGGL((gemm_nt_kernel<32, 32>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_nt_kernel<64, 64>), grid, dim3(256), 0, s, g);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}
