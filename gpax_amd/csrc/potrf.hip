// potrf.hip -- LDS-resident factorisation of one diagonal block (gfx950).
//
// One workgroup takes the NB x NB diagonal block A_kk of the right-looking blocked
// Cholesky (the `cho_factor` of gpax/gp.py:296 / the Cholesky inside
// dist.MultivariateNormal, gpax/gp.py:147-151), keeps it in LDS as 16 x 16 tiles, and
// produces BOTH
//     L_kk          (lower Cholesky factor of the block), and
//     W_k = L_kk^-1 (lower triangular),
// so that every other step of the factorisation (panel solve, posterior solves) is a
// plain MFMA GEMM against W_k instead of a triangular solve.
//
// Per 16-column sub-panel jb:
//   leaf   one wave factors the 16x16 diagonal tile with one matrix ROW per lane: the
//          pivot column / pivot row are broadcast with DPP `row_newbcast` (no LDS, no
//          readlane), and the same sweep forward-substitutes the identity, giving
//          inv(L_jj) for free;
//   phase A  (all waves, one tile each)  L_ib,jb = A_ib,jb * inv(L_jj)^T        [4 MFMA]
//            and row jb of W:  W_jb,cb = -inv(L_jj) * sum_kb L_jb,kb W_kb,cb   [4(jb-cb)+4 MFMA]
//   phase B  trailing tiles  A_ib,lb -= L_ib,jb * L_lb,jb^T                     [4 MFMA]
// Tiles live un-padded in LDS (2 KiB each; 36 + 36 tiles = 144 KiB at NB = 128) with the
// XOR swizzle  elem(r,c) -> r*16 + (c ^ (r & 14)), which makes both the row-fragment
// and the column-fragment ds_read_b64 patterns of the f64 MFMA conflict-free.

#include <type_traits>

#include "gpx_internal.h"

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// broadcast lane L of every 16-lane row to the whole row (DPP row_newbcast, gfx90a+)
template <int L>
__device__ __forceinline__ double row_bcast(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + L, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + L, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int tix(int ib, int jb) { return ib * (ib + 1) / 2 + jb; }
__device__ __forceinline__ int eix(int r, int c) { return r * 16 + (c ^ (r & 14)); }

template <int NB, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void potrf_diag_kernel(const double* __restrict__ A,
                                                                 int64_t lda, double* __restrict__ L,
                                                                 int64_t ldl, double* __restrict__ W,
                                                                 int* info, int col0) {
  constexpr int NT = NB / 16, NTRI = NT * (NT + 1) / 2;
  __shared__ double sm[2 * NTRI * 256];
  double* Lt = sm;
  double* Wt = sm + NTRI * 256;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;

  // ---- load the lower tiles of A_kk (4 lanes x 32 B per tile row)
  {
    const int r = lane >> 2, c4 = (lane & 3) * 4;
    int t = 0;
    for (int ib = 0; ib < NT; ++ib)
      for (int jb = 0; jb <= ib; ++jb, ++t) {
        if (t % NWAVES != wave) continue;
        const double* src = A + (int64_t)(ib * 16 + r) * lda + jb * 16 + c4;
        double* dst = Lt + t * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[eix(r, c4 + q)] = src[q];
      }
  }
  __syncthreads();

#pragma unroll 1
  for (int jb = 0; jb < NT; ++jb) {
    // ------------------------------------------------------------------ leaf (wave 0)
    if (wave == 0) {
      double* Ld = Lt + tix(jb, jb) * 256;
      double* Wd = Wt + tix(jb, jb) * 256;
      const int i = fr;  // matrix row held by this lane (the 4 DPP rows work redundantly)
      double a[16], w[16];
      int first_bad = -1;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        a[c] = Ld[eix(i, c)];
        w[c] = (c == i) ? 1.0 : 0.0;
      }
      static_for<0, 16>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const double d = row_bcast<j>(a[j]);            // pivot a_jj (after updates)
        first_bad = (!(d > 0.0) && first_bad < 0) ? j : first_bad;  // non-PD (or NaN) pivot
        const double ljj = sqrt(d);                     // NaN from here on when d < 0 (JAX semantics)
        const double rinv = 1.0 / ljj;
        const bool eq = (i == j), gt = (i > j);
        const double l = a[j] * rinv;                   // column j of L (meaningful in lanes i > j)
        a[j] = eq ? ljj : l;
        // Rows <= j are finished: give them a zero multiplier instead of predicating every
        // update (entries above the diagonal hold don't-care values and are never read).
        const double leff = gt ? l : 0.0;
        const double sc = eq ? rinv : 1.0;
        // rank-1 update of the trailing rows:  a_ic -= l_ij * l_cj   (c > j)
        static_for<j + 1, 16>([&](auto C) {
          constexpr int c = decltype(C)::value;
          const double lc = row_bcast<c>(leff);
          a[c] = fma(-leff, lc, a[c]);
        });
        // forward substitution of the identity: row j of inv(L_jj) becomes final in lane j
        // (w_j <- w_j / l_jj), rows below take  w_i -= l_ij * w_j
        static_for<0, j + 1>([&](auto C) {
          constexpr int c = decltype(C)::value;
          const double ws = w[c] * sc;
          const double wj = row_bcast<j>(ws);
          w[c] = fma(-leff, wj, ws);
        });
      });
      // LAPACK-style info: 1-based global index of the first non-positive pivot
      if (first_bad >= 0 && lane == 0 && *info == 0) *info = col0 + jb * 16 + first_bad + 1;
      if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          Ld[eix(i, c)] = (c <= i) ? a[c] : 0.0;
          Wd[eix(i, c)] = (c <= i) ? w[c] : 0.0;
        }
      }
    }
    __syncthreads();

    // -------------------------------------------- phase A: sub-panel solve + row jb of W
    {
      const double* Wd = Wt + tix(jb, jb) * 256;
      const int nsolve = NT - 1 - jb;
      for (int q = wave; q < NT - 1; q += NWAVES) {
        if (q < nsolve) {
          const int ib = jb + 1 + q;
          double* At = Lt + tix(ib, jb) * 256;
          v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int k = 4 * s + fk;
            // X = A * inv(L_jj)^T :  A[i][k] , B[k][j] = W_jj[j][k]
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(At[eix(fr, k)], Wd[eix(fr, k)], acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) At[eix(fk + 4 * r, fr)] = acc[r];
        } else {
          const int cb = q - nsolve;  // W tile (jb, cb), cb < jb
          double* Tp = Wt + tix(jb, cb) * 256;
          v4d acc = {0.0, 0.0, 0.0, 0.0};
          for (int kb = cb; kb < jb; ++kb) {
            const double* Lp = Lt + tix(jb, kb) * 256;
            const double* Wp = Wt + tix(kb, cb) * 256;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const int k = 4 * s + fk;
              // T += L_jb,kb * W_kb,cb :  A[i][k] , B[k][j] = W_kb,cb[k][j]
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, k)], Wp[eix(k, fr)], acc, 0, 0, 0);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) Tp[eix(fk + 4 * r, fr)] = acc[r];
          // same-wave LDS round trip (LDS ops of one wave execute in order)
          v4d acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int k = 4 * s + fk;
            // W_jb,cb = -inv(L_jj) * T :  A[i][k] = -W_jj[i][k] , B[k][j] = T[k][j]
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, k)], Tp[eix(k, fr)], acc2, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) Tp[eix(fk + 4 * r, fr)] = acc2[r];
        }
      }
    }
    __syncthreads();

    // ------------------------------------------------ phase B: trailing update in the block
    {
      const int tn = NT - 1 - jb;
      const int ntiles = tn * (tn + 1) / 2;
      for (int q = wave; q < ntiles; q += NWAVES) {
        int ib2 = 0, q2 = q;
        while (q2 > ib2) { q2 -= ib2 + 1; ++ib2; }
        const int ib = jb + 1 + ib2, lb = jb + 1 + q2;
        double* Cp = Lt + tix(ib, lb) * 256;
        const double* Ap = Lt + tix(ib, jb) * 256;
        const double* Bp = Lt + tix(lb, jb) * 256;
        v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = Cp[eix(fk + 4 * r, fr)];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int k = 4 * s + fk;
          // C -= L_ib,jb * L_lb,jb^T :  A[i][k] = -L_ib,jb[i][k] , B[k][j] = L_lb,jb[j][k]
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, k)], Bp[eix(fr, k)], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Cp[eix(fk + 4 * r, fr)] = acc[r];
      }
    }
    __syncthreads();
  }

  // ---- write L_kk (lower, zeros above) and W_k (dense NB x NB, zeros above)
  {
    const int r = lane >> 2, c4 = (lane & 3) * 4;
    for (int t = wave; t < NT * NT; t += NWAVES) {
      const int ib = t / NT, jb = t % NT;
      double* ldst = L + (int64_t)(ib * 16 + r) * ldl + jb * 16 + c4;
      double* wdst = W + (int64_t)(ib * 16 + r) * NB + jb * 16 + c4;
      if (jb <= ib) {
        const double* ls = Lt + tix(ib, jb) * 256;
        const double* ws = Wt + tix(ib, jb) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) { ldst[q] = ls[eix(r, c4 + q)]; wdst[q] = ws[eix(r, c4 + q)]; }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { ldst[q] = 0.0; wdst[q] = 0.0; }
      }
    }
  }
}

int launch_potrf_diag(gpx_handle* h, hipStream_t s, int nb, const double* A, int64_t lda, double* L,
                      int64_t ldl, double* W, int* info, int col0) {
  switch (nb) {
    case 64:
      hipLaunchKernelGGL((potrf_diag_kernel<64, 4>), dim3(1), dim3(256), 0, s, A, lda, L, ldl, W, info, col0);
      break;
    case 128:
      hipLaunchKernelGGL((potrf_diag_kernel<128, 8>), dim3(1), dim3(512), 0, s, A, lda, L, ldl, W, info, col0);
      break;
    default:
      gpx_set_error(h, "potrf_diag: nb must be 64 or 128");
      return GPX_ERR_ARG;
  }
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}
