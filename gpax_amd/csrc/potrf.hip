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

// 64-bit form (one v_mov_b64_dpp)
template <int L>
__device__ __forceinline__ double row_bcast64(double x) {
  return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + L, 0xf, 0xf, false);
}

// acc += row_newbcast<L>(src) * mul  as ONE instruction (v_fmac_f64_dpp: 7.2 cycles issued from one
// wave, the same as a plain v_fma_f64; a v_mov_b64_dpp + v_fma_f64 pair costs 13.6 -- tools/dpp_probe).
// hipcc neither pads hazards inside an asm string nor knows this is a DPP read, so the statements
// are `volatile` (they keep the program order written here) and the one hazard that order leaves
// open -- a DPP read needs 2 wait states after a VALU write of its source VGPR -- is closed by
// dpp_fence() before each group instead of an s_nop per instruction (an s_nop 1 costs ~8 cycles here).
template <int L>
__device__ __forceinline__ void fmac_bcast(double& acc, double src, double mul) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(src), "v"(mul), "i"(L));
}
// Orders every earlier write of `x` before the statement and gives 2 wait states after it.
__device__ __forceinline__ void dpp_fence(double& x) { asm volatile("s_nop 1" : "+v"(x)); }

// 1/sqrt(d) and sqrt(d) from v_rsq_f64 (measured 2^-24.2 relative accuracy on gfx950,
// tools/rsq_probe) and ONE cubic (Halley) step:  y <- y (1 + e/2 + 3e^2/8),  e = 1 - d y^2,
// error O(e^3) ~ 1e-22 -> ~1 ulp after rounding.  Dependency depth 4 fp64 ops after the rsq:
// this chain is the critical path of the whole factorisation (one per matrix column).
// No denormal rescaling: pivots here are O(1e-6 .. 1e3).
__device__ __forceinline__ void sqrt_rsqrt(double d, double& s, double& rinv) {
  const double y = __builtin_amdgcn_rsq(d);
  const double t = d * y;
  const double e = fma(-t, y, 1.0);
  const double c = fma(0.375, e, 0.5);
  const double ye = y * e;
  rinv = fma(ye, c, y);
  s = d * rinv;
}

__device__ __forceinline__ int tix(int ib, int jb) { return ib * (ib + 1) / 2 + jb; }
__device__ __forceinline__ int eix(int r, int c) { return r * 16 + (c ^ (r & 14)); }

// Diagnostic build only (tools/potrf_stamps): s_memtime stamps per phase, written to a buffer of
// their own; in the product build GPX_STAMP expands to nothing.
#ifdef GPX_POTRF_STAMPS
__device__ unsigned long long* g_stamps;
#define GPX_STAMP(slot)                                                              \
  do {                                                                               \
    if (threadIdx.x == 0) g_stamps[slot] = __builtin_amdgcn_s_memtime();             \
  } while (0)
#else
#define GPX_STAMP(slot) do { } while (0)
#endif

// ---- tile-level building blocks (one wave each; all operands are LDS tiles) ----

// X = A * inv(L_jj)^T  in place:  A[i][k], B[k][j] = Wd[j][k]; two accumulators halve the
// dependent-MFMA chain (a dependent f64 MFMA issues every ~190 cycles, an independent one every ~100)
__device__ __forceinline__ void tile_trsm(double* At, const double* Wd, int fr, int fk) {
  v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(At[eix(fr, fk)], Wd[eix(fr, fk)], acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(At[eix(fr, 4 + fk)], Wd[eix(fr, 4 + fk)], acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(At[eix(fr, 8 + fk)], Wd[eix(fr, 8 + fk)], acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(At[eix(fr, 12 + fk)], Wd[eix(fr, 12 + fk)], acc1, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) At[eix(fk + 4 * r, fr)] = acc0[r] + acc1[r];
}

// C -= A * B^T :  A[i][k] = -Ap[i][k], B[k][j] = Bp[j][k]
__device__ __forceinline__ void tile_syrk(double* Cp, const double* Ap, const double* Bp, int fr, int fk) {
  v4d acc0, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) acc0[r] = Cp[eix(fk + 4 * r, fr)];
  acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, fk)], Bp[eix(fr, fk)], acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, 4 + fk)], Bp[eix(fr, 4 + fk)], acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, 8 + fk)], Bp[eix(fr, 8 + fk)], acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, 12 + fk)], Bp[eix(fr, 12 + fk)], acc1, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) Cp[eix(fk + 4 * r, fr)] = acc0[r] + acc1[r];
}

// W tile (jb, cb), cb < jb:  T = sum_{kb=cb}^{jb-1} L_jb,kb * W_kb,cb ;  W_jb,cb = -inv(L_jj) * T
__device__ __forceinline__ void tile_winv(double* Lt, double* Wt, int jb, int cb, int fr, int fk) {
  double* Tp = Wt + tix(jb, cb) * 256;
  const double* Wd = Wt + tix(jb, jb) * 256;
  v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  for (int kb = cb; kb < jb; ++kb) {
    const double* Lp = Lt + tix(jb, kb) * 256;
    const double* Wp = Wt + tix(kb, cb) * 256;
    // A[i][k] = L_jb,kb[i][k] , B[k][j] = W_kb,cb[k][j]
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, fk)], Wp[eix(fk, fr)], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, 4 + fk)], Wp[eix(4 + fk, fr)], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, 8 + fk)], Wp[eix(8 + fk, fr)], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, 12 + fk)], Wp[eix(12 + fk, fr)], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) Tp[eix(fk + 4 * r, fr)] = acc0[r] + acc1[r];
  // same-wave LDS round trip (LDS operations of one wave execute in order): T as the B operand
  v4d a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
  a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, fk)], Tp[eix(fk, fr)], a0, 0, 0, 0);
  a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, 4 + fk)], Tp[eix(4 + fk, fr)], a1, 0, 0, 0);
  a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, 8 + fk)], Tp[eix(8 + fk, fr)], a0, 0, 0, 0);
  a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, 12 + fk)], Tp[eix(12 + fk, fr)], a1, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) Tp[eix(fk + 4 * r, fr)] = a0[r] + a1[r];
}

// 16x16 leaf: Cholesky of the diagonal tile AND its inverse, one matrix row per lane (the 4
// DPP rows of the wave work redundantly), pivot column / pivot row broadcast by DPP.
__device__ __forceinline__ void leaf_potrf_inv(double* Ld, double* Wd, int lane, int* info, int gcol0) {
  const int i = lane & 15;
  double a[16], w[16];
  int first_bad = -1;
  double my_rinv = 1.0;  // 1 / l_ii of this lane's own row (row scaling of inv(L_jj) is deferred)
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    a[c] = Ld[eix(i, c)];
    w[c] = (c == i) ? 1.0 : 0.0;
  }
  static_for<0, 16>([&](auto J) {
    constexpr int j = decltype(J)::value;
    dpp_fence(a[j]);                                // a[j] was written inside an asm (invisible to hipcc's hazard pass)
    const double d = row_bcast64<j>(a[j]);          // pivot a_jj (after updates)
    first_bad = (!(d > 0.0) && first_bad < 0) ? j : first_bad;  // non-PD (or NaN) pivot
    double ljj, rinv;
    sqrt_rsqrt(d, ljj, rinv);                       // NaN from here on when d <= 0 (JAX semantics)
    const bool eq = (i == j), gt = (i > j);
    const double l = a[j] * rinv;                   // column j of L (meaningful in lanes i > j)
    a[j] = eq ? ljj : l;
    my_rinv = eq ? rinv : my_rinv;
    // Rows <= j are finished: they get a zero multiplier instead of a predicate on every
    // update (entries above the diagonal hold don't-care values and are never read).
    const double lm = gt ? l : 0.0;                 // l_ij (0 for finished rows)
    const double nl = -lm;
    const double nlr = nl * rinv;                   // -l_ij / l_jj
    // rank-1 update of the trailing rows:  a_ic -= l_ij * l_cj   (c > j); c = j+1 first, it
    // feeds the next pivot
    double lms = lm;
    dpp_fence(lms);                                 // lm was just written by VALU selects
    static_for<j + 1, 16>([&](auto C) {
      constexpr int c = decltype(C)::value;
      fmac_bcast<c>(a[c], lms, nl);                 // a[c] += bcast_c(l) * (-l_ij)
    });
    // forward substitution of the identity on the UNSCALED rows u_i = l_ii * w_i:
    //   u_i -= (l_ij / l_jj) * u_j   (i > j);   w_i = u_i / l_ii is applied once at the end
    // (w[c] was last written >= 2 instructions ago except after short groups: fence once)
    dpp_fence(w[0]);
    static_for<0, j + 1>([&](auto C) {
      constexpr int c = decltype(C)::value;
      fmac_bcast<j>(w[c], w[c], nlr);               // w[c] += bcast_j(w[c]) * (-l_ij / l_jj)
    });
  });
#pragma unroll
  for (int c = 0; c < 16; ++c) w[c] *= my_rinv;
  // LAPACK-style info: 1-based global index of the first non-positive pivot
  if (first_bad >= 0 && lane == 0 && *info == 0) *info = gcol0 + first_bad + 1;
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      Ld[eix(i, c)] = (c <= i) ? a[c] : 0.0;
      Wd[eix(i, c)] = (c <= i) ? w[c] : 0.0;
    }
  }
}

// Schedule (NT = NB/16 sub-panels; wave 0 carries the critical path):
//   leaf(0) | for jb: phase A (sub-panel solve, all waves) | { wave 0: update tile (jb+1,jb+1),
//   leaf(jb+1) }  ||  { other waves: the rest of the rank-16 update, then row jb of W }
// so the in-block trailing update and the whole inversion hide behind the leaf chain.
template <int NB, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void potrf_diag_kernel(const double* __restrict__ A,
                                                                 int64_t lda, double* __restrict__ L,
                                                                 int64_t ldl, double* __restrict__ W,
                                                                 int* info, int col0) {
  constexpr int NT = NB / 16, NTRI = NT * (NT + 1) / 2;
  constexpr int TPW = (NTRI + NWAVES - 1) / NWAVES;   // lower tiles per wave (load phase)
  __shared__ double sm[2 * NTRI * 256];
  double* Lt = sm;
  double* Wt = sm + NTRI * 256;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  GPX_STAMP(0);

  // ---- load the lower tiles of A_kk: all global loads of a wave in flight, then the LDS writes
  {
    const int r = lane >> 2, c4 = (lane & 3) * 4;
    double v[TPW][4];
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int t = wave + u * NWAVES;
      if (t < NTRI) {
        int ib = 0, jb = t;
        while (jb > ib) { jb -= ib + 1; ++ib; }
        const double* src = A + (int64_t)(ib * 16 + r) * lda + jb * 16 + c4;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] = src[q];
      }
    }
#pragma unroll
    for (int u = 0; u < TPW; ++u) {
      const int t = wave + u * NWAVES;
      if (t < NTRI) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Lt[t * 256 + eix(r, c4 + q)] = v[u][q];
      }
    }
    // zeros above the diagonal of the L block and of W (nothing depends on them)
    for (int t = wave; t < NT * NT; t += NWAVES) {
      const int ib = t / NT, jb = t % NT;
      if (jb > ib) {
        double* ldst = L + (int64_t)(ib * 16 + r) * ldl + jb * 16 + c4;
        double* wdst = W + (int64_t)(ib * 16 + r) * NB + jb * 16 + c4;
#pragma unroll
        for (int q = 0; q < 4; ++q) { ldst[q] = 0.0; wdst[q] = 0.0; }
      }
    }
  }
  __syncthreads();
  GPX_STAMP(1);
  if (wave == 0) leaf_potrf_inv(Lt + tix(0, 0) * 256, Wt + tix(0, 0) * 256, lane, info, col0);
  __syncthreads();

#pragma unroll 1
  for (int jb = 0; jb < NT; ++jb) {
    GPX_STAMP(2 + 2 * jb);
    // ---- phase A: sub-panel solve, tiles (ib, jb), ib > jb; wave 0 takes (jb+1, jb)
    {
      const double* Wd = Wt + tix(jb, jb) * 256;
      for (int ib = jb + 1 + wave; ib < NT; ib += NWAVES) tile_trsm(Lt + tix(ib, jb) * 256, Wd, fr, fk);
    }
    __syncthreads();
    GPX_STAMP(3 + 2 * jb);
    // ---- shadow slot: critical path on wave 0, everything else on the other waves
    if (jb + 1 < NT) {
      if (wave == 0) {
        double* Cp = Lt + tix(jb + 1, jb + 1) * 256;
        const double* Ap = Lt + tix(jb + 1, jb) * 256;
        tile_syrk(Cp, Ap, Ap, fr, fk);
        leaf_potrf_inv(Cp, Wt + tix(jb + 1, jb + 1) * 256, lane, info, col0 + (jb + 1) * 16);
      } else {
        const int tn = NT - 1 - jb;
        const int ntiles = tn * (tn + 1) / 2;
        for (int q = wave; q < ntiles + jb; q += NWAVES - 1) {   // q = 0 is wave 0's tile
          if (q < ntiles) {
            int ib2 = 0, q2 = q;
            while (q2 > ib2) { q2 -= ib2 + 1; ++ib2; }
            const int ib = jb + 1 + ib2, lb = jb + 1 + q2;
            tile_syrk(Lt + tix(ib, lb) * 256, Lt + tix(ib, jb) * 256, Lt + tix(lb, jb) * 256, fr, fk);
          } else {
            tile_winv(Lt, Wt, jb, q - ntiles, fr, fk);   // row jb of W (longest chains last is fine: off the critical path)
          }
        }
      }
    } else {
      // last row of W: nothing left to hide behind, all waves share it
      for (int cb = wave; cb < jb; cb += NWAVES) tile_winv(Lt, Wt, jb, cb, fr, fk);
    }
    __syncthreads();
  }

  GPX_STAMP(2 + 2 * NT);
  // ---- write L_kk (lower) and W_k (lower) tiles
  {
    const int r = lane >> 2, c4 = (lane & 3) * 4;
    for (int t = wave; t < NTRI; t += NWAVES) {
      int ib = 0, jb = t;
      while (jb > ib) { jb -= ib + 1; ++ib; }
      double* ldst = L + (int64_t)(ib * 16 + r) * ldl + jb * 16 + c4;
      double* wdst = W + (int64_t)(ib * 16 + r) * NB + jb * 16 + c4;
      const double* ls = Lt + t * 256;
      const double* ws = Wt + t * 256;
      double lv[4], wv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { lv[q] = ls[eix(r, c4 + q)]; wv[q] = ws[eix(r, c4 + q)]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) { ldst[q] = lv[q]; wdst[q] = wv[q]; }
    }
  }
  GPX_STAMP(3 + 2 * NT);
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
