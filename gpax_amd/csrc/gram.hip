// gram.hip -- tiled pairwise-distance / gram-matrix builder for gfx950.
//
// Replaces, in one fused kernel, the JAX-primitive sequence of
//   _squared_distance            gpax/kernels.py:39-51
//   RBFKernel._kernel_function   gpax/kernels.py:112-126
//   MaternKernel (nu=5/2)        gpax/kernels.py:131-147   (r = sqrt(r2 + 1e-12))
//   PeriodicKernel               gpax/kernels.py:152-168   (direct differences, sin)
// including the diagonal rule `(k_noise + k_jitter) * eye(n)`.
//
// HBM-bound: the output is written exactly once (8 B/entry, 16 B per lane per store) and the inputs are read from L2,
// so HBM traffic is the algorithmic 8*n1*n2 + 8*(n1+n2)*d bytes.
// Two tile kernels:
//   * gram_prep_kernel + gram_mfma_kernel (RBF / Matern, d <= 62; the default since round 3): the reference's own
//     |a|^2 - 2ab + |b|^2 form as one augmented matrix product on v_mfma_f64_16x16x4_f64, clipped at zero -- see the
//     comment in front of them;
//   * gram_kernel: direct differences sum_k (a_k - b_k)^2 on the vector ALU with the X blocks staged in LDS (periodic
//     kernels, whose distance is a sine per dimension, and d > 62; rounds 1-2 used it for every kernel).  The two forms
//     agree to a few 1e-15 (tests/test_oracle.py).

#include <stdlib.h>

#include "gpx_internal.h"

#define GT 64   // output tile edge (64 x 64 entries per 256-thread workgroup)
#define DC 16   // input-dimension chunk staged in LDS per pass

struct GramArgs {
  const double* x1; const double* x2;
  const double* hyp;     // length[d], period[d]
  const double* dev_scalars;  // device {k_scale, noise_scalar, jitter} or nullptr (then the by-value fields)
  const double* noise;   // per-row noise or nullptr
  const int64_t* row_gid;   // global id of each x1 row (distributed path) or nullptr
  double* out;
  int64_t n1, n2, ld;
  int d;
  int lower_only, diag_rule;
  double k_scale, noise_scalar, jitter;
  int64_t col0;      // global index of x2's first row / out's first column (column slices of one matrix)
  int* zero_info;    // when set, workgroup (0,0) clears it (saves the factorisation a memset launch)
  unsigned* zero_flags; int n_zero_flags;   // likewise: the hand-over flags of the split-panel schedule
  int64_t s_out, s_hyp, s_noise, s_info, s_flags;   // per-draw strides of a batched launch (blockIdx.z = draw; x1 / x2 are shared)
  // single draw, first launch of a step: the hyper-parameters arrive in the kernel arguments (hyp_n doubles: length[d], period[d],
  // scalars) and workgroup (0,0) writes them to `hyp_out` (= d_hyp) for the kernels behind this one
  int hyp_n; double* hyp_out; double hyp_in[GPX_HYP_INLINE];
};

// sqrt(a) for a in the normal range (here a = r2 + 1e-12 >= 1e-12): libdevice's sequence -- v_rsq_f64, one coupled
// Newton step on (g, h) = (sqrt a, 1 / (2 sqrt a)), two residual corrections -- without its rescaling of tiny arguments and
// its zero / infinity classification (seven instructions of 22); the same bits for every argument the kernels produce
__device__ __forceinline__ double gpx_sqrt_normal(const double a) {
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, a), h, g);
  g = __builtin_fma(__builtin_fma(-g, g, a), h, g);
  return g;
}

template <int KIND>
__device__ __forceinline__ double gram_value(const double r2, const double k_scale) {
  if (KIND == GPX_RBF) return k_scale * gpx_exp_neg(-0.5 * r2);                       // kernels.py:123
  if (KIND == GPX_PERIODIC) return k_scale * gpx_exp_neg(-2.0 * r2);                  // kernels.py:165
  const double rr = gpx_sqrt_normal(r2 + 1e-12);                                       // kernels.py:142
  const double s5r = 2.2360679774997896964 * rr;
  const double poly = __builtin_fma(5.0 / 3.0, r2, __builtin_fma(2.2360679774997896964, rr, 1.0));
  return k_scale * poly * gpx_exp_neg(-s5r);                                           // kernels.py:144
}

template <int KIND, bool VEC2>
__global__ __launch_bounds__(256) void gram_kernel(GramArgs g) {
  // X1 block row-major [row][k] (read as wave-uniform broadcasts), X2 block
  // transposed [k][col] (read as 16-B pairs by consecutive lanes: conflict-free).
  __shared__ double x1s[GT][DC];
  __shared__ __attribute__((aligned(16))) double x2s[DC][GT];
  __shared__ double len_s[DC], per_s[DC];

  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.y * GT, j0 = (int64_t)blockIdx.x * GT;
  gpx_gptr hyp = (gpx_gptr)(uintptr_t)g.hyp, dsc = (gpx_gptr)(uintptr_t)g.dev_scalars;
  if (g.hyp_n) {
    hyp = gpx_kernarg_doubles(offsetof(GramArgs, hyp_in)); dsc = hyp + 2 * g.d;
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid < g.hyp_n) g.hyp_out[tid] = hyp[tid];
  }
  if (blockIdx.z) {   // draw z of a batch: its own hyper-parameters, output matrix, info word and flags
    const int64_t z = blockIdx.z;
    g.out += z * g.s_out; hyp += z * g.s_hyp;
    if (dsc) dsc += z * g.s_hyp;
    if (g.noise) g.noise += z * g.s_noise;
    if (g.zero_info) g.zero_info += z * g.s_info;
    if (g.zero_flags) g.zero_flags += z * g.s_flags;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0) {
    if (g.zero_info && tid == 0) *g.zero_info = 0;
    if (g.zero_flags) for (int i = tid; i < g.n_zero_flags; i += 256) g.zero_flags[i] = 0u;
  }
  if (g.lower_only && g.col0 + j0 > i0 + GT - 1) return;

  const int cj = tid & 31;   // column pair: cols 2cj, 2cj+1
  const int ri = tid >> 5;   // row group: rows 8ri .. 8ri+7
  double acc[8][2];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r][0] = acc[r][1] = 0.0;

  const int lrow = tid >> 2;        // staging: 4 threads per row, 4 doubles each
  const int lk = (tid & 3) * 4;

  for (int dc0 = 0; dc0 < g.d; dc0 += DC) {
    if (tid < DC) {
      int k = dc0 + tid;
      len_s[tid] = k < g.d ? hyp[k] : 1.0;
      per_s[tid] = k < g.d ? hyp[g.d + k] : 1.0;
    }
    // stage (scaled for RBF/Matern: x / lengthscale as gpax/kernels.py:45-46; raw for periodic)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = dc0 + lk + q;
      double a = 0.0, b = 0.0;
      if (k < g.d) {
        const double len = hyp[k];
        if (i0 + lrow < g.n1) a = g.x1[(i0 + lrow) * g.d + k];
        if (j0 + lrow < g.n2) b = g.x2[(j0 + lrow) * g.d + k];
        if (KIND != GPX_PERIODIC) { a = a / len; b = b / len; }
      }
      x1s[lrow][lk + q] = a;
      x2s[lk + q][lrow] = b;
    }
    __syncthreads();
    const int kmax = (g.d - dc0) < DC ? (g.d - dc0) : DC;
    for (int k = 0; k < kmax; ++k) {
      const v2d b = *reinterpret_cast<const v2d*>(&x2s[k][2 * cj]);
      double pk = 0.0, lk_ = 0.0;
      if (KIND == GPX_PERIODIC) { pk = per_s[k]; lk_ = len_s[k]; }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const double a = x1s[ri * 8 + r][k];
        double d0 = a - b.x, d1 = a - b.y;
        if (KIND == GPX_PERIODIC) {
          // sin(pi * d / k_period) / k_length   (gpax/kernels.py:164)
          d0 = sin(M_PI * d0 / pk) / lk_;
          d1 = sin(M_PI * d1 / pk) / lk_;
        }
        acc[r][0] = fma(d0, d0, acc[r][0]);
        acc[r][1] = fma(d1, d1, acc[r][1]);
      }
    }
    __syncthreads();
  }

  // hyper-parameter scalars: from device memory when the launch is part of a replayed hipGraph
  // (one captured graph then serves every hyper-parameter value), else by value
  const double k_scale = dsc ? dsc[0] : g.k_scale;
  const double noise_scalar = dsc ? dsc[1] : g.noise_scalar;
  const double jitter = dsc ? dsc[2] : g.jitter;
  const int64_t col = j0 + 2 * cj;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int64_t row = i0 + ri * 8 + r;
    if (row >= g.n1) continue;
    double kv[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      double k = gram_value<KIND>(acc[r][c], k_scale);
      const int64_t grow = g.row_gid ? g.row_gid[row] : row;
      if (g.diag_rule && grow == g.col0 + col + c) {
        const double nz = g.noise ? g.noise[grow] : noise_scalar;
        k += (nz + jitter);                                 // kernels.py:124-125
      }
      kv[c] = k;
    }
    double* dst = g.out + row * g.ld + col;
    if (VEC2 && col + 1 < g.n2) {
      v2d v; v.x = kv[0]; v.y = kv[1];
      *reinterpret_cast<v2d*>(dst) = v;
    } else {
      if (col < g.n2) dst[0] = kv[0];
      if (col + 1 < g.n2) dst[1] = kv[1];
    }
  }
}

// ---- RBF / Matern on the matrix cores (round 3) ------------------------------------------------------------------------
// The reference forms the scaled squared distance as |a|^2 - 2 a.b + |b|^2, clipped at zero (gpax/kernels.py:39-51).  That
// form is ONE small matrix product: with the augmented rows  a' = [-2 a/l, |a/l|^2, 1]  and  b' = [b/l, 1, |b/l|^2]
// (d + 2 entries, zero-padded to a multiple of 4),  a' . b' = r2.  v_mfma_f64_16x16x4_f64 does it beside the vector ALU,
// which in gram_kernel above spends 2 d instructions per entry on the differences and, per tile, scales its two 64-row
// input blocks by fp64 divisions, stages them through LDS and meets at two barriers.  Measured (tools/gram_ablate.sh): with
// the exponentials AND the stores removed a tile kernel that stages its own inputs still took 19 of 26 us at N = 4096 and
// 159 of 250 us at N = 16384 -- the per-tile prologue, not the epilogue, was the larger half.  So:
//   gram_prep_kernel   one thread per input row, once per launch: x / l (true division, as the reference), the norm summed in
//                      ascending k, the augmented row written to a scratch array (a few hundred KB, L2-resident);
//   gram_mfma_kernel   one 64 x 64 output tile per 256-thread workgroup, NO LDS and NO barrier: every lane loads its MFMA
//                      operands straight from the augmented arrays (8 B per lane, 32-B row pieces), wave w owns rows
//                      16 w .. 16 w + 15 as four 16 x 16 MFMA tiles whose B rows are dealt so that a lane's tiles 0/1 (2/3)
//                      hold two ADJACENT columns:  tile t, lane column j  <->  output column 32 (t >> 1) + 2 j + (t & 1).
//                      A lane then stores 16 B at a time and a wave instruction writes four 256-B row segments.  Lower
//                      builds enumerate the lower triangle of tiles only (1-D grid).
// Differences from gram_kernel's direct form: <= 4e-15 absolute in r2 for |x / l|^2 <= 16 -- the reference's own
// arithmetic (tests/test_oracle.py pins both forms against each other; every K parity test holds its 1e-12).  The matrix
// core accumulates over K as a chain of fused multiply-adds in ascending k (build_small_kernel below restates the sum on
// the vector ALU and the two paths agree bit for bit, tests/test_gpu_round3.py).  Periodic kernels (a sine per dimension)
// and d > GRAM_MFMA_DMAX keep gram_kernel.
#define GRAM_MFMA_DMAX 62

// the epilogue of one 64 x 64 tile held in MFMA accumulators (wave w: rows 16 w + lk + 4 r; tile t, lane column li <-> column
// 32 (t >> 1) + 2 li + (t & 1)): clip, kernel function, diagonal rule, 16-B stores
template <int KIND, bool VEC2>
__device__ __forceinline__ void gram_tile_store(const GramArgs& g, const v4d (&acc)[4], const int64_t i0, const int64_t j0, const int w,
                                                const int li, const int lk, const double k_scale, const double noise_scalar,
                                                const double jitter, const int ablate) {
  // the diagonal rule can only touch a tile that meets the diagonal (row_gid: rows carry global ids, any tile may)
  const bool diag_tile = g.diag_rule && (g.row_gid || (i0 < g.col0 + j0 + GT && g.col0 + j0 < i0 + GT));
  if (VEC2 && !diag_tile && !ablate && i0 + GT <= g.n1 && j0 + GT <= g.n2) {
    // a full tile off the diagonal (all but ~2 N / 64 of them): no per-lane bounds or diagonal tests, one 32-bit lane offset
    // on a wave-uniform row base
    double* const base = g.out + (i0 + 16 * w) * g.ld + j0;
    const unsigned lo = (unsigned)(lk * (int)g.ld + 2 * li);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        v2d v;
        v.x = gram_value<KIND>(fmax(acc[2 * hf][r], 0.0), k_scale);
        v.y = gram_value<KIND>(fmax(acc[2 * hf + 1][r], 0.0), k_scale);
        *reinterpret_cast<v2d*>(base + (int64_t)(4 * r) * g.ld + 32 * hf + lo) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = i0 + 16 * w + lk + 4 * r;
    if (row >= g.n1) continue;
    double* const drow = g.out + row * g.ld;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int64_t col = j0 + 32 * hf + 2 * li;
      double kv[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) kv[c] = (ablate & 2) ? acc[2 * hf + c][r] : gram_value<KIND>(fmax(acc[2 * hf + c][r], 0.0), k_scale);   // clip (kernels.py:50)
      if ((ablate & 1) && kv[0] + kv[1] != 12345.678) continue;
      if (diag_tile) {
        const int64_t grow = g.row_gid ? g.row_gid[row] : row;
        const double add = (g.noise ? g.noise[grow] : noise_scalar) + jitter;                          // kernels.py:124-125
        if (grow == g.col0 + col) kv[0] += add;
        if (grow == g.col0 + col + 1) kv[1] += add;
      }
      if (VEC2 && col + 1 < g.n2) {
        v2d v; v.x = kv[0]; v.y = kv[1];
        *reinterpret_cast<v2d*>(drow + col) = v;
      } else {
        if (col < g.n2) drow[col] = kv[0];
        if (col + 1 < g.n2) drow[col + 1] = kv[1];
      }
    }
  }
}

// The same tile from the raw inputs, for launches of at most two rounds of workgroups (N <= ~5800 lower builds): no prep
// launch in front -- at these sizes its ~4 us (a kernel boundary and a latency-bound kernel) are more than the per-tile
// staging costs -- the workgroup scales its two 64-row blocks itself (the same divisions, the same ascending-k norm sums:
// the same bits as gram_prep_kernel), through LDS.
template <int KIND, bool VEC2>
__global__ __launch_bounds__(256) void gram_mfma_lds_kernel(GramArgs g, const int dp, const int64_t ntiles, const int tri, const int64_t ntj,
                                                            const int ablate) {
  extern __shared__ __attribute__((aligned(16))) double gram_lds[];   // xs1[64][dp], xs2[64][dp]
  double* const xs1 = gram_lds;
  double* const xs2 = gram_lds + GT * dp;
  const int tid = threadIdx.x;
  gpx_gptr hyp = (gpx_gptr)(uintptr_t)g.hyp, dsc = (gpx_gptr)(uintptr_t)g.dev_scalars;
  if (g.hyp_n) {
    hyp = gpx_kernarg_doubles(offsetof(GramArgs, hyp_in)); dsc = hyp + 2 * g.d;
    if (blockIdx.x == 0 && tid < g.hyp_n) g.hyp_out[tid] = hyp[tid];
  }
  if (blockIdx.z) {
    const int64_t z = blockIdx.z;
    g.out += z * g.s_out; hyp += z * g.s_hyp;
    if (dsc) dsc += z * g.s_hyp;
    if (g.noise) g.noise += z * g.s_noise;
    if (g.zero_info) g.zero_info += z * g.s_info;
    if (g.zero_flags) g.zero_flags += z * g.s_flags;
  }
  if (blockIdx.x == 0) {
    if (g.zero_info && tid == 0) *g.zero_info = 0;
    if (g.zero_flags) for (int i = tid; i < g.n_zero_flags; i += 256) g.zero_flags[i] = 0u;
  }
  const int64_t id = blockIdx.x;
  int64_t ti, tj;
  if (tri) {
    ti = (int64_t)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > id) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
    tj = id - ti * (ti + 1) / 2;
  } else {
    ti = id / ntj; tj = id - ti * ntj;
  }
  const int64_t i0 = ti * GT, j0 = tj * GT;
  if (g.lower_only && g.col0 + j0 > i0 + GT - 1) return;

  // stage: the 64 x d blocks of x1 and x2 are contiguous in memory -- element e of the block is row e / d, dimension e % d
  const int d = g.d, tot = GT * d;
  const int64_t lim1 = (g.n1 - i0) * d, lim2 = (g.n2 - j0) * d;   // elements of the block that exist
  const double* const b1 = g.x1 + i0 * d;
  const double* const b2 = g.x2 + j0 * d;
  for (int e = tid; e < tot; e += 256) {
    const int row = e / d, k = e - row * d;
    const double len = hyp[k];
    const double a = e < lim1 ? b1[e] : 0.0, b = e < lim2 ? b2[e] : 0.0;
    xs1[row * dp + k] = a / len;     // x / lengthscale (kernels.py:45-46)
    xs2[row * dp + k] = b / len;
  }
  __syncthreads();
  if (tid < 2 * GT) {
    double* const xr = (tid < GT ? xs1 : xs2) + (tid & (GT - 1)) * dp;
    double nrm = 0.0;
    for (int k = 0; k < d; ++k) nrm = fma(xr[k], xr[k], nrm);
    if (tid < GT) {
      for (int k = 0; k < d; ++k) xr[k] *= -2.0;
      xr[d] = nrm; xr[d + 1] = 1.0;
    } else {
      xr[d] = 1.0; xr[d + 1] = nrm;
    }
    for (int k = d + 2; k < dp; ++k) xr[k] = 0.0;
  }
  __syncthreads();

  const int lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
  const double* const ap = xs1 + (16 * w + li) * dp + lk;
  const double* const bp = xs2 + (2 * li) * dp + lk;
  v4d acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int s4 = 0; s4 < dp; s4 += 4) {
    const double a = ap[s4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double b = bp[(32 * (t >> 1) + (t & 1)) * dp + s4];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
    }
  }
  const double k_scale = dsc ? dsc[0] : g.k_scale;
  const double noise_scalar = dsc ? dsc[1] : g.noise_scalar;
  const double jitter = dsc ? dsc[2] : g.jitter;
  gram_tile_store<KIND, VEC2>(g, acc, i0, j0, w, li, lk, k_scale, noise_scalar, jitter, ablate);
}

// rows [0, n1p) of the grid: x1 -> aug1 (a' layout); rows [n1p, n1p + n2p): x2 -> aug2 (b' layout); rows beyond n are zero
__global__ __launch_bounds__(256) void gram_prep_kernel(GramArgs g, const int dp, const int64_t n1p, const int64_t n2p, double* aug1,
                                                        double* aug2, const int64_t s_aug) {
  gpx_gptr hyp = (gpx_gptr)(uintptr_t)g.hyp;
  if (g.hyp_n) {
    hyp = gpx_kernarg_doubles(offsetof(GramArgs, hyp_in));
    if (blockIdx.x == 0 && (int)threadIdx.x < g.hyp_n) g.hyp_out[threadIdx.x] = hyp[threadIdx.x];
  }
  if (blockIdx.z) { hyp += (int64_t)blockIdx.z * g.s_hyp; aug1 += (int64_t)blockIdx.z * s_aug; aug2 += (int64_t)blockIdx.z * s_aug; }
  int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n1p + n2p) return;
  const bool second = row >= n1p;
  if (second) row -= n1p;
  const int64_t n = second ? g.n2 : g.n1;
  const double* const x = (second ? g.x2 : g.x1) + row * g.d;
  double* const o = (second ? aug2 : aug1) + row * dp;
  const int d = g.d;
  if (row >= n) { for (int k = 0; k < dp; ++k) o[k] = 0.0; return; }
  double nrm = 0.0;
  int k0 = 0;
  for (; k0 + 8 <= d; k0 += 8) {   // eight dimensions at a time: their loads are in flight together
    double xv[8], lv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { xv[q] = x[k0 + q]; lv[q] = hyp[k0 + q]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const double a = xv[q] / lv[q];     // x / lengthscale (kernels.py:45-46)
      nrm = fma(a, a, nrm);
      o[k0 + q] = second ? a : -2.0 * a;
    }
  }
  for (int k = k0; k < d; ++k) {
    const double a = x[k] / hyp[k];
    nrm = fma(a, a, nrm);
    o[k] = second ? a : -2.0 * a;
  }
  o[d] = second ? 1.0 : nrm;
  o[d + 1] = second ? nrm : 1.0;
  for (int k = d + 2; k < dp; ++k) o[k] = 0.0;
}

template <int KIND, bool VEC2, int DP>   // DP: compile-time padded inner dimension (12, 20), or 0 = the run-time dp
__global__ __launch_bounds__(256) void gram_mfma_kernel(GramArgs g, const int dp_rt, const int64_t ntiles, const int tri, const int64_t ntj,
                                                        const double* aug1, const double* aug2, const int64_t s_aug, const int ablate) {
  const int dp = DP ? DP : dp_rt;
  const int tid = threadIdx.x;
  gpx_gptr dsc = (gpx_gptr)(uintptr_t)g.dev_scalars;
  if (blockIdx.z) {
    const int64_t z = blockIdx.z;
    g.out += z * g.s_out;
    if (dsc) dsc += z * g.s_hyp;
    if (g.noise) g.noise += z * g.s_noise;
    if (g.zero_info) g.zero_info += z * g.s_info;
    if (g.zero_flags) g.zero_flags += z * g.s_flags;
    aug1 += z * s_aug; aug2 += z * s_aug;
  }
  if (blockIdx.x == 0) {
    if (g.zero_info && tid == 0) *g.zero_info = 0;
    if (g.zero_flags) for (int i = tid; i < g.n_zero_flags; i += 256) g.zero_flags[i] = 0u;
  }
  const double k_scale = dsc ? dsc[0] : g.k_scale;
  const double noise_scalar = dsc ? dsc[1] : g.noise_scalar;
  const double jitter = dsc ? dsc[2] : g.jitter;
  const int lane = tid & 63, w = tid >> 6, li = lane & 15, lk = lane >> 4;
  const int la = (16 * w + li) * dp + lk, lb = (2 * li) * dp + lk;   // lane offsets inside the tile's 64 augmented rows

  for (int64_t id = blockIdx.x; id < ntiles; id += gridDim.x) {
    int64_t ti, tj;
    if (tri) {   // row-major over the lower triangle of tiles
      ti = (int64_t)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
      while (ti * (ti + 1) / 2 > id) --ti;
      while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
      tj = id - ti * (ti + 1) / 2;
    } else {
      ti = id / ntj; tj = id - ti * ntj;
    }
    const int64_t i0 = ti * GT, j0 = tj * GT;
    if (g.lower_only && g.col0 + j0 > i0 + GT - 1) continue;   // (wave-uniform)

    const double* const ap = aug1 + i0 * dp + la;
    const double* const bp = aug2 + j0 * dp + lb;
    v4d acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    auto kstep = [&](const int s4) __attribute__((always_inline)) {
      const double a = ap[s4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double b = bp[(32 * (t >> 1) + (t & 1)) * dp + s4];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    };
    if constexpr (DP > 0) {
#pragma unroll
      for (int s4 = 0; s4 < DP; s4 += 4) kstep(s4);   // every operand load of the tile in flight before the first MFMA
    } else {
      for (int s4 = 0; s4 < dp; s4 += 4) kstep(s4);
    }

    gram_tile_store<KIND, VEC2>(g, acc, i0, j0, w, li, lk, k_scale, noise_scalar, jitter, ablate);
  }
}

// (A slab-shaped variant -- 16 rows x 512 columns per workgroup, one contiguous 1-KiB row segment per wave instruction, the
// X2 block staged once per 16 rows, the shape that writes a plain 2-GiB stream at 5.7 instead of 4.7 TB/s -- was written in
// round 3, returned the same bits and ran 1.6-2.3x SLOWER at every size (N = 4096: 46.8 vs 30.0 us, N = 16384: 537 vs 308 us,
// profiles/r03_gram_slab_vs_tiles.txt): the build is bound by its fp64 VALU work, not by the shape of its stores -- ~75 VALU
// instructions per entry, 45 of them libdevice's exp -- and 32 entries per thread at four waves per SIMD hide less of it
// than 16 at eight.  What helped instead is gpx_exp_neg (gpx_internal.h).  Removed.)

int launch_gram(gpx_handle* h, hipStream_t s, const GramLaunch& L) {
  if (L.n1 <= 0 || L.n2 <= 0) return GPX_OK;
  GramArgs g;
  g.x1 = L.x1; g.x2 = L.x2; g.hyp = L.hyp; g.dev_scalars = L.dev_scalars; g.noise = L.noise; g.out = L.out;
  g.row_gid = L.row_gid;
  g.n1 = L.n1; g.n2 = L.n2; g.ld = L.ld; g.d = L.d;
  g.lower_only = L.lower_only; g.diag_rule = L.diag_rule;
  g.k_scale = L.k_scale; g.noise_scalar = L.noise_scalar; g.jitter = L.jitter;
  g.col0 = L.col0; g.zero_info = L.zero_info; g.zero_flags = L.zero_flags; g.n_zero_flags = L.n_zero_flags;
  g.s_out = L.s_out; g.s_hyp = L.s_hyp; g.s_noise = L.s_noise; g.s_info = L.s_info; g.s_flags = L.s_flags;
  g.hyp_n = 0; g.hyp_out = nullptr;
  if (L.hyp_inline && L.hyp_inline_n > 0) {
    if (L.batch != 1 || L.hyp_inline_n > GPX_HYP_INLINE || L.hyp_inline_n < 2 * L.d + 3) { gpx_set_error(h, "gram: inline hyper-parameters out of range"); return GPX_ERR_ARG; }
    g.hyp_n = L.hyp_inline_n; g.hyp_out = const_cast<double*>(L.hyp);
    for (int i = 0; i < L.hyp_inline_n; ++i) g.hyp_in[i] = L.hyp_inline[i];
  }
  if (L.batch < 1 || L.batch > 65535) { gpx_set_error(h, "gram: batch out of range"); return GPX_ERR_ARG; }
  dim3 grid((unsigned)((L.n2 + GT - 1) / GT), (unsigned)((L.n1 + GT - 1) / GT), (unsigned)L.batch);
  const bool vec2 = (L.ld % 2 == 0) && ((reinterpret_cast<uintptr_t>(L.out) & 15) == 0);
#define GPX_GRAM_LAUNCH(KIND)                                                        \
  do {                                                                               \
    if (vec2) hipExtLaunchKernelGGL((gram_kernel<KIND, true>), grid, dim3(256), 0, s, L.start_event, L.stop_event, 0, g);  \
    else hipExtLaunchKernelGGL((gram_kernel<KIND, false>), grid, dim3(256), 0, s, L.start_event, L.stop_event, 0, g);      \
  } while (0)
  const int dp = (L.d + 2 + 3) / 4 * 4;
  // tiles of the MFMA kernel: the lower triangle only where the launch builds the lower part of a square matrix
  const int64_t nti = (L.n1 + GT - 1) / GT, ntj = (L.n2 + GT - 1) / GT;
  const int tri = (L.lower_only && L.col0 == 0 && L.n1 == L.n2 && !L.row_gid) ? 1 : 0;
  const int64_t ntiles = tri ? nti * (nti + 1) / 2 : nti * ntj;
  static const int tpw_env = getenv("GPX_GRAM_TPW") ? atoi(getenv("GPX_GRAM_TPW")) : 0;   // diagnostic: tiles per workgroup
  static const int ablate_env = getenv("GPX_GRAM_ABLATE") ? atoi(getenv("GPX_GRAM_ABLATE")) : 0;   // diagnostic: 1 = no stores, 2 = no exp, 4 / 8 = no fast epilogue
  // the branch-free epilogue of full off-diagonal tiles issues its stores 10 % faster -- which helps up to N = 16384 (N = 4096:
  // 24.0 -> 21.5 us, N = 16384: 235 -> 228 us, same box) and HURTS a multi-gigabyte write stream (N = 32768, 4.3 GB: 940-980 ->
  // 1070-1130 us on three boxes): above 2 GiB of output the tiles keep the branchy epilogue (bit 8)
  const int ablate = ablate_env | ((double)ntiles * GT * GT * 8.0 * L.batch > 2147483648.0 ? 8 : 0);
  const int64_t tpw = tpw_env > 0 ? tpw_env : 1;
  const dim3 mgrid((unsigned)((ntiles + tpw - 1) / tpw), 1u, (unsigned)L.batch);
  const int64_t n1p = nti * GT, n2p = ntj * GT, s_aug = (n1p + n2p) * dp;
  double* aug1 = nullptr; double* aug2 = nullptr;
  static const bool direct_env = getenv("GPX_GRAM_DIRECT") != nullptr;   // diagnostic: time gram_kernel (tools/gram_ab.sh)
  const bool mfma = L.d <= GRAM_MFMA_DMAX && !direct_env && L.kind != GPX_PERIODIC;
  static const int lds_env = getenv("GPX_GRAM_LDS") ? atoi(getenv("GPX_GRAM_LDS")) : -1;   // diagnostic: 1 / 0 force the variant
  const bool lds_variant = lds_env >= 0 ? lds_env != 0 : ntiles * L.batch <= 16 * (int64_t)h->prop.multiProcessorCount;
  const size_t lds = (size_t)(2 * GT * dp) * sizeof(double);
  const dim3 lgrid((unsigned)ntiles, 1u, (unsigned)L.batch);
  if (mfma && !lds_variant) {
    // the augmented rows of this launch: one scratch array per stream of the handle (launches of one stream run one after
    // the other; the step's K build on the chain stream and its cross gram on the panel stream may overlap)
    const int slot = s == h->s_main ? 0 : (s == h->s_far ? 1 : (s == h->s_trail ? 2 : 3));
    GPX_RET(ensure(h, &h->d_aug[slot], &h->aug_cap[slot], s_aug * L.batch));
    aug1 = h->d_aug[slot]; aug2 = aug1 + n1p * dp;
    GramArgs gp = g;   // the prep launch is the step's first launch: it carries the inline hyper-parameters
    g.hyp_n = 0; g.hyp_out = nullptr;
    const dim3 pgrid((unsigned)((n1p + n2p + 255) / 256), 1u, (unsigned)L.batch);
    hipExtLaunchKernelGGL(gram_prep_kernel, pgrid, dim3(256), 0, s, L.start_event, nullptr, 0, gp, dp, n1p, n2p, aug1, aug2, s_aug);
  }
#define GPX_GRAM_MFMA(KIND)                                                          \
  do {                                                                               \
    if (vec2 && dp == 12) hipExtLaunchKernelGGL((gram_mfma_kernel<KIND, true, 12>), mgrid, dim3(256), 0, s, nullptr, L.stop_event, 0, g, dp, ntiles, tri, ntj, aug1, aug2, s_aug, ablate);  \
    else if (vec2 && dp == 20) hipExtLaunchKernelGGL((gram_mfma_kernel<KIND, true, 20>), mgrid, dim3(256), 0, s, nullptr, L.stop_event, 0, g, dp, ntiles, tri, ntj, aug1, aug2, s_aug, ablate);  \
    else if (vec2) hipExtLaunchKernelGGL((gram_mfma_kernel<KIND, true, 0>), mgrid, dim3(256), 0, s, nullptr, L.stop_event, 0, g, dp, ntiles, tri, ntj, aug1, aug2, s_aug, ablate);  \
    else hipExtLaunchKernelGGL((gram_mfma_kernel<KIND, false, 0>), mgrid, dim3(256), 0, s, nullptr, L.stop_event, 0, g, dp, ntiles, tri, ntj, aug1, aug2, s_aug, ablate);      \
  } while (0)
#define GPX_GRAM_MFMA_LDS(KIND)                                                      \
  do {                                                                               \
    if (vec2) hipExtLaunchKernelGGL((gram_mfma_lds_kernel<KIND, true>), lgrid, dim3(256), lds, s, L.start_event, L.stop_event, 0, g, dp, ntiles, tri, ntj, ablate);  \
    else hipExtLaunchKernelGGL((gram_mfma_lds_kernel<KIND, false>), lgrid, dim3(256), lds, s, L.start_event, L.stop_event, 0, g, dp, ntiles, tri, ntj, ablate);      \
  } while (0)
  switch (L.kind) {
    case GPX_RBF: if (mfma && lds_variant) GPX_GRAM_MFMA_LDS(GPX_RBF); else if (mfma) GPX_GRAM_MFMA(GPX_RBF); else GPX_GRAM_LAUNCH(GPX_RBF); break;
    case GPX_MATERN52: if (mfma && lds_variant) GPX_GRAM_MFMA_LDS(GPX_MATERN52); else if (mfma) GPX_GRAM_MFMA(GPX_MATERN52); else GPX_GRAM_LAUNCH(GPX_MATERN52); break;
    case GPX_PERIODIC: GPX_GRAM_LAUNCH(GPX_PERIODIC); break;
    default: gpx_set_error(h, "unknown kernel kind"); return GPX_ERR_ARG;
  }
#undef GPX_GRAM_LAUNCH
#undef GPX_GRAM_MFMA
#undef GPX_GRAM_MFMA_LDS
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

// ---- one launch for the whole bordered matrix of a one- or two-block-column problem (npad <= 256) -------------
// [K ; identity padding ; k_cross ; r ; zero rows], npad columns: what gram + 2 fills + cross gram + border rows build
// in five launches.  At these sizes (the reference's own examples: 25-256 points) a step is bound by launch latency,
// ~8-10 us per kernel in a stream, not by arithmetic.  One thread per entry, inputs straight from L2; the same
// arithmetic in the same order as gram_kernel (scaled coordinates, differences, fma accumulation, same epilogues), so
// the entries carry the same bits.
struct BuildArgs {
  const double* x; const double* xnew; const double* hyp; const double* dev_scalars; const double* noise; const double* r;
  double* out; int64_t ld;
  int64_t n, npad, m_rows, r_row, rows_total;
  int d, diag_rule;
  int* zero_info; unsigned* zero_flags; int n_zero_flags;
  double* r_copy;       // when set: the r row is also written here (r may then live in pinned host memory: no upload in the stream)
  int64_t ident_row0;   // >= 0: rows [ident_row0, ident_row0 + n) are identity rows (the gradient path's border: they become U = L^-T)
  double* zero2;        // when set: a second matrix (same ld) whose rows >= npad are cleared by the same launch (the border of L)
  int64_t s_out, s_hyp, s_noise, s_r, s_info;   // per-draw strides of a batched launch (blockIdx.y = draw)
  int hyp_n; double hyp_in[GPX_HYP_INLINE];     // as GramArgs: hyper-parameters in the kernel arguments, written to `hyp` by workgroup 0
};

template <int KIND>
__global__ void build_small_kernel(BuildArgs g) {
  const int64_t row = blockIdx.x, col = threadIdx.x;
  gpx_gptr hyp = (gpx_gptr)(uintptr_t)g.hyp, dsc = (gpx_gptr)(uintptr_t)g.dev_scalars;
  if (g.hyp_n) {
    hyp = gpx_kernarg_doubles(offsetof(BuildArgs, hyp_in)); dsc = hyp + 2 * g.d;
    if (blockIdx.x == 0 && (int)threadIdx.x < g.hyp_n) const_cast<double*>(g.hyp)[threadIdx.x] = hyp[threadIdx.x];
  }
  if (blockIdx.y) {
    const int64_t z = blockIdx.y;
    g.out += z * g.s_out; hyp += z * g.s_hyp; dsc += z * g.s_hyp;
    if (g.noise) g.noise += z * g.s_noise;
    if (g.r) g.r += z * g.s_r;
    if (g.r_copy) g.r_copy += z * g.s_r;
    if (g.zero2) g.zero2 += z * g.s_out;
    if (g.zero_info) g.zero_info += z * g.s_info;
  }
  if (blockIdx.x == 0) {
    if (g.zero_info && threadIdx.x == 0) *g.zero_info = 0;
    if (g.zero_flags) for (int i = threadIdx.x; i < g.n_zero_flags; i += blockDim.x) g.zero_flags[i] = 0u;
  }
  if (col >= g.npad) return;
  double* dst = g.out + row * g.ld + col;
  if (g.zero2 && row >= g.npad) g.zero2[(row - g.npad) * g.ld + col] = 0.0;
  const double* x1 = nullptr;
  bool diag_row = false;
  if (row < g.n) { x1 = g.x + row * g.d; diag_row = g.diag_rule != 0; }
  else if (row < g.npad) { *dst = (row == col) ? 1.0 : 0.0; return; }
  else if (row < g.npad + g.m_rows) x1 = g.xnew + (row - g.npad) * g.d;
  else if (row == g.r_row) {
    const double v = (col < g.n && g.r) ? g.r[col] : 0.0;
    *dst = v;
    if (g.r_copy && col < g.n) g.r_copy[col] = v;
    return;
  }
  else if (g.ident_row0 >= 0 && row >= g.ident_row0 && row < g.ident_row0 + g.n) { *dst = (col == row - g.ident_row0) ? 1.0 : 0.0; return; }
  else { *dst = 0.0; return; }
  if (col >= g.n) { *dst = 0.0; return; }
  const double* x2 = g.x + col * g.d;
  double acc = 0.0;
  if (KIND != GPX_PERIODIC && g.d <= GRAM_MFMA_DMAX) {
    // gram_mfma_kernel's arithmetic, operation for operation: scaled inputs, their norms summed in ascending k, then the
    // augmented dot product  sum_k (-2 a_k) b_k + |a|^2 * 1 + 1 * |b|^2  as the chain of fused multiply-adds the matrix
    // core runs over K (ascending k from a zero accumulator), clipped at zero
    double na = 0.0, nb_ = 0.0;
    for (int k = 0; k < g.d; ++k) {
      const double len = hyp[k];
      const double a = x1[k] / len, b = x2[k] / len;
      na = fma(a, a, na); nb_ = fma(b, b, nb_);
      acc = fma(-2.0 * a, b, acc);
    }
    acc = fma(na, 1.0, acc);
    acc = fma(1.0, nb_, acc);
    acc = fmax(acc, 0.0);
  } else {
    for (int k = 0; k < g.d; ++k) {
      const double len = hyp[k];
      double a = x1[k], b = x2[k];
      if (KIND != GPX_PERIODIC) { a = a / len; b = b / len; }
      double d0 = a - b;
      if (KIND == GPX_PERIODIC) d0 = sin(M_PI * d0 / hyp[g.d + k]) / len;
      acc = fma(d0, d0, acc);
    }
  }
  const double k_scale = dsc[0], noise_scalar = dsc[1], jitter = dsc[2];
  double kv = gram_value<KIND>(acc, k_scale);
  if (diag_row && row == col) kv += ((g.noise ? g.noise[row] : noise_scalar) + jitter);
  *dst = kv;
}

int launch_build_small(gpx_handle* h, hipStream_t s, int kind, const double* x, const double* xnew, const double* hyp,
                       const double* dev_scalars, const double* noise, const double* r, double* out, int64_t ld, int64_t n,
                       int64_t npad, int64_t m_rows, int64_t r_row, int64_t rows_total, int d, int diag_rule, int* zero_info,
                       unsigned* zero_flags, int n_zero_flags, int64_t ident_row0, double* zero2, double* r_copy, int batch,
                       int64_t s_out, int64_t s_hyp, int64_t s_noise, int64_t s_r, int64_t s_info, const double* hyp_inline,
                       int hyp_inline_n) {
  if (npad > 256 || npad <= 0 || !dev_scalars) { gpx_set_error(h, "build_small: at most 256 columns"); return GPX_ERR_ARG; }
  BuildArgs g;
  g.x = x; g.xnew = xnew; g.hyp = hyp; g.dev_scalars = dev_scalars; g.noise = noise; g.r = r; g.out = out; g.ld = ld;
  g.n = n; g.npad = npad; g.m_rows = m_rows; g.r_row = r_row; g.rows_total = rows_total; g.d = d; g.diag_rule = diag_rule;
  g.zero_info = zero_info; g.zero_flags = zero_flags; g.n_zero_flags = n_zero_flags;
  g.ident_row0 = ident_row0; g.zero2 = zero2; g.r_copy = r_copy;
  g.s_out = s_out; g.s_hyp = s_hyp; g.s_noise = s_noise; g.s_r = s_r; g.s_info = s_info;
  g.hyp_n = 0;
  if (hyp_inline && hyp_inline_n > 0) {
    if (batch != 1 || hyp_inline_n > GPX_HYP_INLINE || hyp_inline_n < 2 * d + 3) { gpx_set_error(h, "build_small: inline hyper-parameters out of range"); return GPX_ERR_ARG; }
    g.hyp_n = hyp_inline_n;
    for (int i = 0; i < hyp_inline_n; ++i) g.hyp_in[i] = hyp_inline[i];
  }
  if (batch > 1 && zero_flags) { gpx_set_error(h, "build_small: zero_flags is not batched"); return GPX_ERR_ARG; }
  const dim3 grid((unsigned)rows_total, (unsigned)batch), block((unsigned)(npad <= 64 ? 64 : (npad <= 128 ? 128 : 256)));
  switch (kind) {
    case GPX_RBF: hipLaunchKernelGGL((build_small_kernel<GPX_RBF>), grid, block, 0, s, g); break;
    case GPX_MATERN52: hipLaunchKernelGGL((build_small_kernel<GPX_MATERN52>), grid, block, 0, s, g); break;
    case GPX_PERIODIC: hipLaunchKernelGGL((build_small_kernel<GPX_PERIODIC>), grid, block, 0, s, g); break;
    default: gpx_set_error(h, "unknown kernel kind"); return GPX_ERR_ARG;
  }
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

// ------------------------------------------------------------------- fill helpers

// out[r][c] = (r == c) ? diag : 0   for r in [r0,r1), c in [c0,c1)
__global__ void fill_rect_kernel(double* A, int64_t ld, int64_t r0, int64_t r1, int64_t c0,
                                 int64_t c1, double diag, int64_t sA) {
  const int64_t c = c0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t r = r0 + blockIdx.y;
  if (c < c1 && r < r1) A[(int64_t)blockIdx.z * sA + r * ld + c] = (r == c) ? diag : 0.0;
}

int launch_fill_rect(gpx_handle* h, hipStream_t s, double* A, int64_t ld, int64_t r0, int64_t r1,
                     int64_t c0, int64_t c1, double diag, int batch, int64_t sA) {
  if (r1 <= r0 || c1 <= c0) return GPX_OK;
  dim3 grid((unsigned)((c1 - c0 + 255) / 256), (unsigned)(r1 - r0), (unsigned)batch);
  hipLaunchKernelGGL(fill_rect_kernel, grid, dim3(256), 0, s, A, ld, r0, r1, c0, c1, diag, sA);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

// Padding of the (npad x npad) matrix to identity outside the valid (n x n) part:
// chol([[K,0],[0,I]]) = [[L,0],[0,I]] exactly, so padding never perturbs L.
int launch_fill_identity_pad(gpx_handle* h, hipStream_t s, double* A, int64_t ld, int64_t n,
                             int64_t npad, int64_t rows_total, int batch, int64_t sA) {
  GPX_RET(launch_fill_rect(h, s, A, ld, 0, rows_total, n, npad, 1.0, batch, sA));  // columns [n,npad) of every row
  GPX_RET(launch_fill_rect(h, s, A, ld, n, npad, 0, n, 1.0, batch, sA));           // rows [n,npad), columns [0,n)
  return GPX_OK;
}

// dst[0:n) = src[0:n) (or 0 when src == nullptr), dst[n:npad) = 0
__global__ void set_row_kernel(double* dst, const double* src, int64_t n, int64_t npad) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < npad) dst[c] = (src && c < n) ? src[c] : 0.0;
}

int launch_set_row(gpx_handle* h, hipStream_t s, double* dst, const double* src, int64_t n,
                   int64_t npad) {
  if (npad <= 0) return GPX_OK;
  hipLaunchKernelGGL(set_row_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, s, dst, src,
                     n, npad);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

// row r_row <- r, rows [z0, z1) <- 0 (columns [0, n) each), *info <- 0: the per-step border bookkeeping in one launch
__global__ void border_rows_kernel(double* A, int64_t ld, int64_t npad, int64_t n, int64_t r_row, const double* r,
                                   int64_t z0, int64_t z1, int* info, int64_t sA, int64_t s_r) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  A += (int64_t)blockIdx.z * sA;
  if (r) r += (int64_t)blockIdx.z * s_r;
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && info) *info = 0;
  if (c >= n) return;   // columns [n, npad) of every row belong to launch_fill_identity_pad
  if (blockIdx.y == 0) {
    if (r_row >= 0) A[r_row * ld + c] = r ? r[c] : 0.0;
    return;
  }
  const int64_t row = z0 + (int64_t)blockIdx.y - 1;
  if (row < z1) A[row * ld + c] = 0.0;
}

int launch_border_rows(gpx_handle* h, hipStream_t s, double* A, int64_t ld, int64_t npad, int64_t n, int64_t r_row,
                       const double* r, int64_t z0, int64_t z1, int* info, int batch, int64_t sA, int64_t s_r) {
  const int64_t nz = z1 > z0 ? z1 - z0 : 0;
  hipLaunchKernelGGL(border_rows_kernel, dim3((unsigned)((npad + 255) / 256), (unsigned)(nz + 1), (unsigned)batch), dim3(256), 0, s, A, ld,
                     npad, n, r_row, r, z0, z1, info, sA, s_r);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}
