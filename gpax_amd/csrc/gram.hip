// gram.hip -- tiled pairwise-distance / gram-matrix builder for gfx950.
//
// Replaces, in one fused kernel, the JAX-primitive sequence of
//   _squared_distance            gpax/kernels.py:39-51
//   RBFKernel._kernel_function   gpax/kernels.py:112-126
//   MaternKernel (nu=5/2)        gpax/kernels.py:131-147   (r = sqrt(r2 + 1e-12))
//   PeriodicKernel               gpax/kernels.py:152-168   (direct differences, sin)
// including the diagonal rule `(k_noise + k_jitter) * eye(n)`.
//
// HBM-bound: the output is written exactly once (8 B/entry, 16 B per lane per store,
// two full 512-B row segments per wave-instruction) and each 64-row X block is read
// once per tile into LDS (coalesced 32-B pieces), so HBM traffic is the algorithmic
// 8*n1*n2 + 8*(n1+n2)*d bytes (the X re-reads per tile are L2 hits).
// Distances use direct differences (sum_k (a_k-b_k)^2): free of the cancellation of
// the |a|^2-2ab+|b|^2 expansion, never negative (so no clip needed), and within
// 1e-15 of the reference's expansion form (tests/test_oracle.py).
// fp64 MFMA is NOT used here on purpose: on gfx950 v_mfma_f64 runs at the vector
// fp64 rate, the inner dimension d is tiny, and the kernel is bound by HBM writes
// and the fp64 exp, not by the distance arithmetic.

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
      const double r2 = acc[r][c];
      double k;
      if (KIND == GPX_RBF) {
        k = k_scale * gpx_exp_neg(-0.5 * r2);               // kernels.py:123
      } else if (KIND == GPX_MATERN52) {
        const double rr = sqrt(r2 + 1e-12);                   // kernels.py:142
        const double s5r = 2.2360679774997896964 * rr;             // 5**0.5 * r
        // 1 + sqrt5 r + 5/3 r2 with the contractions hipcc chose for the plain expression in rounds 1-2, spelled out
        // (the entries keep their bits whatever else in this function changes)
        const double poly = __builtin_fma(5.0 / 3.0, r2, __builtin_fma(2.2360679774997896964, rr, 1.0));
        k = k_scale * poly * gpx_exp_neg(-s5r);                    // kernels.py:144
      } else {
        k = k_scale * gpx_exp_neg(-2.0 * r2);               // kernels.py:165
      }
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
  switch (L.kind) {
    case GPX_RBF: GPX_GRAM_LAUNCH(GPX_RBF); break;
    case GPX_MATERN52: GPX_GRAM_LAUNCH(GPX_MATERN52); break;
    case GPX_PERIODIC: GPX_GRAM_LAUNCH(GPX_PERIODIC); break;
    default: gpx_set_error(h, "unknown kernel kind"); return GPX_ERR_ARG;
  }
#undef GPX_GRAM_LAUNCH
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
  for (int k = 0; k < g.d; ++k) {
    const double len = hyp[k];
    double a = x1[k], b = x2[k];
    if (KIND != GPX_PERIODIC) { a = a / len; b = b / len; }
    double d0 = a - b;
    if (KIND == GPX_PERIODIC) d0 = sin(M_PI * d0 / hyp[g.d + k]) / len;
    acc = fma(d0, d0, acc);
  }
  const double k_scale = dsc[0], noise_scalar = dsc[1], jitter = dsc[2];
  double kv;
  if (KIND == GPX_RBF) kv = k_scale * gpx_exp_neg(-0.5 * acc);
  else if (KIND == GPX_MATERN52) {
    const double rr = sqrt(acc + 1e-12);
    const double s5r = 2.2360679774997896964 * rr;
    const double poly = __builtin_fma(5.0 / 3.0, acc, __builtin_fma(2.2360679774997896964, rr, 1.0));   // as in gram_kernel
    kv = k_scale * poly * gpx_exp_neg(-s5r);
  } else kv = k_scale * gpx_exp_neg(-2.0 * acc);
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
