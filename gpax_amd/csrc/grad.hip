// grad.hip -- gradient of the log-marginal likelihood with respect to the kernel hyper-parameters.
//
// In the reference this is jax.grad of the NumPyro potential of `_gp_prior` (gpax/gp.py:133-151),
// evaluated by NUTS at every leapfrog step (gpax/gp.py:515-530) and by SVI at every optimiser step
// (gpax/gp.py:583-587): reverse-mode autodiff through the gram build (gpax/kernels.py:112-168), the
// Cholesky and the MVN log-density.  Here it is the closed form
//     d logp / d theta = 1/2 sum_ij G_ij dK_ij/dtheta,    G = alpha alpha^T - K^-1,  alpha = K^-1 r,
// with K^-1 = U U^T from U = L^-T (api.hip: the identity rides through the factorisation as border
// rows) and dK/dtheta RECOMPUTED from x inside the reduction, so no dK matrix ever exists in HBM:
// one pass over the lower triangle of K^-1 (8 B per entry) yields the sums for every parameter.
//
// With t_d the scaled difference of dimension d ((a-b)/l_d, or sin(pi (a-b)/p_d)/l_d for the periodic
// kernel), r2 = sum_d t_d^2, k = s*e0(r2) and q(r2) = -2/s dk/d(r2):
//     dk/ds = e0,   dk/dl_d = s q t_d^2 / l_d,   dk/dp_d = s q u_d sin(2 u_d) / (2 p_d l_d^2), u_d = pi (a-b)/p_d,
//     RBF: e0 = q = exp(-r2/2);   Matern-5/2: e0 = (1 + sqrt5 r + 5/3 r2) e^{-sqrt5 r}, q = 5/3 (1 + sqrt5 r2/r) e^{-sqrt5 r},
//     r = sqrt(r2 + 1e-12) (gpax/kernels.py:142);   periodic: e0 = exp(-2 r2), q = 4 e0;
//     d(diagonal rule)/d k_noise = identity.
// HBM-bound on the K^-1 read; the work per entry is the gram kernel's plus 3 flops per parameter.

#include "gpx_internal.h"

#define GT 64
#define DC 16
#define NPV (4 + 2 * DC)   // per-workgroup partial sums: S, N, L_iso, P_iso, L[DC], P[DC]

struct GradArgs {
  const double* x;        // (n, d)
  const double* hyp;      // length[d], period[d]
  const double* kinv;     // lower tiles valid, leading dimension ldk
  const double* alpha;    // (n)
  double* part;           // [gridDim.y][gridDim.x][NPV]
  double* mirror;         // single workgroup per chunk (part IS the sums): the pinned host mirror of the sums, or NULL
  int64_t n, ldk, ntiles;
  int d, per_dim;
  // batched hyper-parameter draws (blockIdx.z = draw): per-draw strides in doubles; x is shared
  int64_t s_hyp, s_kinv, s_alpha, s_part;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int KIND>
__global__ __launch_bounds__(256) void grad_reduce_kernel(GradArgs g) {
  __shared__ double x1s[GT][DC];
  __shared__ __attribute__((aligned(16))) double x2s[DC][GT];
  __shared__ double len_s[DC], per_s[DC];
  __shared__ double red[4][NPV];

  const int tid = threadIdx.x;
  const int cj = tid & 31, ri = tid >> 5;
  const int lrow = tid >> 2, lk = (tid & 3) * 4;
  const int chunk0 = blockIdx.y * DC;   // the dimension chunk whose per-dimension sums this workgroup owns
  if (blockIdx.z) {
    const int64_t z = blockIdx.z;
    g.hyp += z * g.s_hyp; g.kinv += z * g.s_kinv; g.alpha += z * g.s_alpha; g.part += z * g.s_part;
  }

  double accS = 0.0, accN = 0.0, accLi = 0.0, accPi = 0.0;
  double accL[DC], accP[DC];
#pragma unroll
  for (int k = 0; k < DC; ++k) accL[k] = accP[k] = 0.0;

  auto stage = [&](int64_t i0, int64_t j0, int dc0, bool scaled) {
    if (tid < DC) {
      const int k = dc0 + tid;
      len_s[tid] = k < g.d ? g.hyp[k] : 1.0;
      per_s[tid] = k < g.d ? g.hyp[g.d + k] : 1.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = dc0 + lk + q;
      double a = 0.0, b = 0.0;
      if (k < g.d) {
        if (i0 + lrow < g.n) a = g.x[(i0 + lrow) * g.d + k];
        if (j0 + lrow < g.n) b = g.x[(j0 + lrow) * g.d + k];
        if (scaled) { const double len = g.hyp[k]; a = a / len; b = b / len; }
      }
      x1s[lrow][lk + q] = a;
      x2s[lk + q][lrow] = b;
    }
  };

  for (int64_t t = blockIdx.x; t < g.ntiles; t += gridDim.x) {
    // lower-triangle tile (ti >= tj) from the linear index
    int64_t ti = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > t) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int64_t tj = t - ti * (ti + 1) / 2;
    const int64_t i0 = ti * GT, j0 = tj * GT;

    // ---- pass A: r2 (and, periodic, psum = sum_d u sin(2u) / l_d^2) over all dimensions
    double r2[8][2], ps[8][2];
#pragma unroll
    for (int r = 0; r < 8; ++r) r2[r][0] = r2[r][1] = ps[r][0] = ps[r][1] = 0.0;
    for (int dc0 = 0; dc0 < g.d; dc0 += DC) {
      __syncthreads();
      stage(i0, j0, dc0, KIND != GPX_PERIODIC);
      __syncthreads();
      const int kmax = (g.d - dc0) < DC ? (g.d - dc0) : DC;
      for (int k = 0; k < kmax; ++k) {
        const v2d b = *reinterpret_cast<const v2d*>(&x2s[k][2 * cj]);
        const double pk = per_s[k], il = 1.0 / len_s[k];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const double a = x1s[ri * 8 + r][k];
          double d0 = a - b.x, d1 = a - b.y;
          if (KIND == GPX_PERIODIC) {
            const double u0 = M_PI * d0 / pk, u1 = M_PI * d1 / pk;
            double s0, c0, s1, c1;
            sincos(u0, &s0, &c0);
            sincos(u1, &s1, &c1);
            ps[r][0] = fma(u0 * 2.0 * s0 * c0, il * il, ps[r][0]);
            ps[r][1] = fma(u1 * 2.0 * s1 * c1, il * il, ps[r][1]);
            d0 = s0 * il; d1 = s1 * il;
          }
          r2[r][0] = fma(d0, d0, r2[r][0]);
          r2[r][1] = fma(d1, d1, r2[r][1]);
        }
      }
    }

    // ---- per entry: G = alpha_i alpha_j - Kinv_ij, weight 2 off the diagonal (lower triangle only)
    double gq[8][2];
    const int64_t col = j0 + 2 * cj;
    const double aj0 = col < g.n ? g.alpha[col] : 0.0, aj1 = col + 1 < g.n ? g.alpha[col + 1] : 0.0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int64_t row = i0 + ri * 8 + r;
      const bool rv = row < g.n;
      const double ai = rv ? g.alpha[row] : 0.0;
      v2d kv = {0.0, 0.0};
      if (rv) kv = *reinterpret_cast<const v2d*>(g.kinv + row * g.ldk + col);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int64_t cc = col + c;
        const double w = (rv && cc < g.n) ? (cc < row ? 2.0 : (cc == row ? 1.0 : 0.0)) : 0.0;
        const double G = w * (ai * (c ? aj1 : aj0) - (c ? kv.y : kv.x));
        const double rr2 = r2[r][c];
        double e0, q;
        if (KIND == GPX_RBF) {
          e0 = exp(-0.5 * rr2); q = e0;
        } else if (KIND == GPX_MATERN52) {
          const double rr = sqrt(rr2 + 1e-12);
          const double s5 = 2.2360679774997896964;
          const double ex = exp(-s5 * rr);
          e0 = (1.0 + s5 * rr + (5.0 / 3.0) * rr2) * ex;
          q = (5.0 / 3.0) * (1.0 + s5 * rr2 / rr) * ex;
        } else {
          e0 = exp(-2.0 * rr2); q = 4.0 * e0;
        }
        accS = fma(G, e0, accS);
        if (cc == row) accN += G;
        const double v = G * q;
        gq[r][c] = v;
        accLi = fma(v, rr2, accLi);
        if (KIND == GPX_PERIODIC) accPi = fma(v, ps[r][c], accPi);
      }
    }

    // ---- pass B: per-dimension sums of this workgroup's chunk (ARD length-scales / periods only)
    if (g.per_dim) {
      __syncthreads();
      stage(i0, j0, chunk0, KIND != GPX_PERIODIC);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < DC; ++k) {
        if (chunk0 + k < g.d) {   // uniform
          const v2d b = *reinterpret_cast<const v2d*>(&x2s[k][2 * cj]);
          const double pk = per_s[k], il = 1.0 / len_s[k];
          double sl = 0.0, sp = 0.0;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const double a = x1s[ri * 8 + r][k];
            double d0 = a - b.x, d1 = a - b.y;
            if (KIND == GPX_PERIODIC) {
              const double u0 = M_PI * d0 / pk, u1 = M_PI * d1 / pk;
              double s0, c0, s1, c1;
              sincos(u0, &s0, &c0);
              sincos(u1, &s1, &c1);
              sp = fma(gq[r][0], u0 * 2.0 * s0 * c0 * il * il, sp);
              sp = fma(gq[r][1], u1 * 2.0 * s1 * c1 * il * il, sp);
              d0 = s0 * il; d1 = s1 * il;
            }
            sl = fma(gq[r][0], d0 * d0, sl);
            sl = fma(gq[r][1], d1 * d1, sl);
          }
          accL[k] += sl; accP[k] += sp;
        }
      }
    }
  }

  // ---- one reduction per workgroup, fixed order: lanes (butterfly), then waves 0..3
  const int lane = tid & 63, wave = tid >> 6;
  double vals[NPV];
  vals[0] = accS; vals[1] = accN; vals[2] = accLi; vals[3] = accPi;
#pragma unroll
  for (int k = 0; k < DC; ++k) { vals[4 + k] = accL[k]; vals[4 + DC + k] = accP[k]; }
#pragma unroll
  for (int v = 0; v < NPV; ++v) {
    const double s = wave_sum(vals[v]);
    if (lane == 0) red[wave][v] = s;
  }
  __syncthreads();
  if (tid < NPV) {
    const double s = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    g.part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * NPV + tid] = s;
    if (g.mirror) g.mirror[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * NPV + tid] = s;
  }
}

// sums[chunk][v] = sum over workgroups of part[chunk][wg][v], fixed association (strided partial sums per
// thread, then a binary tree): bitwise reproducible for a given launch geometry
__global__ __launch_bounds__(256) void grad_final_kernel(const double* part, double* sums, int nwg, int64_t s_part, int64_t s_sums,
                                                         double* mirror) {
  __shared__ double sh[256];
  const int chunk = blockIdx.y, v = blockIdx.x, tid = threadIdx.x;
  part += (int64_t)blockIdx.z * s_part; sums += (int64_t)blockIdx.z * s_sums;
  double s = 0.0;
  for (int w = tid; w < nwg; w += 256) s += part[((int64_t)chunk * nwg + w) * NPV + v];
  sh[tid] = s;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) sh[tid] += sh[tid + off];
    __syncthreads();
  }
  if (tid == 0) {
    sums[chunk * NPV + v] = sh[0];
    if (mirror) mirror[chunk * NPV + v] = sh[0];   // single draw only (the caller passes NULL for a batch)
  }
}

int grad_partials_per_chunk(void) { return NPV; }
int grad_chunk_dims(void) { return DC; }

// part: device scratch of nchunk * nwg * NPV doubles; sums: device, nchunk * NPV doubles
int launch_grad_reduce(gpx_handle* h, hipStream_t s, int kind, const double* x, const double* hyp, const double* kinv,
                       int64_t ldk, const double* alpha, int64_t n, int d, int per_dim, int nwg, double* part,
                       double* sums, int batch, int64_t s_hyp, int64_t s_kinv, int64_t s_out, double* mirror_sums) {
  GradArgs g;
  if (batch != 1) mirror_sums = nullptr;
  g.x = x; g.hyp = hyp; g.kinv = kinv; g.alpha = alpha; g.part = part;
  g.n = n; g.ldk = ldk; g.d = d; g.per_dim = per_dim;
  if (batch < 1) { gpx_set_error(h, "grad_reduce: batch < 1"); return GPX_ERR_ARG; }
  const int64_t T = (n + GT - 1) / GT;
  g.ntiles = T * (T + 1) / 2;
  const int nchunk = per_dim ? (d + DC - 1) / DC : 1;
  if ((ldk & 1) || (reinterpret_cast<uintptr_t>(kinv) & 15)) {
    gpx_set_error(h, "grad_reduce: K^-1 buffer must be 16-B aligned with an even leading dimension");
    return GPX_ERR_ARG;
  }
  // one workgroup per chunk: its partial sums ARE the sums (same layout), no final-sum launch -- at the sizes where this
  // applies (the caller passes nwg = 1 for up to three tiles) a launch costs more than the three tiles
  const bool direct = nwg == 1;
  if (direct) g.part = sums;
  g.mirror = direct ? mirror_sums : nullptr;
  // draw b: hyp + b s_hyp, kinv + b s_kinv, alpha and sums inside its result block (+ b s_out), part + b nchunk nwg NPV
  g.s_hyp = s_hyp; g.s_kinv = s_kinv; g.s_alpha = s_out; g.s_part = direct ? s_out : (int64_t)nchunk * nwg * NPV;
  dim3 grid((unsigned)nwg, (unsigned)nchunk, (unsigned)batch);
  switch (kind) {
    case GPX_RBF: hipLaunchKernelGGL((grad_reduce_kernel<GPX_RBF>), grid, dim3(256), 0, s, g); break;
    case GPX_MATERN52: hipLaunchKernelGGL((grad_reduce_kernel<GPX_MATERN52>), grid, dim3(256), 0, s, g); break;
    case GPX_PERIODIC: hipLaunchKernelGGL((grad_reduce_kernel<GPX_PERIODIC>), grid, dim3(256), 0, s, g); break;
    default: gpx_set_error(h, "unknown kernel kind"); return GPX_ERR_ARG;
  }
  GPX_HIP_CHECK(h, hipGetLastError());
  if (direct) return GPX_OK;
  hipLaunchKernelGGL(grad_final_kernel, dim3(NPV, (unsigned)nchunk, (unsigned)batch), dim3(256), 0, s, part, sums, nwg,
                     (int64_t)nchunk * nwg * NPV, s_out, mirror_sums);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

// alpha_i = sum_{k >= i} U_ik z_k  (U = L^-T upper triangular, row-major): one wave per row
__global__ __launch_bounds__(256) void utri_gemv_kernel(const double* U, int64_t ld, const double* z, double* alpha,
                                                        int64_t n, int64_t ncols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const double* u = U + row * ld;
  double s = 0.0;
  for (int64_t k = row + lane; k < ncols; k += 64) s = fma(u[k], z[k], s);
  s = wave_sum(s);
  if (lane == 0) alpha[row] = s;
}

int launch_utri_gemv(gpx_handle* h, hipStream_t s, const double* U, int64_t ld, const double* z, double* alpha,
                     int64_t n, int64_t ncols) {
  if (n <= 0) return GPX_OK;
  hipLaunchKernelGGL(utri_gemv_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, U, ld, z, alpha, n, ncols);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}
