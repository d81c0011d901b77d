// reduce.hip -- epilogue reductions and the vector back-substitution helper (gfx950).
//
//   log-marginal pieces   sum_i log L_ii  and  |z|^2, z = L^-1 (y - m)   (numpyro MVN.log_prob
//                         reached from gpax/gp.py:147-151)
//   posterior             mean = m_new + V^T z,  var = k(x*,x*) - |V_m|^2   with V^T = k_cross L^-T
//                         (gpax/gp.py:297-301 in the V-form; only diag(cov) is used by
//                         _fast_predict, gpax/gp.py:484)
// All HBM-bound row reductions: each row is read once, contiguously.

#include "gpx_internal.h"

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// deterministic block-wide sums of two values; result valid in thread 0
template <int NT>
__device__ __forceinline__ void block_sum2(double& a, double& b) {
  __shared__ double sa[NT / 64], sb[NT / 64];
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sa[wave] = a; sb[wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < NT / 64; ++w) { ta += sa[w]; tb += sb[w]; }
    a = ta; b = tb;
  }
}

__global__ __launch_bounds__(256) void post_reduce_kernel(const double* vt, int64_t ld, int64_t n,
                                                         const double* z, const double* m_new,
                                                         double prior_diag, const double* dev_prior_diag,
                                                         double* mean, double* var) {
  const int64_t row = blockIdx.x;
  const double* v = vt + row * ld;
  double dot = 0.0, ss = 0.0;
  for (int64_t c = threadIdx.x; c < n; c += 256) {
    const double x = v[c];
    dot = fma(x, z[c], dot);
    ss = fma(x, x, ss);
  }
  block_sum2<256>(dot, ss);
  if (threadIdx.x == 0) {
    mean[row] = (m_new ? m_new[row] : 0.0) + dot;
    var[row] = (dev_prior_diag ? dev_prior_diag[0] : prior_diag) - ss;
  }
}

// Fused epilogue: block 0 = the log-marginal pieces (and, on request, the copy of z into the handle's z
// buffer for later posterior calls), blocks 1..m = one test row each.  One launch instead of a device copy and
// two kernels at the tail of every fit+predict step.
// Column range [c0, c1) and `mode`: 0 = everything in one launch (c0 = 0, c1 = n); 1 = PARTIAL sums over the range,
// written raw to part[] = [logdet, quad, dot(m), ss(m)] -- launched on the panel stream as soon as all block columns
// but the last are final, i.e. beside the last potrf_diag; 2 = FINAL: the rest of the range plus part[], finalised.
__global__ __launch_bounds__(256) void epilogue_kernel(const double* L, int64_t ld, int64_t n, const double* z,
                                                      double* z_copy, int64_t npad, double* scal, const double* vt,
                                                      int64_t vt_ld, const double* m_new, double prior_diag,
                                                      const double* dev_prior_diag, double* mean, double* var,
                                                      int64_t c0, int64_t c1, double* part, int64_t m, int mode,
                                                      const double* U, double* alpha, int64_t gemv_ncols, int gemv_block0,
                                                      int64_t s_mat, int64_t s_out, int64_t s_zcopy, int64_t s_mnew, int64_t s_hyp,
                                                      int64_t s_part, double* mirror, const int* info_src, int64_t mirror_var_off) {
  if (blockIdx.y) {   // draw y of a batch
    const int64_t b = blockIdx.y;
    L += b * s_mat; if (z) z += b * s_mat; if (vt) vt += b * s_mat; if (U) U += b * s_mat;
    scal += b * s_out; if (mean) mean += b * s_out; if (var) var += b * s_out; if (alpha) alpha += b * s_out;
    if (z_copy) z_copy += b * s_zcopy;
    if (m_new) m_new += b * s_mnew;
    if (dev_prior_diag) dev_prior_diag += b * s_hyp;
    if (part) part += b * s_part;
  }
  if (U && (int)blockIdx.x >= gemv_block0) {
    // gradient path, riding on the same launch: alpha_i = sum_{k >= i} U_ik z_k (U = L^-T, row-major), one wave per row
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)(blockIdx.x - gemv_block0) * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const double* u = U + row * ld;
    double s = 0.0;
    for (int64_t k = row + lane; k < gemv_ncols; k += 64) s = fma(u[k], z[k], s);
    s = wave_sum(s);
    if (lane == 0) {
      alpha[row] = s;
      if (mirror && mode != 1) mirror[GPX_OUT_HEAD + row] = s;   // alpha sits where the posterior mean would (api.hip: lml_grad_once)
    }
    return;
  }
  if (blockIdx.x == 0) {
    double s1 = 0.0, s2 = 0.0;
    if (mode != 1 && z_copy)
      for (int64_t i = threadIdx.x; i < npad; i += 256) z_copy[i] = z ? z[i] : 0.0;
    for (int64_t i = c0 + threadIdx.x; i < c1; i += 256) {
      const double v = z ? z[i] : 0.0;
      if (i < n) { s1 += log(L[i * ld + i]); s2 = fma(v, v, s2); }
    }
    block_sum2<256>(s1, s2);
    if (threadIdx.x == 0) {
      if (mode == 1) { part[0] = s1; part[1] = s2; }
      else if (mode == 2) { scal[0] = part[0] + s1; scal[1] = part[1] + s2; }
      else { scal[0] = s1; scal[1] = s2; }
      if (mirror && mode != 1) {   // the pinned host mirror of the result block (every writer of info has finished: stream order)
        mirror[0] = scal[0]; mirror[1] = scal[1];
        reinterpret_cast<int*>(mirror + 8)[0] = __hip_atomic_load(info_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  const int64_t row = blockIdx.x - 1;
  const double* v = vt + row * vt_ld;
  double dot = 0.0, ss = 0.0;
  const int64_t cend = c1 < n ? c1 : n;
  for (int64_t c = c0 + threadIdx.x; c < cend; c += 256) {
    const double x = v[c];
    dot = fma(x, z[c], dot);
    ss = fma(x, x, ss);
  }
  block_sum2<256>(dot, ss);
  if (threadIdx.x == 0) {
    if (mode == 1) { part[2 + row] = dot; part[2 + m + row] = ss; return; }
    if (mode == 2) { dot += part[2 + row]; ss += part[2 + m + row]; }
    const double mu = (m_new ? m_new[row] : 0.0) + dot, vr = (dev_prior_diag ? dev_prior_diag[0] : prior_diag) - ss;
    mean[row] = mu;
    var[row] = vr;
    if (mirror) { mirror[GPX_OUT_HEAD + row] = mu; mirror[GPX_OUT_HEAD + mirror_var_off + row] = vr; }
  }
}

static int launch_epilogue_range(gpx_handle* h, hipStream_t s, const EpilogueLaunch& e, int64_t c0, int64_t c1, double* part, int mode) {
  const int64_t rows = (e.m > 0 && e.z) ? e.m : 0;
  const int64_t gemv_blocks = (e.gemv_U && e.z) ? (e.n + 3) / 4 : 0;
  hipLaunchKernelGGL(epilogue_kernel, dim3((unsigned)(rows + 1 + gemv_blocks), (unsigned)e.batch), dim3(256), 0, s, e.L, e.ld, e.n, e.z, e.z_copy,
                     e.z_copy ? e.npad : e.n, e.scal, e.vt, e.vt_ld, e.m_new, e.prior_diag, e.dev_prior_diag, e.mean, e.var,
                     c0, c1, part, rows, mode, gemv_blocks ? e.gemv_U : nullptr, e.gemv_alpha, e.gemv_ncols, (int)(rows + 1),
                     e.s_mat, e.s_out, e.s_zcopy, e.s_mnew, e.s_hyp, e.s_part, (mode != 1 && e.batch == 1) ? e.mirror : nullptr, e.info_src,
                     e.mirror_var_off);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

int launch_epilogue(gpx_handle* h, hipStream_t s, const EpilogueLaunch& e) {
  return launch_epilogue_range(h, s, e, 0, e.z_copy ? e.npad : e.n, nullptr, 0);
}

// the two halves of a split epilogue: columns [0, c1) early (partial sums into part[2 + 2m]), [c1, n) + part at the end
int launch_epilogue_partial(gpx_handle* h, hipStream_t s, const EpilogueLaunch& e, int64_t c1, double* part) {
  return launch_epilogue_range(h, s, e, 0, c1, part, 1);
}
int launch_epilogue_final(gpx_handle* h, hipStream_t s, const EpilogueLaunch& e, int64_t c0, double* part) {
  return launch_epilogue_range(h, s, e, c0, e.z_copy ? e.npad : e.n, part, 2);
}

int launch_post_reduce(gpx_handle* h, hipStream_t s, const EpilogueLaunch& e) {
  if (e.m <= 0) return GPX_OK;
  hipLaunchKernelGGL(post_reduce_kernel, dim3((unsigned)e.m), dim3(256), 0, s, e.vt, e.vt_ld, e.n, e.z, e.m_new, e.prior_diag,
                     e.dev_prior_diag, e.mean, e.var);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

// y[0:ncols) = beta*y + alpha * sum_{r<nrows} x[r] * A[r][0:ncols)    (A row-major: coalesced along c)
__global__ __launch_bounds__(256) void gemv_t_kernel(const double* A, int64_t lda, int nrows,
                                                    int64_t ncols, const double* x, double* y,
                                                    double alpha, double beta) {
  __shared__ double xs[256];
  if (threadIdx.x < nrows) xs[threadIdx.x] = x[threadIdx.x];
  __syncthreads();
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= ncols) return;
  double acc = 0.0;
  for (int r = 0; r < nrows; ++r) acc = fma(xs[r], A[(int64_t)r * lda + c], acc);
  y[c] = (beta != 0.0 ? beta * y[c] : 0.0) + alpha * acc;
}

int launch_gemv_t(gpx_handle* h, hipStream_t s, const double* A, int64_t lda, int nrows, int64_t ncols,
                  const double* x, double* y, double alpha, double beta) {
  if (ncols <= 0) return GPX_OK;
  if (nrows > 256) { gpx_set_error(h, "gemv_t: nrows > 256"); return GPX_ERR_ARG; }
  hipLaunchKernelGGL(gemv_t_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, s, A, lda, nrows,
                     ncols, x, y, alpha, beta);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

// ---- acquisition epilogue over the resident (mean, var) of the last posterior (gpax/acquisition.py:260-266, 291-298) ----
// kind 0: expected improvement  EI = sd * pdf(u) + (mu - y_max) * cdf(u),  u = (mu - y_max) / sd
// kind 1: upper confidence bound  mu + sqrt(beta + 1e-12) * sd   (beta = inf: sd; beta = 0: mu)
// mu, sd in RAW units: mu = mean * mu_scale + mu_shift, sd = sqrt(var) * sd_scale (the output transform's reverse,
// gpax/transforms.py:262-274, 311-317, is affine).  One workgroup; also the arg-max (first index of the largest value).
__global__ __launch_bounds__(1024) void acquisition_kernel(const double* mean, const double* var, int64_t m, int kind, double param,
                                                          double mu_scale, double mu_shift, double sd_scale, double* acq,
                                                          double* best /* [value, index] */) {
  __shared__ double sv[1024];
  __shared__ long long si[1024];
  double bv = -INFINITY; long long bi = -1;
  for (int64_t i = threadIdx.x; i < m; i += 1024) {
    const double mu = fma(mean[i], mu_scale, mu_shift), sd = sqrt(var[i]) * sd_scale;
    double a;
    if (kind == 0) {
      const double u = (mu - param) / sd;
      const double cdf = 0.5 * erfc(-u * 0.70710678118654752440);
      const double pdf = exp(-0.5 * u * u - 0.91893853320467274178);   // - 0.5 * log(2 pi)
      a = sd * pdf + (mu - param) * cdf;
    } else {
      a = isinf(param) ? sd : (param == 0.0 ? mu : mu + sqrt(param + 1e-12) * sd);
    }
    acq[i] = a;
    if (a > bv || bi < 0) { bv = a; bi = i; }   // strided scan: the first index wins on ties within a thread
  }
  sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      const double ov = sv[threadIdx.x + o]; const long long oi = si[threadIdx.x + o];
      if (oi >= 0 && (si[threadIdx.x] < 0 || ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x]))) {
        sv[threadIdx.x] = ov; si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { best[0] = sv[0]; best[1] = (double)si[0]; }
}

int launch_acquisition(gpx_handle* h, hipStream_t s, const double* mean, const double* var, int64_t m, int kind, double param,
                       double mu_scale, double mu_shift, double sd_scale, double* acq, double* best) {
  hipLaunchKernelGGL(acquisition_kernel, dim3(1), dim3(1024), 0, s, mean, var, m, kind, param, mu_scale, mu_shift, sd_scale, acq, best);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}
