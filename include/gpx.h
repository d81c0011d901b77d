/* gpx.h -- C ABI of the MI355X-native exact-GP core (libgpx_hip.so).
 *
 * Drop-in boundary for ONE hot path of matthewcarbone/gpax: gram build ->
 * Cholesky -> log-marginal likelihood -> predictive posterior, all fp64.
 * The reference has no FFI of its own (pure Python over JAX/NumPyro); each entry
 * point below names the reference interface it replaces (file:line relative to
 * the reference checkout).  INTEGRATION.md shows the ctypes binding a gpax
 * maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types;
 *   - every array is row-major contiguous `double`; HOST pointers, owned by the
 *     caller and never retained past the call (device copies live in the handle);
 *   - every function returns an int status: 0 = ok, > 0 = LAPACK-style `info`
 *     (1-based index of the first non-positive pivot; outputs are NaN-filled from
 *     there on, mirroring JAX's silent-NaN Cholesky), < 0 = GPX_ERR_*;
 *   - one caller thread per handle (not re-entrant); ctypes drops the GIL.
 */
#ifndef GPX_H
#define GPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPX_ABI_VERSION 1

/* kernel kinds: gpax/kernels.py:110 (RBFKernel), :129 (MaternKernel nu=5/2), :150 (PeriodicKernel) */
#define GPX_RBF 0
#define GPX_MATERN52 1
#define GPX_PERIODIC 2

#define GPX_OK 0
#define GPX_ERR_ARG (-1)         /* bad argument / shape */
#define GPX_ERR_HIP (-2)         /* HIP runtime error (see gpx_last_error) */
#define GPX_ERR_STATE (-3)       /* call order: e.g. posterior before fit */
#define GPX_ERR_NOMEM (-4)
#define GPX_ERR_COMM (-5)        /* RCCL error */
#define GPX_ERR_UNSUPPORTED (-6)

typedef struct gpx_handle gpx_handle;

/* The keyword set of Kernel.__call__ / _kernel_function
 * (gpax/kernels.py:112-121 RBF, :131-140 Matern, :152-162 periodic).
 * Hyper-parameters arrive from gpax with shape (1,) or (plate,):
 *   k_length: n_length = 1 (isotropic) or d (ARD, gpax/kernels.py:45-46 broadcast)
 *   k_period: n_period = 0 (not periodic), 1 or d
 *   k_noise : n_noise  = 0 (noise = 0.0), 1 (scalar k_noise) or n1 (vector, the
 *             y_std**2 of gpax/gp.py:289-293)
 * The diagonal rule `(k_noise + k_jitter) * eye(n)` is applied iff
 * apply_noise != 0 AND n1 == n2 (gpax/kernels.py:124: keyed on shapes). */
typedef struct gpx_kernel_params {
  int32_t kind;
  int32_t n_length;
  int32_t n_period;
  int32_t n_noise;
  int32_t apply_noise;
  int32_t reserved;
  double k_scale;
  double k_jitter;
  const double* k_length;
  const double* k_period;
  const double* k_noise;
} gpx_kernel_params;

int gpx_abi_version(void);

/* Replaces gpax.state.set_device (gpax/state.py:61-70): bind a handle to one GPU.
 * The handle owns its HIP streams, workspaces and the cached factor. */
int gpx_create(int device_id, gpx_handle** out);
int gpx_destroy(gpx_handle* h);
/* Text of the last error on this handle (h may be NULL: last create error). */
const char* gpx_last_error(const gpx_handle* h);
int gpx_device_name(gpx_handle* h, char* buf, int len);
int gpx_synchronize(gpx_handle* h);

/* Kernel.__call__(x1, x2, **params) -> (n1, n2)  (gpax/kernels.py:93-94 and the
 * three _kernel_function bodies).  x1 (n1,d), x2 (n2,d), out (n1,n2). */
int gpx_gram(gpx_handle* h, const gpx_kernel_params* p, const double* x1, int64_t n1,
             const double* x2, int64_t n2, int32_t d, double* out);

/* jax.device_put(x, state.device) of gpax/gp.py:527 / :396: make the (already
 * transformed) training inputs x (n,d) / test inputs x_new (m,d) resident in HBM. */
int gpx_set_train(gpx_handle* h, const double* x, int64_t n, int32_t d);
int gpx_set_test(gpx_handle* h, const double* x_new, int64_t m);

/* k_train = f(x, x, k_noise, k_jitter) + cho_factor  (gpax/gp.py:293,296; the same
 * factor is the one inside dist.MultivariateNormal at gpax/gp.py:147-151).
 * Builds K on the device, factorises K = L L^T (lower) and caches L.
 * K_out / L_out: optional (may be NULL) host copies, (n,n) each; L_out has zeros
 * above the diagonal. */
int gpx_fit_factor(gpx_handle* h, const gpx_kernel_params* p, double* K_out, double* L_out);

/* log-density of MultivariateNormal(loc=m, covariance_matrix=k) at y
 * (gpax/gp.py:145-151, `obs=y`): uses the cached L.  y_minus_m (n,) = y - mean(x).
 * logp = -0.5*|L^-1 r|^2 - sum(log diag L) - n/2*log(2 pi).
 * alpha_out: optional (n,) = K^-1 r  (the `B` of gpax/gp.py:300). */
int gpx_log_marginal(gpx_handle* h, const double* y_minus_m, double* logp, double* alpha_out);

/* Log-marginal likelihood of gpx_log_marginal AND its gradient with respect to the kernel
 * hyper-parameters at one setting: what jax.grad of the NumPyro potential of _gp_prior
 * (gpax/gp.py:133-151) hands NUTS at every leapfrog step (gpax/gp.py:515-530) and SVI at every
 * optimiser step (gpax/gp.py:583-587), here in closed form
 *   d logp/d theta = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta),  alpha = K^-1 r,
 * with dK/dtheta of the kernels of gpax/kernels.py:112-168 recomputed on the fly (no dK matrix in HBM).
 * grad_out (doubles): [0] d/dk_scale | n_length entries d/dk_length | n_period entries d/dk_period
 * (periodic kernel only) | 1 entry d/dk_noise (0 unless apply_noise and n_noise == 1; a per-point
 * noise vector is data).  k_length / k_period with one entry get the derivative summed over dimensions.
 * alpha_out optional (n,) = K^-1 r = d logp / d mean(x).  Caches L and z like gpx_fit_predict.
 * Returns info like gpx_fit_factor (info > 0: NaN outputs). */
int gpx_lml_grad(gpx_handle* h, const gpx_kernel_params* p, const double* y_minus_m, double* logp,
                 double* grad_out, double* alpha_out);

/* _get_mean_and_covariance (gpax/gp.py:247-312, Cholesky branch :295-301) at the
 * resident x_new with the cached factor and the y_minus_m of the last
 * gpx_log_marginal:  mean = m_new + k_cross K^-1 r,  cov = k_new - k_cross K^-1 k_cross^T,
 * where k_new = f(x_new, x_new, k_noise=new_noise, k_jitter) (gpax/gp.py:283-287:
 * new_noise = k_noise if observation_noise else 0).
 * m_new (m,) may be NULL (= 0).  Outputs (each optional): mean_out (m,),
 * var_out (m,) = diag(cov)  [what _fast_predict uses, gpax/gp.py:484], cov_out (m,m). */
int gpx_posterior(gpx_handle* h, const double* m_new, double new_noise, double* mean_out,
                  double* var_out, double* cov_out);

/* _get_mean_and_covariance_unconditioned's covariance (gpax/gp.py:212-245):
 * cov = f(x_new, x_new, k_noise=new_noise, k_jitter) at the resident x_new. */
int gpx_prior_cov(gpx_handle* h, const gpx_kernel_params* p, double new_noise, double* cov_out);

/* dist.MultivariateNormal(mean, cov).sample(key, sample_shape=(n,))  (gpax/gp.py:335-337):
 * out (nsamp,m) = mean + eps @ chol(cov)^T with the caller's standard-normal draws eps (nsamp,m)
 * (JAX's threefry stream cannot be reproduced: statistical parity only).  cov (m,m); mean (m,) may be
 * NULL.  Returns info > 0 (and NaNs) when cov is not positive definite, as numpyro would. */
int gpx_mvn_sample(gpx_handle* h, const double* mean, const double* cov, int64_t m, const double* eps,
                   int64_t nsamp, double* out);

/* One whole fit+predict step at one hyper-parameter setting -- the unit of the
 * benchmark metric and of _fast_predict (gpax/gp.py:464-484):
 * gram(x,x) + Cholesky + log-marginal + gram(x_new,x) + posterior mean/variance,
 * run as ONE bordered factorisation (k_cross and r ride along as extra rows, so
 * the triangular solves are absorbed into the panel updates).  Also caches L.
 * Requires gpx_set_train; uses the resident x_new if gpx_set_test was called (m may be 0).
 * Any of logp / mean_out / var_out may be NULL. */
int gpx_fit_predict(gpx_handle* h, const gpx_kernel_params* p, const double* y_minus_m,
                    const double* m_new, double new_noise, double* logp, double* mean_out,
                    double* var_out);

/* Same work as gpx_fit_predict but only ENQUEUED on the handle's streams: inputs must
 * already be resident (gpx_set_train/gpx_set_test/gpx_set_targets), results stay in HBM
 * until gpx_fetch_results.  This is what bench.py times (inputs resident in HBM). */
int gpx_set_targets(gpx_handle* h, const double* y_minus_m, const double* m_new);
int gpx_fit_predict_async(gpx_handle* h, const gpx_kernel_params* p, double new_noise);
int gpx_fetch_results(gpx_handle* h, double* logp, double* mean_out, double* var_out);

/* ---- measurement hooks (bench.py; HIP events on the library's own streams) ---- */
#define GPX_STAGE_GRAM 0        /* gram build of K (n x n) */
#define GPX_STAGE_GRAM_CROSS 1  /* gram build of k_cross (m x n) */
#define GPX_STAGE_POTRF_DIAG 2  /* LDS-resident diagonal-block factorisation */
#define GPX_STAGE_TRSM_PANEL 3  /* panel solve as MFMA GEMM against inv(L_kk) */
#define GPX_STAGE_TRAILING 4    /* bulk trailing SYRK/GEMM update on fp64 MFMA (the dominant kernel) */
#define GPX_STAGE_EPILOGUE 5    /* log-det / quadratic form / mean / variance reductions */
#define GPX_STAGE_FACTOR_TOTAL 6
#define GPX_STAGE_LOOKAHEAD 7   /* near / far look-ahead column updates (small latency-path GEMMs) */
#define GPX_N_STAGES 8
/* mode 0 = off (default), 1 = bracket every launch of each stage with HIP events, 2 = only the whole step,
 * the gram build and the epilogue (3 pairs per step), 3 = only the launches of the two roofline kernels (bulk
 * update, gram build): what bench.py uses, since brackets on the chain stream stretch what they measure. */
int gpx_profile_mode(gpx_handle* h, int mode);
int gpx_profile_reset(gpx_handle* h);
/* diagnostic: CSV (stage,start_us,end_us) of the bracketed launches since the last reset/get */
int gpx_profile_dump(gpx_handle* h, const char* path);
/* total milliseconds, number of launches and algorithmic flops+bytes summed over them. */
int gpx_profile_get(gpx_handle* h, int stage, double* total_ms, int64_t* launches,
                    double* flops, double* bytes);

/* ---- tuning knobs (ints; unknown names -> GPX_ERR_ARG).  Defaults are the measured best on MI355X:
 *   "nb"           panel width, 64 | 128, or 0 = automatic: 64 for n <= 64, else 128 (0)
 *   "lookahead"    0 = everything on one stream, 1 = multi-stream schedules (1)
 *   "depth"        1 | 2 = classic look-ahead on two / three streams, 3 = split-panel schedule (3)
 *   "resident"     split-panel schedule: 0 = the chain's kernels hand over at kernel boundaries, 1 = potrf_diag starts
 *                  resident and sleeps on a device flag, 2 = so does S = A W_k^T (2; falls back to 0 under
 *                  hardware-counter collection, inside a stream capture and under per-launch bracketing)
 *   "pair_min" / "quad_min"  group the bulk updates two / four panels at a time while more than this many block
 *                  columns remain; 0 = never (28 / 40)
 *   "diag_band"    fused band kernel, ungrouped steps: the previous panel's update of the diagonal block is done by the band kernel
 *                  itself while it waits for potrf_diag; the panel stream's update of column k+2 starts below that block (1)
 *   "split_min_blocks" the split-panel (three-stream) schedule runs from this many block columns; below, one stream (3)
 *   "build_small"  one or two block columns on one stream (npad <= 256): the whole bordered matrix -- K, identity padding, k_cross, r, identity / zero
 *                  border rows, the cleared border of L -- is built by ONE launch instead of five to nine (1)
 *   "potrf_persist" ONE potrf_diag launch for all diagonal blocks of a factorisation (the workgroup keeps its CU; one-wave relay
 *                  kernels hand W_k to the panel stream).  Measured slower than one launch per block (0)
 *   "epi_split"    fit+predict: the posterior / log-marginal reductions over all block columns but the last are taken on the
 *                  panel stream as soon as those columns are final (beside the last potrf_diag); the epilogue behind the last
 *                  panel only adds the last block column (1)
 *   "lp_flag"      the last border panel waits for the panel stream's last column update through a device counter instead
 *                  of a stream-level event wait in front of the last potrf_diag (1)
 *   "border_side"  split-panel schedule: the border (k_cross rows, r row) is built on the panel stream, whose first kernel is its
 *                  first reader, instead of on the chain stream in front of potrf_diag(0) (1)
 *   "fuse_band"    in-kernel hand-over schedule: the chain's two band kernels (S = A W_k^T, A -= S S^T) run as ONE launch
 *                  whose second phase waits on a device counter for the first phase's tiles (1)
 *   "near_block"   grouped phases: at a group end the next group's block columns are updated by the panel stream in a
 *                  launch of its own and the bulk launch starts behind them, so the next group's chain runs while the bulk
 *                  launch does (1)
 *   "la_rows"      look-ahead column updates run on 64x64 tiles while at least this many rows remain below them; 0 = never (3072)
 *   "col_split"    the two look-ahead columns are updated by separate kernels (only column k+2 behind the wait for the
 *                  bulk update) while at least this many block columns are left to the bulk update; 0 = never (0)
 *   "host_threads" 2 = the panel / bulk streams are fed by a second host thread (1)
 *   "graph"        1 = replay the fused step from a captured hipGraph (0: measured slower on ROCm 7.2) */
int gpx_set_option(gpx_handle* h, const char* name, int64_t value);
int gpx_get_option(gpx_handle* h, const char* name, int64_t* value);

#ifdef __cplusplus
}
#endif
#endif /* GPX_H */
