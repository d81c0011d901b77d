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

#define GPX_ABI_VERSION 2

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
#define GPX_ERR_RETRY (-7)       /* an in-kernel hand-over timed out; the handle has switched to kernel-boundary hand-overs:
                                    repeat the call (the synchronous entry points do so themselves; of the asynchronous
                                    pair, gpx_fetch_results returns it) */
#define GPX_ERR_INTERNAL (-8)    /* a bounded wait between the waves of potrf_diag ran out: a logic error, never observed;
                                    the results of the call are undefined */

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

/* `batch` gradient evaluations in ONE schedule -- the reference's NUTS chains run vectorised (chain_method, gpax/gp.py:507-509,
 * 517-525) as one jitted program; here every kernel of gpx_lml_grad is launched once with the draw index in its grid.
 * params[batch] (same kind, apply_noise, form of k_noise / k_length / k_period), y_minus_m (batch, n), logp[batch],
 * grad_out (batch, ng) with ng = 2 + n_length (+ n_period for the periodic kernel) laid out per draw as in gpx_lml_grad,
 * alpha_out (batch, n) optional, info_out[batch] optional.  A draw's numbers are those of a gpx_lml_grad call with the same
 * grouping thresholds, bit for bit.  Leaves no cached factor.  Returns the largest info. */
int gpx_lml_grad_batch(gpx_handle* h, int32_t batch, const gpx_kernel_params* params, const double* y_minus_m,
                       double* logp, double* grad_out, double* alpha_out, int32_t* info_out);

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

/* jax.vmap over hyper-parameter draws (gpax/gp.py:354-364, 384-390) for the fused step: `batch` draws of the hyper-
 * parameters in ONE schedule -- one bordered matrix per draw, every kernel launched once for all draws (draw index in the
 * grid), one result block.  params[batch]; y_minus_m (batch, n); m_new (batch, m) or NULL; new_noise[batch] or NULL;
 * outputs logp[batch], mean_out / var_out (batch, m), info_out[batch] (each optional).  All draws share the kernel kind,
 * apply_noise and the form of k_noise (scalar or per-point vector).  A draw's results carry the same bits as a
 * gpx_fit_predict call with its parameters.  Returns the largest info (> 0: that draw's outputs are NaN).  Does not leave
 * a cached factor behind. */
int gpx_fit_predict_batch(gpx_handle* h, int32_t batch, const gpx_kernel_params* params, const double* y_minus_m,
                          const double* m_new, const double* new_noise, double* logp, double* mean_out, double* var_out,
                          int32_t* info_out);

/* _sample_gp_given_single_hp under the vmap of _sample_prior / _sample_posterior (gpax/gp.py:314-338, 352-365, 382-391) for
 * `batch` hyper-parameter draws in one schedule, device-resident: per draw the posterior (conditioned != 0, gpax/gp.py:247-312
 * Cholesky branch) or prior (conditioned == 0, gpax/gp.py:212-245) mean and covariance at the resident x_new, sd =
 * sqrt(diag cov), and nsamp draws  mean + eps chol(cov)^T  (eps (batch, nsamp, m) standard-normal draws from the caller).
 * The (m, m) covariances and their factors never leave HBM.  Outputs: mean_out / sd_out (batch, m) (optional),
 * samples_out (batch, nsamp, m), info_out[batch] (optional). */
int gpx_sample_batch(gpx_handle* h, int32_t batch, const gpx_kernel_params* params, const double* y_minus_m,
                     const double* m_new, const double* new_noise, const double* eps, int64_t nsamp, int32_t conditioned,
                     double* mean_out, double* sd_out, double* samples_out, int32_t* info_out);

/* Analytic acquisition functions (gpax/acquisition.py:260-266 UpperConfidenceBound._analytic, :291-298
 * ExpectedImprovement._analytic) as an epilogue over the posterior mean / variance the last gpx_posterior /
 * gpx_fit_predict[_async] left in HBM:
 *   mu = mean * mu_scale + mu_shift,  sd = sqrt(var) * sd_scale     (the output transform's reverse, gpax/gp.py:499-500)
 *   GPX_ACQ_EI : param = y_max;  u = (mu - y_max) / sd;  sd * pdf(u) + (mu - y_max) * cdf(u)
 *   GPX_ACQ_UCB: param = beta;   mu + sqrt(beta + 1e-12) * sd  (beta = inf: sd, beta = 0: mu)
 * acq_out (m) optional; argmax_out / best_out: first index of the largest value and that value (what
 * _optimize_halton keeps, gpax/acquisition.py:211-216).
 * The resident (mean, var) is dropped by gpx_set_train / gpx_set_test / every new factorisation / gpx_lml_grad
 * (GPX_ERR_STATE until the next posterior).  The factorisation's info word is read with the result: > 0 is returned as is
 * (non-PD factor: the values are NaN), a timed-out in-kernel hand-over as GPX_ERR_RETRY (repeat the posterior call). */
#define GPX_ACQ_EI 0
#define GPX_ACQ_UCB 1
int gpx_acquisition(gpx_handle* h, int32_t kind, double param, double mu_scale, double mu_shift, double sd_scale,
                    double* acq_out, int64_t* argmax_out, double* best_out);

/* "built=<HIP_VERSION> runtime=<hipRuntimeGetVersion()> path=<shared object the HIP runtime was loaded from>".  Returns
 * GPX_ERR_UNSUPPORTED when the runtime the process resolved is not the major.minor this library was compiled against
 * (e.g. PyTorch's bundled libamdhip64 imported first), GPX_OK otherwise. */
int gpx_runtime_info(char* buf, int len);

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
 *   "lookahead"    0 = everything on one stream, 1 = the split-panel schedule on four streams (1)
 *   "resident"     split-panel schedule: 0 = the chain's kernels hand over at kernel boundaries, 1 = potrf_diag starts
 *                  resident and sleeps on a device flag, 2 = so does the band kernel S = A W_k^T, A -= S S^T, 3 = and ONE
 *                  potrf_diag workgroup per draw walks the whole diagonal, keeping its CU (2; falls back to 0 under
 *                  hardware-counter collection, under per-launch bracketing, and for good once a wait timed out)
 *   "pair_min" / "quad_min"  group the bulk updates two / four panels at a time while more than this many block
 *                  columns remain; 0 = never (28 / 40)
 *   "split_min_blocks" the split-panel schedule runs from this many block columns; below, one stream (3)
 *   "la_rows"      look-ahead column updates run on 64x64 tiles while at least this many rows remain below them; 0 = never (3072)
 *   "bulk_first_col" ungrouped bulk launches enumerate and publish their first block column first, so that the next column
 *                  update and the band kernel wait for that counter instead of the launch's end; 0 = completion events (1)
 *   "batch_group"  batched draws (gpx_fit_predict_batch, gpx_sample_batch) with four or more draws: 2 | 4 = the bulk updates are
 *                  grouped that many panels at a time from the first step to the last (a batch is throughput-bound throughout);
 *                  the arithmetic is that of a single-draw call with "pair_min" = 1 and "quad_min" = 2 (4) or 0 (2).
 *                  0 = the single-draw thresholds, i.e. the bits of default single-draw calls (4)
 *   "debug_force_timeout"  test hook: 1 = the first in-kernel wait of the next factorisation gives up at once, so the
 *                  GPX_ERR_RETRY path (kernel-boundary hand-overs, repeated call) can be exercised; clears itself (0) */
int gpx_set_option(gpx_handle* h, const char* name, int64_t value);
int gpx_get_option(gpx_handle* h, const char* name, int64_t* value);

#ifdef __cplusplus
}
#endif
#endif /* GPX_H */
