/* gpx_dist.h -- C ABI of the distributed (one process per GPU) factorisation, libgpx_hip.so.
 *
 * The reference has no multi-GPU code (gpax/state.py:64 "# TODO: multi-gpu support?"); these entry
 * points are the second sharding axis of SURVEY.md section 8(e): ONE covariance matrix distributed by
 * block rows, 1-D block-cyclic, with an RCCL broadcast of the panel factor inv(L_kk) and an all-gather of
 * the panel column per step.  They extend, not replace, include/gpx.h (same handle, same conventions).
 */
#ifndef GPX_DIST_H
#define GPX_DIST_H

#include "gpx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Local layout for `rank` of `nranks`; x (n,d) and x_new (m,d) are the full, replicated, already
 * transformed inputs (what jax.device_put(x) of gpax/gp.py:527 moves, on every rank). */
int gpx_dist_configure(gpx_handle* h, int rank, int nranks, const double* x, int64_t n, int32_t d,
                       const double* x_new, int64_t m);
int gpx_dist_set_targets(gpx_handle* h, const double* y_minus_m);
int gpx_dist_info(gpx_handle* h, int64_t* nsteps, int64_t* nloc_rows, int* r_owner, int64_t* n_test_local);

/* RCCL communicator: rank 0 creates the id, every rank calls comm_init with it.  A communicator of ONE rank is
 * allowed: gpx_dist_fit_predict issues every collective whenever a communicator exists (that is how the RCCL
 * calls are exercised on a one-GPU box). */
int gpx_dist_unique_id(char* out128);
int gpx_dist_comm_init(gpx_handle* h, const char* uid128, int rank, int nranks);
int gpx_dist_comm_destroy(gpx_handle* h);

/* Whole distributed fit+predict step (gram of own block rows, factorisation with RCCL broadcast /
 * all-gather per panel, log-marginal, posterior mean / variance); outputs complete on every rank.
 * The all-gather of step k moves max_r(rows rank r still holds below block k) x nb doubles per rank; GPX_DIST_P2P=1 replaces
 * it by direct per-peer ncclSend / ncclRecv pairs in one group (xGMI is point-to-point: 7 concurrent transfers, one per link,
 * instead of a ring bound by a single link). */
int gpx_dist_fit_predict(gpx_handle* h, const gpx_kernel_params* p, const double* m_new, double new_noise,
                         double* logp, double* mean_out, double* var_out);

/* The same step as separate phases + the buffers a transport must move, so that the schedule can be
 * driven (and validated) without RCCL: see gpax_amd/dist.py and tests/test_gpu_dist.py. */
int gpx_dist_build(gpx_handle* h, const gpx_kernel_params* p, double new_noise);
int gpx_dist_step_diag(gpx_handle* h, int64_t k);     /* owner of block k: potrf_diag -> W_k */
int gpx_dist_step_panel(gpx_handle* h, int64_t k);    /* after W_k arrived: own panel rows, packed into `send` */
int gpx_dist_step_update(gpx_handle* h, int64_t k);   /* after the all-gather: own trailing rows (= gather + update_cols of everything) */
/* the two halves of gpx_dist_step_update, for look-ahead schedules: the received chunks into block order (panel
 * buffer k mod 2), then the update of the matrix block columns [cfirst, cfirst+ncol) (ncol < 0: to the right edge)
 * on the main (on_trail = 0) or the bulk stream */
int gpx_dist_step_gather(gpx_handle* h, int64_t k);
int gpx_dist_step_update_cols(gpx_handle* h, int64_t k, int64_t cfirst, int64_t ncol, int on_trail);
/* early-diagonal schedule (GPX_DIST_EARLY_DIAG=1 in gpx_dist_fit_predict): on the owner of block k1, A_{k1,k1} -=
 * L_{k1,k1-1} L_{k1,k1-1}^T from its own panel rows and potrf_diag(k1), on the handle's second chain stream; and the
 * update of block column k+1 that leaves that block row out on its owner */
int gpx_dist_step_diag_early(gpx_handle* h, int64_t k1);
int gpx_dist_step_update_next_col(gpx_handle* h, int64_t k);
/* grouped bulk updates (the default of gpx_dist_fit_predict with the look-ahead; GPX_DIST_GROUP=0 switches them off): the
 * received chunks of panel k into column slot `slot` of the K-contiguous group buffer `buf` (0 | 1), and the update of the
 * matrix block columns [cfirst, cfirst+ncol) by the npan panels from kfirst on in ONE K = npan nb pass; the group size by
 * the block columns still to go (options "pair_min" / "quad_min", as on one GPU) */
int gpx_dist_step_gather_group(gpx_handle* h, int64_t k, int slot, int buf);
int gpx_dist_step_update_group(gpx_handle* h, int64_t kfirst, int64_t npan, int64_t cfirst, int64_t ncol, int on_trail, int buf);
int gpx_dist_group_size(gpx_handle* h, int64_t left);
int gpx_dist_buffers(gpx_handle* h, void** w, int64_t* w_count, void** send, void** recv, int64_t* chunk_count,
                     void** z, int64_t* z_count);
int gpx_dist_copy(gpx_handle* h, void* dst, const void* src, int64_t count);   /* device-to-device, doubles */
int gpx_dist_extract_z(gpx_handle* h);
int gpx_dist_local_results(gpx_handle* h, const double* m_new, double* logdet_part, double* quad,
                           int64_t* test_index, double* mean_loc, double* var_loc, int* info);

#ifdef __cplusplus
}
#endif
#endif /* GPX_DIST_H */
