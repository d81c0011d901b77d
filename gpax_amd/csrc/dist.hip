// dist.hip -- ONE factorisation distributed over the GPUs of a node.
//
// The reference has no distributed code (gpax/state.py:64 "# TODO: multi-gpu support?"); this is the
// second sharding axis of the path: the bordered matrix [K ; k_cross ; r] is distributed by BLOCK ROWS,
// 1-D block-cyclic (global block row i -> rank i % P, local slot i / P), one process per GPU.
// Right-looking step k:
//   owner (k % P)   potrf_diag  A_kk -> L_kk, W_k = inv(L_kk)
//   RCCL            broadcast   W_k (nb x nb)                       <- the panel factor over xGMI
//   every rank      panel       L_ik = A_ik W_k^T for its own rows below k          (MFMA GEMM)
//   RCCL            all-gather  the panel column (each rank needs L_jk of every j as B operand)
//   every rank      update      A_ij -= L_ik L_jk^T for its own rows i, all j in (k, i]   (MFMA GEMM)
// The gram build needs no exchange at all (every rank builds its own block rows from the replicated X).
//
// The step is exposed as separate PHASES (gpx_dist_step_*) plus the device buffers the transport must
// move (gpx_dist_buffers), so that the very same compute path can be driven two ways:
//   * production: gpx_dist_fit_predict() -- this file's loop with ncclBroadcast / ncclAllGather /
//     ncclAllReduce on the handle's stream (one process per GPU, communicator from gpx_dist_comm_init);
//   * validation on ONE GPU: tests/test_gpu_dist.py creates P handles ("virtual ranks") in one process
//     and plays the transport with device-to-device copies (gpx_dist_copy) -- every index map, GEMM
//     and reduction of the distributed algorithm runs for real; only the RCCL calls are not exercised.

#include <rccl/rccl.h>
#include <math.h>
#include <string.h>

#include <algorithm>

#include "gpx_internal.h"

#define GPX_NCCL_CHECK(h, expr)                                                          \
  do {                                                                                   \
    ncclResult_t r__ = (expr);                                                           \
    if (r__ != ncclSuccess) {                                                            \
      gpx_set_error(h, std::string(#expr) + ": " + ncclGetErrorString(r__));             \
      return GPX_ERR_COMM;                                                               \
    }                                                                                    \
  } while (0)

template <typename T>
static int dist_ensure(gpx_handle* h, T** p, int64_t* cap, int64_t count) {
  if (count <= *cap && *p) return GPX_OK;
  if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
  if (count <= 0) count = 1;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T));
  if (e != hipSuccess) { gpx_set_error(h, std::string("hipMalloc failed: ") + hipGetErrorString(e)); *p = nullptr; return GPX_ERR_NOMEM; }
  *cap = count;
  return GPX_OK;
}

// Optionally (GPX_DIST_RESERVED_CUS=n, default 0 = off) the bulk passes of the distributed driver run on a stream whose CU
// mask leaves n CUs (n / 8 per XCD) to the chain.  potrf_diag is ONE workgroup that needs a whole CU (144 KB of LDS, 2 x 208
// VGPRs per SIMD); behind a saturating bulk pass -- three 168-VGPR workgroups per CU, refilled one by one as they finish --
// no CU ever frees up, and the kernel waits the pass out: measured 174 us per potrf_diag on average (2.3 ms at worst) at
// N = 16384 on one rank, 22 of the step's 40 ms.  The single-GPU schedule avoids that wait by starting potrf_diag resident
// a step early.  Measured here (one rank, N = 16384): 39.8 ms without a mask, 41.0 / 40.8 / 38.1 ms with 8 / 16 / 32 CUs
// reserved -- the mask does not buy the chain a CU reliably; the fix on the list is the resident early diagonal.
static hipStream_t bulk_stream(gpx_handle* h) {
  DistState& D = h->dist;
  if (D.s_bulk) return D.s_bulk;
  const char* env = getenv("GPX_DIST_RESERVED_CUS");
  int drop = env ? atoi(env) : 0;
  const int ncu = h->prop.multiProcessorCount;
  if (drop <= 0 || drop >= ncu) return h->s_trail;
  drop = (drop + 7) / 8 * 8;   // mask bit i is CU i / 8 of XCD i % 8: the first `drop` bits take drop / 8 CUs from every XCD
  std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
  for (int i = drop; i < ncu; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
  hipStream_t st = nullptr;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) return h->s_trail;
  D.s_bulk = st;
  return st;
}

// ---------------------------------------------------------------------------------- layout
// global row blocks: [0, nblk) the matrix, [nblk, nblk + eblk) the border (k_cross rows, then r)
// leading dimension of the K-contiguous group buffers: GPX_DIST_GMAX panels side by side (rows exactly 4 KiB apart; 16 or 32
// doubles of padding measured no different: 37.2-37.7 ms per one-rank N = 16384 step either way)
static inline int64_t group_ld(int64_t nb) { return (int64_t)GPX_DIST_GMAX * nb; }
static inline int64_t owner_of(const DistState& d, int64_t i) { return i % d.P; }
static inline int64_t slot_of(const DistState& d, int64_t i) { return i / d.P; }
static inline int64_t nloc_of(const DistState& d, int r) { return (d.RB - r + d.P - 1) / d.P; }
// first local slot of rank r whose global block is > k
static inline int64_t first_slot_after(const DistState& d, int r, int64_t k) { return k >= r ? (k - r) / d.P + 1 : 0; }

// row_type: 0 = kernel row against x (matrix or k_cross), 1 = identity (matrix padding), 2 = zero, 3 = r
__global__ void dist_fixup_kernel(double* A, int64_t ld, int64_t nrows, int64_t n, int64_t npad, const int64_t* gid,
                                  const int* type, const double* r) {
  const int64_t row = blockIdx.y;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nrows || c >= npad) return;
  const int t = type[row];
  double* a = A + row * ld;
  if (t == 0) { if (c >= n) a[c] = 0.0; }                       // kernel rows: only the padded columns
  else if (t == 1) a[c] = (c == gid[row]) ? 1.0 : 0.0;
  else if (t == 2) a[c] = 0.0;
  else a[c] = c < n ? r[c] : 0.0;
}

// panel[(i - k - 1) * nb + t][:] = recv[i % P][(i / P - first_slot_after(i % P, k)) * nb + t][:]
__global__ void dist_unpermute_kernel(double* panel, const double* recv, int64_t k, int64_t RB, int P, int nb,
                                      int64_t chunk_rows) {
  const int64_t i = k + 1 + blockIdx.x;   // global block
  if (i >= RB) return;
  const int r = (int)(i % P);
  const int64_t s0 = k >= r ? (k - r) / P + 1 : 0;
  const double* src = recv + ((int64_t)r * chunk_rows + (i / P - s0) * nb) * nb;
  double* dst = panel + (int64_t)blockIdx.x * nb * nb;
  // 16 B per lane, the block split over gridDim.y workgroups (one workgroup copying 128 KB eight bytes at a time took 31 us
  // per launch beside the bulk update: profiles/r03_kernel_stats_dist_one_rank.csv)
  const int per = nb * nb / 2 / (int)gridDim.y;
  const v2d* s2 = reinterpret_cast<const v2d*>(src) + (int64_t)blockIdx.y * per;
  v2d* d2 = reinterpret_cast<v2d*>(dst) + (int64_t)blockIdx.y * per;
  for (int e = threadIdx.x; e < per; e += blockDim.x) d2[e] = s2[e];
}

// the same chunks into a group buffer: block row i (absolute) -> rows [i nb, (i+1) nb), columns [slot nb, (slot+1) nb) of G (ld = ldg)
__global__ void dist_unpermute_group_kernel(double* G, int64_t ldg, int slot, const double* recv, int64_t k, int64_t RB, int P, int nb,
                                            int64_t chunk_rows) {
  const int64_t i = k + 1 + blockIdx.x;   // global block
  if (i >= RB) return;
  const int r = (int)(i % P);
  const int64_t s0 = k >= r ? (k - r) / P + 1 : 0;
  const double* src = recv + ((int64_t)r * chunk_rows + (i / P - s0) * nb) * nb;
  double* dst = G + i * nb * ldg + (int64_t)slot * nb;
  const int rows_per = nb / (int)gridDim.y, hw = nb / 2;   // rows of the block per workgroup; 16-B pieces per row
  const int r0 = (int)blockIdx.y * rows_per;
  for (int e = threadIdx.x; e < rows_per * hw; e += blockDim.x) {
    const int rr = r0 + e / hw, c2 = e % hw;
    *reinterpret_cast<v2d*>(dst + (int64_t)rr * ldg + 2 * c2) = *reinterpret_cast<const v2d*>(src + (int64_t)rr * nb + 2 * c2);
  }
}

__global__ __launch_bounds__(256) void dist_logdet_kernel(const double* L, int64_t ld, int64_t nslots, int P, int rank,
                                                         int nb, int64_t n, double* out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (int64_t q = threadIdx.x; q < nslots * nb; q += 256) {
    const int64_t slot = q / nb, t = q % nb;
    const int64_t g = (slot * P + rank) * nb + t;   // global row == diagonal column
    if (g < n) s += log(L[(slot * nb + t) * ld + g]);
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = sh[0];
}

// ------------------------------------------------------------------------------ configure

extern "C" int gpx_dist_configure(gpx_handle* h, int rank, int nranks, const double* x, int64_t n, int32_t d,
                                  const double* x_new, int64_t m) {
  if (!h) return GPX_ERR_ARG;
  if (!x || n <= 0 || d <= 0 || nranks <= 0 || rank < 0 || rank >= nranks || m < 0 || (m > 0 && !x_new)) {
    gpx_set_error(h, "gpx_dist_configure: bad argument"); return GPX_ERR_ARG;
  }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  DistState& D = h->dist;
  const int nb = h->nb;
  D.rank = rank; D.P = nranks; D.n = n; D.m = m; D.d = d;
  D.npad = round_up(n, nb);
  D.nblk = D.npad / nb;
  D.eblk = (m + 1 + nb - 1) / nb;          // k_cross rows + the r row
  D.RB = D.nblk + D.eblk;
  D.nloc = nloc_of(D, rank);
  D.ld = D.npad;
  const int64_t rows = D.nloc * nb;
  int64_t maxloc = 0;
  for (int r = 0; r < nranks; ++r) maxloc = std::max(maxloc, nloc_of(D, r));
  D.chunk_rows = maxloc * nb;
  GPX_RET(dist_ensure(h, &D.A, &D.a_cap, rows * D.ld));
  GPX_RET(dist_ensure(h, &D.L, &D.l_cap, rows * D.ld));
  GPX_RET(dist_ensure(h, &D.W, &D.w_cap, (int64_t)nb * nb));
  GPX_RET(dist_ensure(h, &D.send, &D.send_cap, D.chunk_rows * nb));
  GPX_RET(dist_ensure(h, &D.recv, &D.recv_cap, (int64_t)nranks * D.chunk_rows * nb));
  GPX_RET(dist_ensure(h, &D.panel, &D.panel_cap, D.RB * nb * nb));
  GPX_RET(dist_ensure(h, &D.panel2, &D.panel2_cap, D.RB * nb * nb));
  // grouped bulk updates: two K-contiguous group buffers, (RB nb) rows x (GPX_DIST_GMAX nb) columns each -- the gathered
  // panels of a group side by side, indexed by ABSOLUTE block row, so that one K = g nb pass can apply them all
  GPX_RET(dist_ensure(h, &D.group[0], &D.group_cap[0], D.RB * nb * group_ld(nb)));
  GPX_RET(dist_ensure(h, &D.group[1], &D.group_cap[1], D.RB * nb * group_ld(nb)));
  GPX_RET(dist_ensure(h, &D.xloc, &D.xloc_cap, rows * d));
  GPX_RET(dist_ensure(h, &D.gid, &D.gid_cap, rows));
  GPX_RET(dist_ensure(h, &D.type, &D.type_cap, rows));
  GPX_RET(dist_ensure(h, &D.r, &D.r_cap, n));
  GPX_RET(dist_ensure(h, &D.z, &D.z_cap, D.npad));
  GPX_RET(dist_ensure(h, &D.small, &D.small_cap, 4 + 2 * rows));
  // replicated training inputs (the gram build needs no exchange)
  GPX_RET(gpx_set_train(h, x, n, d));
  // local x1 rows, global ids and row types
  std::vector<double> xl((size_t)(rows * d), 0.0);
  std::vector<int64_t> gid((size_t)rows);
  std::vector<int> type((size_t)rows);
  D.test_rows.clear();
  for (int64_t s = 0; s < D.nloc; ++s) {
    const int64_t i = s * nranks + rank;
    for (int t = 0; t < nb; ++t) {
      const int64_t lr = s * nb + t;
      if (i < D.nblk) {
        const int64_t g = i * nb + t;
        gid[(size_t)lr] = g;
        if (g < n) { type[(size_t)lr] = 0; memcpy(&xl[(size_t)(lr * d)], x + g * d, sizeof(double) * d); }
        else type[(size_t)lr] = 1;
      } else {
        const int64_t e = (i - D.nblk) * nb + t;   // border row index
        gid[(size_t)lr] = -1;                      // never on the diagonal
        if (e < m) { type[(size_t)lr] = 0; memcpy(&xl[(size_t)(lr * d)], x_new + e * d, sizeof(double) * d); D.test_rows.push_back({lr, e}); }
        else if (e == m) { type[(size_t)lr] = 3; D.r_local_row = lr; }
        else type[(size_t)lr] = 2;
      }
    }
  }
  D.r_owner = (int)owner_of(D, D.nblk + m / nb);
  if (D.r_owner != rank) D.r_local_row = -1;
  hipStream_t s = h->s_main;
  GPX_HIP_CHECK(h, hipMemcpyAsync(D.xloc, xl.data(), xl.size() * sizeof(double), hipMemcpyHostToDevice, s));
  GPX_HIP_CHECK(h, hipMemcpyAsync(D.gid, gid.data(), gid.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
  GPX_HIP_CHECK(h, hipMemcpyAsync(D.type, type.data(), type.size() * sizeof(int), hipMemcpyHostToDevice, s));
  GPX_HIP_CHECK(h, hipStreamSynchronize(s));
  D.configured = true;
  D.have_r = false;
  return GPX_OK;
}

extern "C" int gpx_dist_set_targets(gpx_handle* h, const double* y_minus_m) {
  if (!h || !h->dist.configured || !y_minus_m) { if (h) gpx_set_error(h, "gpx_dist_set_targets: configure first"); return GPX_ERR_STATE; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  GPX_HIP_CHECK(h, hipMemcpyAsync(h->dist.r, y_minus_m, (size_t)h->dist.n * sizeof(double), hipMemcpyHostToDevice, h->s_main));
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_main));
  h->dist.have_r = true;
  return GPX_OK;
}

extern "C" int gpx_dist_info(gpx_handle* h, int64_t* nsteps, int64_t* nloc_rows, int* r_owner, int64_t* n_test_local) {
  if (!h || !h->dist.configured) return GPX_ERR_STATE;
  if (nsteps) *nsteps = h->dist.nblk;
  if (nloc_rows) *nloc_rows = h->dist.nloc * h->nb;
  if (r_owner) *r_owner = h->dist.r_owner;
  if (n_test_local) *n_test_local = (int64_t)h->dist.test_rows.size();
  return GPX_OK;
}

// ----------------------------------------------------------------------------------- build

// hyper-parameter upload is shared with the single-GPU path (api.hip)
int gpx_upload_params(gpx_handle* h, hipStream_t s, const gpx_kernel_params* p, int64_t n, int d,
                      const double** noise_dev, double* noise_scalar, double new_noise);

extern "C" int gpx_dist_build(gpx_handle* h, const gpx_kernel_params* p, double new_noise) {
  if (!h) return GPX_ERR_ARG;
  DistState& D = h->dist;
  if (!D.configured || !D.have_r) { gpx_set_error(h, "gpx_dist_build: configure and set targets first"); return GPX_ERR_STATE; }
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  hipStream_t s = h->s_main;
  const int nb = h->nb;
  const int64_t rows = D.nloc * nb;
  const double* noise_dev; double noise_scalar;
  GPX_RET(gpx_upload_params(h, s, p, D.n, D.d, &noise_dev, &noise_scalar, new_noise));
  D.kind = p->kind; D.k_scale = p->k_scale; D.k_jitter = p->k_jitter;
  GramLaunch g;
  g.kind = p->kind; g.d = D.d; g.k_scale = p->k_scale; g.hyp = h->d_hyp;
  g.x1 = D.xloc; g.x2 = h->d_x; g.n1 = rows; g.n2 = D.n; g.out = D.A; g.ld = D.ld;
  g.lower_only = 0; g.diag_rule = p->apply_noise ? 1 : 0;
  g.noise = noise_dev; g.noise_scalar = noise_scalar; g.jitter = p->k_jitter;
  g.row_gid = D.gid;
  if (rows > 0) {   // a rank may hold no block row at all (more ranks than block rows)
    GPX_RET(launch_gram(h, s, g));
    dim3 grid((unsigned)((D.npad + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL(dist_fixup_kernel, grid, dim3(256), 0, s, D.A, D.ld, rows, D.n, D.npad, D.gid, D.type, D.r);
    GPX_HIP_CHECK(h, hipGetLastError());
  }
  GPX_HIP_CHECK(h, hipMemsetAsync(h->d_info, 0, sizeof(int), s));
  return GPX_OK;
}

// ------------------------------------------------------------------------------ step phases

extern "C" int gpx_dist_step_diag(gpx_handle* h, int64_t k) {
  DistState& D = h->dist;
  if (owner_of(D, k) != D.rank) return GPX_OK;
  const int nb = h->nb;
  const int64_t s = slot_of(D, k), c0 = k * nb;
  PotrfLaunch p;
  p.nb = nb; p.A = D.A + s * nb * D.ld + c0; p.lda = D.ld; p.L = D.L + s * nb * D.ld + c0; p.ldl = D.ld; p.W = D.W;
  p.info = h->d_info; p.col0 = (int)c0;
  return launch_potrf_diag(h, h->s_main, p);
}

extern "C" int gpx_dist_step_panel(gpx_handle* h, int64_t k) {
  DistState& D = h->dist;
  const int nb = h->nb;
  const int64_t s0 = first_slot_after(D, D.rank, k), c0 = k * nb;
  const int64_t mrows = (D.nloc - s0) * nb;
  if (mrows <= 0) return GPX_OK;
  GemmLaunch g;
  g.A = D.A + s0 * nb * D.ld + c0; g.lda = D.ld;
  g.B = D.W; g.ldb = nb;
  g.C = D.L + s0 * nb * D.ld + c0; g.ldc = D.ld;
  g.M = mrows; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0;
  g.lower = 0; g.row_off = 0; g.col_off = 0; g.b_lower_tri = 1; g.small_tile = 1;
  GPX_RET(launch_gemm_nt(h, h->s_main, g));
  // pack this rank's panel rows densely for the all-gather  (writing them from the panel kernel's own tiles as a second
  // output instead was measured in round 3: 36.6-37.0 vs 36.4 ms per one-rank N = 16384 step -- not kept)
  GPX_HIP_CHECK(h, hipMemcpy2DAsync(D.send, (size_t)nb * sizeof(double), D.L + s0 * nb * D.ld + c0, (size_t)D.ld * sizeof(double),
                                    (size_t)nb * sizeof(double), (size_t)mrows, hipMemcpyDeviceToDevice, h->s_main));
  return GPX_OK;
}

static inline double* panel_buf(DistState& D, int64_t k) { return (k & 1) ? D.panel2 : D.panel; }

// after the all-gather: the received chunks (rank-major) into global block order, panel buffer k mod 2
extern "C" int gpx_dist_step_gather(gpx_handle* h, int64_t k) {
  DistState& D = h->dist;
  const int64_t nbelow = D.RB - k - 1;
  if (nbelow <= 0) return GPX_OK;
  hipLaunchKernelGGL(dist_unpermute_kernel, dim3((unsigned)nbelow, 4u), dim3(256), 0, h->s_main, panel_buf(D, k), D.recv, k, D.RB,
                     D.P, h->nb, D.gather_stride_rows > 0 ? D.gather_stride_rows : D.chunk_rows);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

// own trailing rows, matrix block columns [cfirst, cfirst + ncol) -= panel k (ncol < 0: all the way to the right);
// on_trail != 0 enqueues on the bulk stream (the look-ahead driver), else on the main stream
// rows: this rank's block rows AFTER global block `after` (>= k): `after` = k is the plain update; the early-diagonal
// schedule passes k+1 for block column k+1, whose diagonal block row the owner has already updated and factorised
static int dist_update_cols(gpx_handle* h, int64_t k, int64_t cfirst, int64_t ncol, int on_trail, int64_t after) {
  DistState& D = h->dist;
  const int nb = h->nb;
  if (cfirst < k + 1) cfirst = k + 1;
  if (ncol < 0 || cfirst + ncol > D.nblk) ncol = D.nblk - cfirst;
  const int64_t s0 = first_slot_after(D, D.rank, after), c0 = k * nb;
  const int64_t mrows = (D.nloc - s0) * nb;
  if (mrows <= 0 || ncol <= 0) return GPX_OK;
  GemmLaunch t;
  t.A = D.L + s0 * nb * D.ld + c0; t.lda = D.ld;
  t.B = panel_buf(D, k) + (cfirst - (k + 1)) * nb * nb; t.ldb = nb;   // rows of global blocks cfirst .. (B rows = matrix columns)
  t.C = D.A + s0 * nb * D.ld + cfirst * nb; t.ldc = D.ld;
  t.M = mrows; t.N = ncol * nb; t.K = nb; t.alpha = -1.0; t.beta = 1.0;
  t.lower = 1; t.row_off = s0 * nb; t.col_off = cfirst * nb; t.b_lower_tri = 0; t.small_tile = 0;
  t.cyc_P = D.P; t.cyc_r = D.rank; t.cyc_nb = nb;
  return launch_gemm_nt(h, on_trail ? bulk_stream(h) : h->s_main, t);
}

extern "C" int gpx_dist_step_update_cols(gpx_handle* h, int64_t k, int64_t cfirst, int64_t ncol, int on_trail) {
  return dist_update_cols(h, k, cfirst, ncol, on_trail, k);
}

// Early-diagonal schedule: block column k+1 without the block row k+1 on its owner (that block is gpx_dist_step_diag_early's)
extern "C" int gpx_dist_step_update_next_col(gpx_handle* h, int64_t k) {
  return dist_update_cols(h, k, k + 1, 1, 0, k + 1);
}

// Early-diagonal schedule, owner of block k1 = k+1 only, on the second chain stream: A_{k1,k1} -= L_{k1,k} L_{k1,k}^T with
// the owner's OWN panel rows (no other rank's data: it need not wait for the all-gather of panel k), then potrf_diag
// -> L_{k1,k1}, W_{k1}.  The caller orders it behind the panel solve of step k and the bulk update that last wrote
// the block, and orders the broadcast of W_{k1} behind it.
extern "C" int gpx_dist_step_diag_early(gpx_handle* h, int64_t k1) {
  DistState& D = h->dist;
  if (k1 < 1 || k1 >= D.nblk || owner_of(D, k1) != D.rank) return GPX_OK;
  const int nb = h->nb;
  const int64_t sl = slot_of(D, k1), c0 = (k1 - 1) * nb, c1 = k1 * nb;
  GemmLaunch u;
  u.A = D.L + sl * nb * D.ld + c0; u.lda = D.ld;
  u.B = D.L + sl * nb * D.ld + c0; u.ldb = D.ld;
  u.C = D.A + sl * nb * D.ld + c1; u.ldc = D.ld;
  u.M = nb; u.N = nb; u.K = nb; u.alpha = -1.0; u.beta = 1.0;
  u.lower = 1; u.row_off = 0; u.col_off = 0; u.b_lower_tri = 0; u.small_tile = 1;
  GPX_RET(launch_gemm_nt(h, h->s_main2, u));
  PotrfLaunch p;
  p.nb = nb; p.A = D.A + sl * nb * D.ld + c1; p.lda = D.ld; p.L = D.L + sl * nb * D.ld + c1; p.ldl = D.ld; p.W = D.W;
  p.info = h->d_info; p.col0 = (int)c1;
  return launch_potrf_diag(h, h->s_main2, p);
}

// Grouped schedule: the received chunks of panel k into group buffer `buf` (0 | 1), column slot `slot` (= k - first panel of the group)
extern "C" int gpx_dist_step_gather_group(gpx_handle* h, int64_t k, int slot, int buf) {
  DistState& D = h->dist;
  if (slot < 0 || slot >= GPX_DIST_GMAX || (buf != 0 && buf != 1)) { gpx_set_error(h, "gpx_dist_step_gather_group: bad slot / buffer"); return GPX_ERR_ARG; }
  const int64_t nbelow = D.RB - k - 1;
  if (nbelow <= 0) return GPX_OK;
  hipLaunchKernelGGL(dist_unpermute_group_kernel, dim3((unsigned)nbelow, 4u), dim3(256), 0, h->s_main, D.group[buf], group_ld(h->nb),
                     slot, D.recv, k, D.RB, D.P, h->nb, D.gather_stride_rows > 0 ? D.gather_stride_rows : D.chunk_rows);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

// Grouped schedule: own block rows after global block k = kfirst + npan - 1, matrix block columns [cfirst, cfirst + ncol)
// (ncol < 0: to the right edge) -= panels kfirst .. k in ONE K = npan nb pass against group buffer `buf`
extern "C" int gpx_dist_step_update_group(gpx_handle* h, int64_t kfirst, int64_t npan, int64_t cfirst, int64_t ncol, int on_trail, int buf) {
  DistState& D = h->dist;
  const int nb = h->nb;
  if (npan < 1 || npan > GPX_DIST_GMAX || (buf != 0 && buf != 1)) { gpx_set_error(h, "gpx_dist_step_update_group: bad group"); return GPX_ERR_ARG; }
  const int64_t k = kfirst + npan - 1;
  if (cfirst < k + 1) cfirst = k + 1;
  if (ncol < 0 || cfirst + ncol > D.nblk) ncol = D.nblk - cfirst;
  // own rows from global block cfirst on: the blocks between k+1 and cfirst have no lower tile in these columns (and on one
  // rank the launch is then the square-topped trapezoid that the 1-D tile enumeration serves without empty workgroups)
  const int64_t s0 = first_slot_after(D, D.rank, cfirst - 1);
  const int64_t mrows = (D.nloc - s0) * nb, ldg = group_ld(nb);
  if (mrows <= 0 || ncol <= 0) return GPX_OK;
  GemmLaunch t;
  t.A = D.L + s0 * nb * D.ld + kfirst * nb; t.lda = D.ld;          // own rows: the group's panels are adjacent columns of L
  t.B = D.group[buf] + cfirst * nb * ldg; t.ldb = ldg;              // rows of global blocks cfirst .. (B rows = matrix columns)
  t.C = D.A + s0 * nb * D.ld + cfirst * nb; t.ldc = D.ld;
  t.M = mrows; t.N = ncol * nb; t.K = npan * nb; t.alpha = -1.0; t.beta = 1.0;
  t.lower = 1; t.row_off = s0 * nb; t.col_off = cfirst * nb; t.b_lower_tri = 0; t.small_tile = 0;
  t.cyc_P = D.P; t.cyc_r = D.rank; t.cyc_nb = nb;
  return launch_gemm_nt(h, on_trail ? bulk_stream(h) : h->s_main, t);
}

// group size by the block columns still to go: the single-GPU thresholds ("quad_min" / "pair_min")
extern "C" int gpx_dist_group_size(gpx_handle* h, int64_t left) {
  if (!h) return 1;
  return (h->quad_min > 0 && left > h->quad_min) ? 4 : ((h->pair_min > 0 && left > h->pair_min) ? 2 : 1);
}

extern "C" int gpx_dist_step_update(gpx_handle* h, int64_t k) {
  GPX_RET(gpx_dist_step_gather(h, k));
  return gpx_dist_step_update_cols(h, k, k + 1, -1, 0);
}

extern "C" int gpx_dist_buffers(gpx_handle* h, void** w, int64_t* w_count, void** send, void** recv, int64_t* chunk_count,
                                void** z, int64_t* z_count) {
  if (!h || !h->dist.configured) return GPX_ERR_STATE;
  DistState& D = h->dist;
  if (w) *w = D.W;
  if (w_count) *w_count = (int64_t)h->nb * h->nb;
  if (send) *send = D.send;
  if (recv) *recv = D.recv;
  if (chunk_count) *chunk_count = D.chunk_rows * h->nb;
  if (z) *z = D.z;
  if (z_count) *z_count = D.npad;
  return GPX_OK;
}

// test transport: device-to-device copy between buffers of handles living in one process
extern "C" int gpx_dist_copy(gpx_handle* h, void* dst, const void* src, int64_t count) {
  if (!h) return GPX_ERR_ARG;
  GPX_HIP_CHECK(h, hipMemcpyAsync(dst, src, (size_t)count * sizeof(double), hipMemcpyDeviceToDevice, h->s_main));
  GPX_HIP_CHECK(h, hipStreamSynchronize(h->s_main));
  return GPX_OK;
}

// ------------------------------------------------------------------------------- epilogue

// After the last step: the owner of the r row copies z = L^-1 r into its D.z (others receive it by the
// transport); then every rank reduces its own part.
extern "C" int gpx_dist_extract_z(gpx_handle* h) {
  DistState& D = h->dist;
  if (D.r_local_row < 0) return GPX_OK;
  GPX_HIP_CHECK(h, hipMemcpyAsync(D.z, D.L + D.r_local_row * D.ld, (size_t)D.npad * sizeof(double), hipMemcpyDeviceToDevice, h->s_main));
  return GPX_OK;
}

// local contributions: logdet_part (sum of log L_ii over owned diagonal blocks), quad = |z|^2 (same on all
// ranks once z is replicated), and mean/var of the locally held test rows (test_index[] says which).
extern "C" int gpx_dist_local_results(gpx_handle* h, const double* m_new, double* logdet_part, double* quad,
                                      int64_t* test_index, double* mean_loc, double* var_loc, int* info) {
  if (!h || !h->dist.configured) return GPX_ERR_STATE;
  DistState& D = h->dist;
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  hipStream_t s = h->s_main;
  const int nb = h->nb;
  hipLaunchKernelGGL(dist_logdet_kernel, dim3(1), dim3(256), 0, s, D.L, D.ld, std::min<int64_t>(D.nloc, (D.nblk - D.rank + D.P - 1) / D.P),
                     D.P, D.rank, nb, D.n, D.small);
  GPX_HIP_CHECK(h, hipGetLastError());
  // quad and the per-row reductions reuse the single-GPU epilogue kernels on the local rows
  const int64_t nt = (int64_t)D.test_rows.size();
  std::vector<double> zh((size_t)D.n);
  GPX_HIP_CHECK(h, hipMemcpyAsync(zh.data(), D.z, (size_t)D.n * sizeof(double), hipMemcpyDeviceToHost, s));
  double ld_part = 0.0; int inf = 0;
  GPX_HIP_CHECK(h, hipMemcpyAsync(&ld_part, D.small, sizeof(double), hipMemcpyDeviceToHost, s));
  GPX_HIP_CHECK(h, hipMemcpyAsync(&inf, h->d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  // V^T rows of this rank: first local test row is contiguous after the matrix slots of this rank
  if (nt > 0) {
    const int64_t lr0 = D.test_rows.front().first;
    bool contiguous = true;
    for (int64_t q = 0; q < nt; ++q) if (D.test_rows[(size_t)q].first != lr0 + q) contiguous = false;
    if (!contiguous) { gpx_set_error(h, "internal: local test rows not contiguous"); return GPX_ERR_STATE; }
    EpilogueLaunch pe;
    pe.L = D.L; pe.ld = D.ld; pe.n = D.n; pe.z = D.z; pe.vt = D.L + lr0 * D.ld; pe.vt_ld = D.ld; pe.m = nt; pe.m_new = nullptr;
    pe.prior_diag = 0.0; pe.dev_prior_diag = h->d_hyp + 2 * D.d + 3; pe.scal = D.small + 2;
    pe.mean = D.small + 4; pe.var = D.small + 4 + nt;
    GPX_RET(launch_post_reduce(h, s, pe));
    GPX_HIP_CHECK(h, hipMemcpyAsync(mean_loc, D.small + 4, (size_t)nt * sizeof(double), hipMemcpyDeviceToHost, s));
    GPX_HIP_CHECK(h, hipMemcpyAsync(var_loc, D.small + 4 + nt, (size_t)nt * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  GPX_HIP_CHECK(h, hipStreamSynchronize(s));
  double q = 0.0;
  for (int64_t c = 0; c < D.n; ++c) q += zh[(size_t)c] * zh[(size_t)c];
  if (logdet_part) *logdet_part = ld_part;
  if (quad) *quad = q;
  if (info) *info = inf;
  for (int64_t t = 0; t < nt; ++t) {
    const int64_t e_idx = D.test_rows[(size_t)t].second;
    if (test_index) test_index[t] = e_idx;
    if (mean_loc && m_new) mean_loc[t] += m_new[e_idx];
  }
  return GPX_OK;
}

// ------------------------------------------------------------------- RCCL production driver

extern "C" int gpx_dist_unique_id(char* out128) {
  if (!out128) return GPX_ERR_ARG;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return GPX_ERR_COMM;
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return GPX_OK;
}

extern "C" int gpx_dist_comm_init(gpx_handle* h, const char* uid128, int rank, int nranks) {
  if (!h || !uid128) return GPX_ERR_ARG;
  GPX_HIP_CHECK(h, hipSetDevice(h->device));
  if (h->dist.comm) { (void)ncclCommDestroy((ncclComm_t)h->dist.comm); h->dist.comm = nullptr; }
  ncclUniqueId id;
  memcpy(id.internal, uid128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm;
  GPX_NCCL_CHECK(h, ncclCommInitRank(&comm, nranks, id, rank));
  h->dist.comm = comm;
  return GPX_OK;
}

extern "C" int gpx_dist_comm_destroy(gpx_handle* h) {
  if (!h) return GPX_ERR_ARG;
  if (h->dist.comm) { (void)ncclCommDestroy((ncclComm_t)h->dist.comm); h->dist.comm = nullptr; }
  return GPX_OK;
}

// One process per GPU: build + factorise + epilogue with RCCL over xGMI.  Outputs are complete on
// every rank.  (Not exercised on the one-GPU test box beyond nranks = 1; the phases it strings
// together are the ones tests/test_gpu_dist.py validates with P virtual ranks.)
extern "C" int gpx_dist_fit_predict(gpx_handle* h, const gpx_kernel_params* p, const double* m_new, double new_noise,
                                    double* logp, double* mean_out, double* var_out) {
  if (!h) return GPX_ERR_ARG;
  DistState& D = h->dist;
  if (!D.configured) { gpx_set_error(h, "gpx_dist_fit_predict: configure first"); return GPX_ERR_STATE; }
  if (D.P > 1 && !D.comm) { gpx_set_error(h, "gpx_dist_fit_predict: no communicator (gpx_dist_comm_init)"); return GPX_ERR_STATE; }
  // Every collective below runs whenever a communicator exists -- also a single-rank one (tests/test_gpu_dist.py drives
  // the RCCL calls, their stream ordering and the per-step counts that way on the one-GPU box).
  ncclComm_t comm = (ncclComm_t)D.comm;
  hipStream_t s = h->s_main;
  const int nb = h->nb;
  GPX_RET(gpx_dist_build(h, p, new_noise));
  // Look-ahead (depth 1): the main stream carries factor -> broadcast -> panel -> all-gather -> update of block column
  // k+1 only; the bulk update of the columns >= k+2 runs on the second stream while the NEXT step's diagonal block,
  // broadcast, panel solve and all-gather are already under way -- the collectives' latency (the per-step cost that
  // does not shrink with the rank count) hides behind the bulk update.  Column k+2 goes back to the main stream one
  // step later, so the only wait on the critical path is for the PREVIOUS step's bulk update.
  const char* la_env = getenv("GPX_DIST_LOOKAHEAD");   // read per call (tests switch it)
  const int dist_la = la_env ? atoi(la_env) : 1;
  hipStream_t t = bulk_stream(h);
  while ((int64_t)D.ev_s.size() < D.nblk) {
    hipEvent_t a, b;
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&a, hipEventDisableTiming | hipEventDisableSystemFence));
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&b, hipEventDisableTiming | hipEventDisableSystemFence));
    D.ev_s.push_back(a); D.ev_t.push_back(b);
  }
  // Early diagonal (GPX_DIST_EARLY_DIAG=1, with the look-ahead): the owner of block k+1 updates and factorises its
  // diagonal block on a second stream as soon as its own panel rows of step k are solved -- it needs nobody else's --
  // so potrf_diag(k+1) (~30 us) runs beside the all-gather of panel k instead of behind it, and the broadcast of
  // W_{k+1} only waits for the later of the two.  The collectives stay on ONE stream in ONE order on every rank.
  const char* ed_env = getenv("GPX_DIST_EARLY_DIAG");
  const bool early = dist_la && ed_env && atoi(ed_env) != 0;
  hipStream_t e = h->s_main2;
  while (early && (int64_t)D.ev_pan.size() < D.nblk) {
    hipEvent_t a, b;
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&a, hipEventDisableTiming | hipEventDisableSystemFence));
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&b, hipEventDisableTiming | hipEventDisableSystemFence));
    D.ev_pan.push_back(a); D.ev_diag.push_back(b);
  }
  // Grouped bulk updates (with the look-ahead, not with the early diagonal): as in the single-GPU schedule the bulk stream
  // applies the g most recent panels in ONE K = g nb pass every g-th step (g = 4 / 2 / 1 by the block columns still to go),
  // which cuts the read-modify-write passes over the trailing rows g-fold; the gathered panels of a group sit side by
  // side in a K-contiguous group buffer, and the main stream brings block column k+1 up to date with the panels since
  // the last bulk pass at every step.  Two group buffers alternate: the bulk pass of one group reads its buffer while
  // the next group's panels arrive.
  const char* gr_env = getenv("GPX_DIST_GROUP");
  const bool grouped = dist_la && !early && !(gr_env && atoi(gr_env) == 0);
  const char* p2p_env = getenv("GPX_DIST_P2P");
  const bool p2p = p2p_env && atoi(p2p_env) != 0;
  int64_t last_t = -1, group_end = -1, prev_t = -1; int gbuf = 0;
  for (int64_t k = 0; k < D.nblk; ++k) {
    if (!early || k == 0) GPX_RET(gpx_dist_step_diag(h, k));
    else if (owner_of(D, k) == D.rank) GPX_HIP_CHECK(h, hipStreamWaitEvent(s, D.ev_diag[(size_t)k], 0));   // potrf_diag(k) ran early on `e`
    if (comm) GPX_NCCL_CHECK(h, ncclBroadcast(D.W, D.W, (size_t)nb * nb, ncclDouble, (int)owner_of(D, k), comm, s));
    GPX_RET(gpx_dist_step_panel(h, k));
    if (early && k + 1 < D.nblk && owner_of(D, k + 1) == D.rank) {
      GPX_HIP_CHECK(h, hipEventRecord(D.ev_pan[(size_t)k], s));
      GPX_HIP_CHECK(h, hipStreamWaitEvent(e, D.ev_pan[(size_t)k], 0));
      if (k >= 1) GPX_HIP_CHECK(h, hipStreamWaitEvent(e, D.ev_t[(size_t)(k - 1)], 0));   // block (k+1,k+1) was last written by the bulk update of step k-1
      GPX_RET(gpx_dist_step_diag_early(h, k + 1));
      GPX_HIP_CHECK(h, hipEventRecord(D.ev_diag[(size_t)(k + 1)], e));
    }
    // The all-gather moves only what is left below block k: the count is the largest number of panel rows any rank
    // still holds (every rank computes the same value), not the full-height chunk of step 0 -- half the bytes overall.
    int64_t step_rows = 0;
    for (int r = 0; r < D.P; ++r) step_rows = std::max(step_rows, (nloc_of(D, r) - first_slot_after(D, r, k)) * nb);
    D.gather_stride_rows = step_rows;
    if (step_rows > 0) {
      const size_t cnt = (size_t)(step_rows * nb);
      if (comm && p2p && D.P > 1) {
        // xGMI is point-to-point (7 links per GPU): instead of a ring all-gather, which is bound by ONE link, every rank
        // sends its chunk to every peer directly -- 7 concurrent transfers, one per link (GPX_DIST_P2P=1)
        GPX_NCCL_CHECK(h, ncclGroupStart());
        for (int peer = 0; peer < D.P; ++peer) {
          if (peer == D.rank) continue;
          GPX_NCCL_CHECK(h, ncclSend(D.send, cnt, ncclDouble, peer, comm, s));
          GPX_NCCL_CHECK(h, ncclRecv(D.recv + (size_t)peer * cnt, cnt, ncclDouble, peer, comm, s));
        }
        GPX_NCCL_CHECK(h, ncclGroupEnd());
        GPX_HIP_CHECK(h, hipMemcpyAsync(D.recv + (size_t)D.rank * cnt, D.send, cnt * sizeof(double), hipMemcpyDeviceToDevice, s));
      } else if (comm) {
        GPX_NCCL_CHECK(h, ncclAllGather(D.send, D.recv, cnt, ncclDouble, comm, s));
      } else {
        GPX_HIP_CHECK(h, hipMemcpyAsync(D.recv, D.send, cnt * sizeof(double), hipMemcpyDeviceToDevice, s));
      }
    }
    if (!dist_la) { GPX_RET(gpx_dist_step_update(h, k)); D.gather_stride_rows = 0; continue; }
    if (grouped) {
      if (k > group_end) group_end = std::min<int64_t>(D.nblk - 1, k + gpx_dist_group_size(h, D.nblk - k) - 1);
      const int64_t kfirst = last_t + 1, npan = k - last_t;
      GPX_RET(gpx_dist_step_gather_group(h, k, (int)(k - kfirst), gbuf));
      D.gather_stride_rows = 0;
      // column k+1 gets every panel it still lacks; it lies in the near block of the previous group end (below), so the
      // chain never waits for a bulk pass inside a group
      GPX_RET(gpx_dist_step_update_group(h, kfirst, npan, k + 1, 1, 0, gbuf));
      if (k == group_end) {
        // Near block: the NEXT group's columns (k+2 .. k+1+g_next) get this group's panels from the main stream, in one
        // K = npan nb launch; the bulk pass starts behind them.  With the bulk pass covering everything from column k+2
        // the chain would stall at the first step of every group until that pass had finished, and the bulk stream would
        // then idle for the g chain steps that produce its next panels (~25 % of the time at g = 4).
        const int64_t g_next = k + 1 < D.nblk ? gpx_dist_group_size(h, D.nblk - (k + 1)) : 0;
        const int64_t near = std::max<int64_t>(0, std::min<int64_t>(g_next, D.nblk - (k + 2)));
        if (near > 0) {
          if (prev_t >= 0) GPX_HIP_CHECK(h, hipStreamWaitEvent(s, D.ev_t[(size_t)prev_t], 0));   // those columns were last written by the previous bulk pass
          GPX_RET(gpx_dist_step_update_group(h, kfirst, npan, k + 2, near, 0, gbuf));
        }
        GPX_HIP_CHECK(h, hipEventRecord(D.ev_s[(size_t)k], s));
        GPX_HIP_CHECK(h, hipStreamWaitEvent(t, D.ev_s[(size_t)k], 0));
        GPX_RET(gpx_dist_step_update_group(h, kfirst, npan, k + 2 + near, -1, 1, gbuf));                    // the bulk pass, K = npan nb
        GPX_HIP_CHECK(h, hipEventRecord(D.ev_t[(size_t)k], t));
        prev_t = k;
        last_t = k; gbuf ^= 1;
      }
      continue;
    }
    GPX_RET(gpx_dist_step_gather(h, k));
    D.gather_stride_rows = 0;
    if (k >= 1) GPX_HIP_CHECK(h, hipStreamWaitEvent(s, D.ev_t[(size_t)(k - 1)], 0));   // column k+1 was last written by the bulk update of step k-1
    if (early) GPX_RET(gpx_dist_step_update_next_col(h, k));   // the owner's block (k+1,k+1) is already on its way through potrf_diag
    else GPX_RET(gpx_dist_step_update_cols(h, k, k + 1, 1, 0));
    GPX_HIP_CHECK(h, hipEventRecord(D.ev_s[(size_t)k], s));
    GPX_HIP_CHECK(h, hipStreamWaitEvent(t, D.ev_s[(size_t)k], 0));
    GPX_RET(gpx_dist_step_update_cols(h, k, k + 2, -1, 1));
    GPX_HIP_CHECK(h, hipEventRecord(D.ev_t[(size_t)k], t));
    last_t = k;
  }
  if (dist_la && last_t >= 0) GPX_HIP_CHECK(h, hipStreamWaitEvent(s, D.ev_t[(size_t)last_t], 0));
  GPX_RET(gpx_dist_extract_z(h));
  if (comm) GPX_NCCL_CHECK(h, ncclBroadcast(D.z, D.z, (size_t)D.npad, ncclDouble, D.r_owner, comm, s));
  const int64_t nt = (int64_t)D.test_rows.size();
  std::vector<int64_t> idx((size_t)std::max<int64_t>(nt, 1));
  std::vector<double> ml((size_t)std::max<int64_t>(nt, 1)), vl((size_t)std::max<int64_t>(nt, 1));
  double ld_part = 0.0, quad = 0.0; int info = 0;
  GPX_RET(gpx_dist_local_results(h, m_new, &ld_part, &quad, idx.data(), ml.data(), vl.data(), &info));
  // combine: logdet and info by all-reduce; mean/var scattered into full-length vectors and summed
  const int64_t m = D.m;
  std::vector<double> full((size_t)(2 * m + 2), 0.0);
  for (int64_t t = 0; t < nt; ++t) { full[(size_t)idx[(size_t)t]] = ml[(size_t)t]; full[(size_t)(m + idx[(size_t)t])] = vl[(size_t)t]; }
  full[(size_t)(2 * m)] = ld_part;
  // LAPACK-style info is the FIRST failing pivot over all ranks: min over the positive values, carried
  // as max of the negated value (0 = "no failure" maps to -inf)
  double neg_info = info > 0 ? -(double)info : -1e300;
  if (comm) {
    GPX_RET(dist_ensure(h, &D.red, &D.red_cap, 2 * m + 2));
    GPX_HIP_CHECK(h, hipMemcpyAsync(D.red, full.data(), (size_t)(2 * m + 1) * sizeof(double), hipMemcpyHostToDevice, s));
    GPX_HIP_CHECK(h, hipMemcpyAsync(D.red + 2 * m + 1, &neg_info, sizeof(double), hipMemcpyHostToDevice, s));
    GPX_NCCL_CHECK(h, ncclAllReduce(D.red, D.red, (size_t)(2 * m + 1), ncclDouble, ncclSum, comm, s));
    GPX_NCCL_CHECK(h, ncclAllReduce(D.red + 2 * m + 1, D.red + 2 * m + 1, 1, ncclDouble, ncclMax, comm, s));
    GPX_HIP_CHECK(h, hipMemcpyAsync(full.data(), D.red, (size_t)(2 * m + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
    GPX_HIP_CHECK(h, hipMemcpyAsync(&neg_info, D.red + 2 * m + 1, sizeof(double), hipMemcpyDeviceToHost, s));
    GPX_HIP_CHECK(h, hipStreamSynchronize(s));
  }
  full[(size_t)(2 * m + 1)] = neg_info > -1e299 ? -neg_info : 0.0;
  if (mean_out) memcpy(mean_out, full.data(), (size_t)m * sizeof(double));
  if (var_out) memcpy(var_out, full.data() + m, (size_t)m * sizeof(double));
  if (logp) *logp = -0.5 * quad - full[(size_t)(2 * m)] - 0.5 * (double)D.n * log(2.0 * M_PI);
  return (int)full[(size_t)(2 * m + 1)];
}

void gpx_dist_free(gpx_handle* h) {
  DistState& D = h->dist;
  for (auto e : D.ev_s) (void)hipEventDestroy(e);
  for (auto e : D.ev_t) (void)hipEventDestroy(e);
  for (auto e : D.ev_pan) (void)hipEventDestroy(e);
  for (auto e : D.ev_diag) (void)hipEventDestroy(e);
  D.ev_s.clear(); D.ev_t.clear(); D.ev_pan.clear(); D.ev_diag.clear();
  void* bufs[] = {D.A, D.L, D.W, D.send, D.recv, D.panel, D.panel2, D.xloc, D.gid, D.type, D.r, D.z, D.small, D.red, D.group[0], D.group[1]};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (D.comm) (void)ncclCommDestroy((ncclComm_t)D.comm);
  if (D.s_bulk) (void)hipStreamDestroy(D.s_bulk);
  D = DistState();
}
