// gpx_internal.h -- shared declarations of the HIP implementation (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gpx.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ exp for the gram epilogues
// exp(x) for x <= 0 with the constants in SGPRs.  The arithmetic is libdevice's (__ocml_exp_f64: Cody-Waite reduction by
// ln 2, degree-11 polynomial, v_ldexp_f64) operation for operation -- the results carry the same bits -- but hipcc
// materialises every 64-bit constant of that routine with two v_mov_b32 in front of a v_fmac_f64 (22 of the routine's ~45
// VALU instructions; a v_mov_b32 costs a wave the same four issue cycles as a v_fma_f64), and the gram build is bound by
// exactly that: 8.4 M entries x ~75 VALU instructions at N = 4096.  v_fma_f64 takes the addend from an SGPR pair instead.
#ifdef __HIPCC__
__device__ __forceinline__ double gpx_fma_sc(double a, double b, double c_uniform) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_uniform));
  return d;
}
// c_uniform * b + addend (the constant is a multiplicand)
__device__ __forceinline__ double gpx_fma_cs(double c_uniform, double b, double addend) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "s"(c_uniform), "v"(b), "v"(addend));
  return d;
}
__device__ __forceinline__ double gpx_exp_neg(double x) {
  double t;
  asm("v_mul_f64 %0, %1, %2" : "=v"(t) : "s"(0x1.71547652b82fep+0), "v"(x));
  t = __builtin_rint(t);                                              // n = rint(x log2 e)
  double r = gpx_fma_cs(-0x1.62e42fefa39efp-1, t, x);                 // r = x - n ln2_hi - n ln2_lo
  r = gpx_fma_cs(-0x1.abc9e3b39803fp-56, t, r);
  double p = gpx_fma_sc(r, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
  p = gpx_fma_sc(r, p, 0x1.71dee623fde64p-19);
  p = gpx_fma_sc(r, p, 0x1.a01997c89e6b0p-16);
  p = gpx_fma_sc(r, p, 0x1.a01a014761f6ep-13);
  p = gpx_fma_sc(r, p, 0x1.6c16c1852b7b0p-10);
  p = gpx_fma_sc(r, p, 0x1.1111111122322p-7);
  p = gpx_fma_sc(r, p, 0x1.55555555502a1p-5);
  p = gpx_fma_sc(r, p, 0x1.5555555555511p-3);
  p = gpx_fma_sc(r, p, 0x1.000000000000bp-1);
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  const double v = __builtin_amdgcn_ldexp(p, (int)t);
  return x < -1075.0 ? 0.0 : v;
}
#endif

#if defined(__HIPCC__)
// The device pointer of a hyper-parameter array that travels in the kernel arguments (field at byte offset `off` of the
// kernel's single by-value argument): a global-address-space pointer into the kernarg segment, so that indexed reads stay
// plain global loads (taking the address of the by-value copy would put it in scratch).
typedef const __attribute__((address_space(1))) double* gpx_gptr;
__device__ __forceinline__ gpx_gptr gpx_kernarg_doubles(size_t off) {
  const __attribute__((address_space(4))) char* ka = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  return (gpx_gptr)(uintptr_t)(ka + off);
}
#endif

// ------------------------------------------------------------------ error plumbing
#define GPX_HIP_CHECK(h, expr)                                                         \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess) {                                                           \
      gpx_set_error(h, std::string(#expr) + ": " + hipGetErrorString(e__) + " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
      return GPX_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

#define GPX_RET(expr)                \
  do {                               \
    int r__ = (expr);                \
    if (r__ != GPX_OK) return r__;   \
  } while (0)

void gpx_set_error(gpx_handle* h, const std::string& msg);

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// --------------------------------------------------------------------- device hyper
// Hyper-parameters as the gram kernel consumes them: per-dimension length-scale and
// period expanded to d entries (device buffer, 2*d doubles: length[d], period[d]).
// Hyper-parameters of a single draw can ride in the kernel arguments of the step's FIRST launch (which also writes them to
// d_hyp for the kernels behind it) instead of a host-to-device copy in front of it: at most this many doubles (2 d + 4)
#define GPX_HYP_INLINE 40
struct GramLaunch {
  const double* hyp_inline = nullptr; int hyp_inline_n = 0;   // host values for the kernel arguments (see GPX_HYP_INLINE); `hyp` is where they are written
  int kind;
  int d;
  double k_scale;
  const double* hyp;    // device: length[d] then period[d]
  const double* dev_scalars = nullptr;  // device {k_scale, noise_scalar, jitter}: overrides the by-value fields
  const double* x1;     // device (n1,d)
  const double* x2;     // device (n2,d)
  int64_t n1, n2;
  double* out;          // device, leading dimension ld
  int64_t ld;
  int lower_only;       // skip tiles strictly above the diagonal (x1 == x2 layouts)
  int diag_rule;        // add noise+jitter on the diagonal
  const double* noise;  // device vector (n1) or NULL
  const int64_t* row_gid = nullptr;  // device: global row id per x1 row (diagonal rule / noise index); NULL: row itself
  double noise_scalar;  // used when noise == NULL
  double jitter;
  int64_t col0 = 0;     // x2 / out are the column slice starting at this global column (lower test, diagonal rule)
  int* zero_info = nullptr;  // device int cleared by the launch
  unsigned* zero_flags = nullptr; int n_zero_flags = 0;   // device flag words cleared by the launch
  hipEvent_t start_event = nullptr, stop_event = nullptr;   // the kernel's own dispatch timestamps (profiling)
  // batched hyper-parameter draws (blockIdx.z = draw): per-draw strides, in elements of the pointer they apply to.
  // x1 / x2 are shared by the draws; hyp and dev_scalars move together (dev_scalars = hyp + 2 d).
  int batch = 1;
  int64_t s_out = 0, s_hyp = 0, s_noise = 0, s_info = 0, s_flags = 0;
};
int launch_gram(gpx_handle* h, hipStream_t s, const GramLaunch& g);
// one-block-column problems: [K ; identity padding ; k_cross ; r ; zero rows] in ONE launch (same bits as the separate launches)
int launch_build_small(gpx_handle* h, hipStream_t s, int kind, const double* x, const double* xnew, const double* hyp,
                       const double* dev_scalars, const double* noise, const double* r, double* out, int64_t ld, int64_t n,
                       int64_t npad, int64_t m_rows, int64_t r_row, int64_t rows_total, int d, int diag_rule, int* zero_info,
                       unsigned* zero_flags, int n_zero_flags, int64_t ident_row0 = -1, double* zero2 = nullptr, double* r_copy = nullptr,
                       int batch = 1, int64_t s_out = 0, int64_t s_hyp = 0, int64_t s_noise = 0, int64_t s_r = 0, int64_t s_info = 0,
                       const double* hyp_inline = nullptr, int hyp_inline_n = 0);

// C[M,N] = beta*C + alpha * A[M,K] * B[N,K]^T   (all row-major; K contiguous in A and B)
struct GemmLaunch {
  const double* A; int64_t lda;
  const double* B; int64_t ldb;
  double* C; int64_t ldc;
  int64_t M, N, K;
  double alpha, beta;
  int lower;            // only tiles that intersect {row_off + i >= col_off + j}
  int64_t row_off, col_off;
  int b_lower_tri;      // B is lower triangular (B[n][k] = 0 for k > n): skip dead K
  int small_tile;       // 1: 32x32 workgroup tiles (latency path), 0: 64x64 (throughput path)
  int role = 0;         // 64x64 path: 1 = look-ahead / near-block launch (a separately named instantiation of the same kernel)
  int cyc_P = 1, cyc_r = 0, cyc_nb = 128;   // block-cyclic local->global row map for the `lower` test (distributed path)
  int a_upper_tri = 0;  // A[m][k] = 0 for k < m: skip the dead K range below each tile's first row (64x64 path)
  // completion event attached to the kernel's own dispatch (hipExtLaunchKernelGGL stopEvent): no separate
  // barrier packet behind the kernel (tools/sync_cost: 3.7 -> 2.3 us to the next kernel of the stream)
  hipEvent_t stop_event = nullptr;
  hipEvent_t start_event = nullptr;   // dispatch start time (profiling: the kernel's own timestamps, not record packets around it)
  // one-pass (32x32-tile) path: every workgroup that stores a tile publishes it (write-through stores, then one
  // agent-scope add to this counter) for a consumer that is already resident (potrf_diag's resident start)
  unsigned* signal = nullptr;
  int64_t signal_m = 0;   // > 0: only the tiles whose first row is below signal_m publish and count (the rest store plainly)
  int first_col = 0;      // bulk launch (role 0, 1-D tile map): enumerate the first `first_col` tile columns first; they publish and count into `signal`
  // one-pass path: the kernel is launched early and sleeps until *wait_flag >= wait_count (a producer that is
  // still running publishes its output with write-through stores and then raises the flag)
  const unsigned* wait_flag = nullptr; unsigned wait_count = 0;
  const unsigned* wait_flag2 = nullptr; unsigned wait_count2 = 0;   // a second producer
  int64_t wait_m = 0;     // > 0: only the tiles whose first row is below wait_m wait (the rest start at once)
  int* wait_info = nullptr;   // device int set to -7 if the bounded wait gives up
  // batched hyper-parameter draws (blockIdx.z = draw): per-draw strides in elements (doubles for A/B/C, unsigned for the
  // flags, int for wait_info); 0 = shared by the draws
  int batch = 1;
  int64_t sA = 0, sB = 0, sC = 0, s_flag = 0, s_info = 0;
};
int launch_gemm_nt(gpx_handle* h, hipStream_t s, const GemmLaunch& g);
// the chain's band kernel (gemm.hip, chain_band_kernel): the diagonal band between potrf_diag(k) and potrf_diag(k+1) in one
// launch -- phase 0 (optional), S1 = A_{k+1,k} W_k^T, A_{k+1,k+1} -= S1 S1^T, and (row2) S2 = A_{k+2,k} W_k^T,
// A_{k+2,k+1} -= S2 S1^T
struct BandLaunch {
  int nb = 128;
  const double* W = nullptr;
  const double* A1 = nullptr; const double* A2 = nullptr;
  double* D11 = nullptr; double* D21 = nullptr;
  const double* P0 = nullptr; int64_t ldp = 0, sP0 = 0;   // phase-0 operand L_{k+1,k-1}
  double* S1 = nullptr; double* S2 = nullptr;             // scratch nb x nb each
  int64_t ld = 0;
  const unsigned* flagW = nullptr;                          // raised by potrf_diag(k) behind W_k
  const unsigned* flagF = nullptr; unsigned countF = 0;     // row 2: tiles published by the panel stream's column updates of step k-1
  unsigned* flagS = nullptr; unsigned* flagS2 = nullptr;    // tiles of S1 / S2 published
  unsigned* flagU = nullptr;                                // tiles of A_{k+1,k+1} published (potrf_diag(k+1) polls)
  unsigned* flagD = nullptr;                                // tiles of A_{k+2,k+1} published (the panel stream's panel of step k+1 polls)
  const unsigned* flagB = nullptr; unsigned countB = 0;     // phase 0: first block column of the bulk launch that last wrote A_{k+1,k+1}
  int* info = nullptr;
  int row2 = 0, big = 0;   // big: row 2's update in the summation order of the 64x64 look-ahead kernel
  hipEvent_t stop_event = nullptr;
  int batch = 1;
  int64_t sA = 0, sW = 0, s_mini = 0, s_flag = 0, s_info = 0;
};
int launch_chain_band(gpx_handle* h, hipStream_t s, const BandLaunch& L);
#define GPX_N_FLAG_ARRAYS 7   // flagsU, flagsW, flagsS, flagsF, flagsS2, flagsD, flagsB: nblk words each, per draw

// Diagonal block: A (nb x nb at lda) -> L (lower, written at ldl) and W = inv(L) (dense nb x nb)
struct PotrfLaunch {
  int nb;
  const double* A; int64_t lda;
  double* L; int64_t ldl;
  double* W;
  int* info; int col0;
  hipEvent_t stop_event = nullptr;
  const unsigned* wait_flag = nullptr; unsigned wait_count = 0;   // resident start: sleep until *wait_flag >= wait_count
  unsigned* done_flag = nullptr;                                   // raised behind the last store of W
  unsigned long long wait_ticks = 25000000ull;                     // bound of the resident wait (100 MHz ticks: 0.25 s)
  int batch = 1;                                                   // blockIdx.x = draw
  int64_t sA = 0, sL = 0, sW = 0, s_info = 0, s_flag = 0;
};
int launch_potrf_diag(gpx_handle* h, hipStream_t s, const PotrfLaunch& p);
// every diagonal block [k0, k1) of the factorisation in one launch: one resident workgroup per draw walks down the diagonal
// (block k at A + k nb (lda + 1), W_k at W + k nb^2), handing over through flagsU / flagsW (potrf.hip: potrf_chain_kernel)
struct PotrfChainLaunch {
  int nb = 128;
  const double* A = nullptr; int64_t lda = 0;   // block 0
  double* L = nullptr; int64_t ldl = 0;
  double* W = nullptr;
  int* info = nullptr;
  const unsigned* flagsU = nullptr; unsigned* flagsW = nullptr;
  int k0 = 0, k1 = 0, raise_until = 0, force_k = -1;
  unsigned wait_count = 0;
  unsigned long long wait_ticks = 25000000ull;
  int batch = 1;
  int64_t sA = 0, sL = 0, sW = 0, s_info = 0, s_flag = 0;
};
int launch_potrf_chain(gpx_handle* h, hipStream_t s, const PotrfChainLaunch& p);
// one wave per draw that sleeps until *flag >= count: the kernel boundary behind it releases the next kernel of the stream
int launch_chain_relay(gpx_handle* h, hipStream_t s, const unsigned* flag, unsigned count, int* info, int batch, int64_t s_flag,
                       int64_t s_info);

// y[0:ncols) = beta*y + alpha * x[0:nrows)^T A[0:nrows, 0:ncols)
int launch_gemv_t(gpx_handle* h, hipStream_t s, const double* A, int64_t lda, int nrows, int64_t ncols,
                  const double* x, double* y, double alpha, double beta);

// fill helpers
int launch_fill_rect(gpx_handle* h, hipStream_t s, double* A, int64_t ld, int64_t r0, int64_t r1,
                     int64_t c0, int64_t c1, double diag, int batch = 1, int64_t sA = 0);
int launch_fill_identity_pad(gpx_handle* h, hipStream_t s, double* A, int64_t ld, int64_t n,
                             int64_t npad, int64_t rows_total, int batch = 1, int64_t sA = 0);
int launch_set_row(gpx_handle* h, hipStream_t s, double* dst, const double* src, int64_t n,
                   int64_t npad);
// one launch: row `r_row` of A <- r (n entries, zero padded to npad; skipped when r_row < 0), rows [z0, z1) <- 0, *info <- 0
int launch_border_rows(gpx_handle* h, hipStream_t s, double* A, int64_t ld, int64_t npad, int64_t n, int64_t r_row,
                       const double* r, int64_t z0, int64_t z1, int* info, int batch = 1, int64_t sA = 0, int64_t s_r = 0);

// epilogue reductions
#define GPX_OUT_HEAD 10   // result block: doubles in front of mean -- scal[8], then 4 ints of info
struct EpilogueLaunch {
  const double* L; int64_t ld;   // factor (diagonal is read for the log-determinant)
  int64_t n;                     // valid order
  const double* z;               // device (n): z = L^-1 (y - m), or NULL (then only logdet)
  const double* vt; int64_t vt_ld;  // V^T = k_cross L^-T, m rows
  int64_t m;                     // test rows
  const double* m_new;           // device (m) or NULL
  double prior_diag;             // k(x*,x*) incl. noise+jitter of the diagonal rule
  const double* dev_prior_diag = nullptr;  // device copy (graph replays); overrides prior_diag
  double* scal;                  // device: [0]=logdet [1]=quad
  double* mean; double* var;     // device (m)
  double* z_copy = nullptr; int64_t npad = 0;   // also copy z[0:npad) here (the handle's z for later posterior calls)
  // gradient path: alpha = U z (U = L^-T, ld as L, rows n, columns gemv_ncols) computed by extra blocks of the same launch
  const double* gemv_U = nullptr; double* gemv_alpha = nullptr; int64_t gemv_ncols = 0;
  // batched hyper-parameter draws (blockIdx.y = draw): per-draw strides in doubles.  L, z, vt and gemv_U live in the factor
  // matrix (s_mat); scal, mean, var and gemv_alpha in the result block (s_out); z_copy, m_new, dev_prior_diag, part on their own
  int batch = 1;
  int64_t s_mat = 0, s_out = 0, s_zcopy = 0, s_mnew = 0, s_hyp = 0, s_part = 0;
  // single draw: the final launch also writes the result block [scal | info | mean | var] to this pinned host mirror (same
  // layout as the device block, var at GPX_OUT_HEAD + mirror_var_off; info copied from info_src), so that fetching the
  // results is a stream synchronisation and no device-to-host copy behind the last kernel
  double* mirror = nullptr; const int* info_src = nullptr; int64_t mirror_var_off = 0;
};
int launch_epilogue(gpx_handle* h, hipStream_t s, const EpilogueLaunch& e);
// split epilogue: columns [0, c1) early as partial sums (part: 2 + 2 m doubles), then [c1, n) + part, finalised
int launch_epilogue_partial(gpx_handle* h, hipStream_t s, const EpilogueLaunch& e, int64_t c1, double* part);
int launch_epilogue_final(gpx_handle* h, hipStream_t s, const EpilogueLaunch& e, int64_t c0, double* part);

// acquisition epilogue over device-resident (mean, var): EI / UCB values and their arg-max (reduce.hip)
int launch_acquisition(gpx_handle* h, hipStream_t s, const double* mean, const double* var, int64_t m, int kind, double param,
                       double mu_scale, double mu_shift, double sd_scale, double* acq, double* best);

// post_reduce only (mean / variance of m rows of V^T against z)
int launch_post_reduce(gpx_handle* h, hipStream_t s, const EpilogueLaunch& e);

// gradient of the log-marginal likelihood (grad.hip)
int grad_partials_per_chunk(void);
int grad_chunk_dims(void);
int launch_grad_reduce(gpx_handle* h, hipStream_t s, int kind, const double* x, const double* hyp, const double* kinv,
                       int64_t ldk, const double* alpha, int64_t n, int d, int per_dim, int nwg, double* part,
                       double* sums, int batch = 1, int64_t s_hyp = 0, int64_t s_kinv = 0, int64_t s_out = 0,
                       double* mirror_sums = nullptr);   // single draw: the sums also go to this pinned host address (EpilogueLaunch::mirror)
int launch_utri_gemv(gpx_handle* h, hipStream_t s, const double* U, int64_t ld, const double* z, double* alpha,
                     int64_t n, int64_t ncols);

// ---------------------------------------------------------------- distributed state (dist.hip)
#define GPX_DIST_GMAX 4   // panels per grouped bulk pass, at most
struct DistState {
  bool configured = false, have_r = false;
  int rank = 0, P = 1, d = 0;
  int64_t n = 0, m = 0, npad = 0, nblk = 0, eblk = 0, RB = 0, nloc = 0, ld = 0, chunk_rows = 0;
  int kind = 0; double k_scale = 0, k_jitter = 0;
  double *A = nullptr, *L = nullptr, *W = nullptr, *send = nullptr, *recv = nullptr, *panel = nullptr;
  double* group[2] = {nullptr, nullptr}; int64_t group_cap[2] = {0, 0};   // K-contiguous group buffers of the grouped bulk updates
  double* panel2 = nullptr; int64_t panel2_cap = 0;   // panels alternate between two buffers: the bulk update of step k still
                                                        // reads panel k while step k+1 gathers the next one (look-ahead)
  std::vector<hipEvent_t> ev_s, ev_t;                   // look-ahead hand-overs of the RCCL driver
  std::vector<hipEvent_t> ev_pan, ev_diag;              // early-diagonal schedule: panel rows solved (main -> second chain stream), potrf_diag done (back)
  double *xloc = nullptr, *r = nullptr, *z = nullptr, *small = nullptr, *red = nullptr;
  int64_t* gid = nullptr; int* type = nullptr;
  int64_t a_cap = 0, l_cap = 0, w_cap = 0, send_cap = 0, recv_cap = 0, panel_cap = 0, xloc_cap = 0, r_cap = 0, z_cap = 0,
          small_cap = 0, red_cap = 0, gid_cap = 0, type_cap = 0;
  int r_owner = 0; int64_t r_local_row = -1;
  std::vector<std::pair<int64_t, int64_t>> test_rows;   // (local row, test index) of the k_cross rows held here
  void* comm = nullptr;                                  // ncclComm_t
  hipStream_t s_bulk = nullptr;                          // bulk stream on a CU mask that leaves a few CUs to the chain (dist.hip)
  int64_t gather_stride_rows = 0;                        // rows per rank in `recv` for the step being gathered (0: chunk_rows)
};
void gpx_dist_free(gpx_handle* h);

// ------------------------------------------------------------------------ the handle
struct StageProf {
  double ms = 0; int64_t launches = 0; double flops = 0, bytes = 0;
};

// One batch of bordered matrices being factorised (schedule.hip): `batch` independent hyper-parameter draws of the same
// shape, draw b at base + b * stride.  The GP factor (d_A / d_L / d_W) and the M x M covariance of the sampling path
// (d_A2 / d_L2 / d_W2) are two such sets.
struct Mats {
  double* A = nullptr; double* L = nullptr; double* W = nullptr;
  int64_t npad = 0, ld = 0, rows_total = 0;
  int nb = 128, batch = 1;
  int64_t sA = 0, sW = 0;          // per-draw strides (doubles) of A / L and of W
};

struct gpx_handle {
  int device = 0;
  // P (s_main): the chain -- diagonal blocks; P2 (s_main2): the chain's band kernels; Q (s_far): panels and look-ahead
  // columns; T (s_trail): bulk updates
  hipStream_t s_main = nullptr, s_trail = nullptr, s_far = nullptr, s_main2 = nullptr;
  unsigned* d_flags = nullptr; int64_t flags_cap = 0;   // per draw: flagsU, flagsW, flagsS, flagsF, flagsS2, flagsD (nblk each)
  bool flags_clean = false;        // the flags were cleared by the gram launch of this call (no memset needed)
  hipEvent_t ev_chain2 = nullptr;
  const EpilogueLaunch* epi = nullptr;   // set by fit_predict around the factorisation: the split-panel loop may launch the early half
  int64_t epi_split_col = 0;             // columns [0, epi_split_col) were reduced early into d_epart (0: not split)
  double* d_epart = nullptr; int64_t epart_cap = 0;
  hipEvent_t ev_build = nullptr;   // K built on the chain stream -> the border rows are built on the panel stream
  hipEvent_t ev_near = nullptr;    // the panel stream's most recent near-block launch (split-panel schedule, grouped phases)
  std::string err;
  hipDeviceProp_t prop;

  // options (include/gpx.h)
  int nb = 128;
  bool nb_auto = true;   // no explicit "nb" option: 64 for n <= 64 (potrf_diag<64> runs 12 us, <128> 29 us), else 128
  int lookahead = 1;     // 0: everything on one stream; 1: the split-panel schedule from split_min_blocks block columns on
  int resident = 2;      // split-panel schedule: 0 = every hand-over of the chain is a kernel boundary; 1 = potrf_diag starts resident
                         // (sleeps on the tile counter of A -= S S^T); 2 = also S = A W_k^T starts resident (sleeps on potrf_diag's flag);
                         // 3 = also ONE potrf_diag workgroup per draw for the whole factorisation (it keeps its CU; measured, not the
                         // default: the relay kernels it needs on the panel stream cost what the CU wait did)
  int pair_min = 28;     // pair the bulk updates (K = 2 nb) while more than this many block columns remain; 0 = never
  int quad_min = 40;     // ... and four at a time (K = 4 nb) while more than this many remain; 0 = never
  int split_min_blocks = 3;   // the split-panel (multi-stream) schedule runs from this many block columns; below, one stream
  int la_rows = 3072;    // look-ahead column updates use 64x64 tiles while at least this many rows remain below; 0 = never
  int batch_group = 4;   // batches of >= 4 draws: bulk updates grouped g panels at a time throughout (0: the single-draw thresholds)
  double* h_mirror = nullptr; int64_t mirror_cap = 0; bool mirror_valid = false;   // pinned mirror of the result block (EpilogueLaunch::mirror)
  double hyp_host[GPX_HYP_INLINE]; int hyp_host_n = 0;   // the draw's hyper-parameters, not uploaded yet: the next first launch carries them
  bool hyp_defer = false;     // set by the entry points whose first launch can carry them (upload_params_batch then skips the copy)
  int bulk_first_col = 1;  // ungrouped bulk launches enumerate and publish their first block column first (schedule.hip)
  int debug_force_timeout = 0;   // test hook: the next factorisation's first resident wait gives up at once (info = -7)

  // resident inputs (x and x_new are shared by the draws of a batch; r, m_new, hyp, noise are per draw)
  double* d_x = nullptr; int64_t n = 0; int d = 0; int64_t x_cap = 0;
  double* d_xnew = nullptr; int64_t m = 0; int64_t xnew_cap = 0;
  double* d_r = nullptr; int64_t r_cap = 0; bool have_r = false;
  double* d_mnew = nullptr; int64_t mnew_cap = 0; bool have_mnew = false;
  double* d_hyp = nullptr; int64_t hyp_cap = 0;
  double* d_noise = nullptr; int64_t noise_cap = 0;

  // factor workspace: A (working copy, lower) and L (result) share a layout:
  // rows [0,npad) the matrix, rows [npad, npad+epad) the border rows (k_cross, then r)
  double* d_A = nullptr; double* d_L = nullptr; int64_t a_cap = 0, l_cap = 0;
  double* d_W = nullptr; int64_t w_cap = 0;     // (npad/nb) blocks of nb*nb: inv(L_kk)
  int64_t npad = 0, epad = 0, ld = 0;
  int64_t m_in_factor = 0; bool r_in_factor = false;
  int64_t m_resident = 0;          // the result block of draw 0 holds the posterior mean / variance of this many test points
  bool have_factor = false;
  // batched draws: number of draws of the current factorisation and their strides (doubles; flags: unsigned)
  int batch = 1;
  int64_t s_mat = 0, s_w = 0, s_z = 0, s_r = 0, s_mnew = 0, s_hyp = 0, s_noise = 0, s_out = 0, s_flags = 0, s_mini = 0, s_epart = 0;

  // cached hyper-parameters of the factor (for later posterior calls)
  int f_kind = 0; double f_scale = 0, f_jitter = 0;
  std::vector<double> f_length, f_period;

  // solve workspace for posterior-with-cached-factor
  double* d_S = nullptr; int64_t s_cap = 0;      // rows x npad
  double* d_cov = nullptr; int64_t cov_cap = 0;
  double* d_z = nullptr; int64_t z_cap = 0; bool have_z = false;  // z = L^-1 r (npad)

  // scratch of the stateless gpx_gram and of the vector back-substitution
  double* d_aug[4] = {nullptr, nullptr, nullptr, nullptr}; int64_t aug_cap[4] = {0, 0, 0, 0};   // gram.hip: augmented input rows of the MFMA gram kernel, one array per stream (chain, panel, bulk, any other)
  double* d_g1 = nullptr; int64_t g1_cap = 0;
  double* d_g2 = nullptr; int64_t g2_cap = 0;
  double* d_gout = nullptr; int64_t gout_cap = 0;
  double* d_vec = nullptr; int64_t vec_cap = 0;
  // gradient workspace: K^-1 (npad x npad, lower tiles), per-workgroup partial sums, final sums
  double* d_kinv = nullptr; int64_t kinv_cap = 0;
  // set by gpx_lml_grad around its factorisation: accumulate K^-1 += U_{:,k} U_{:,k}^T on the bulk stream as the
  // panels complete (U = the identity border rows of d_L) instead of one U U^T product after the last panel
  const double* kinv_U = nullptr; int64_t kinv_rows = 0;
  int64_t s_kinv = 0;   // per-draw stride (doubles) of d_kinv in a batched gradient call
  double* d_gpart = nullptr; int64_t gpart_cap = 0;
  // secondary factor buffers (Cholesky of the posterior covariance for MVN sampling)
  double* d_A2 = nullptr; int64_t a2_cap = 0;
  double* d_L2 = nullptr; int64_t l2_cap = 0;
  double* d_W2 = nullptr; int64_t w2_cap = 0;
  double* d_eps = nullptr; int64_t eps_cap = 0;   // standard-normal draws / samples of the sampling path
  double* d_smp = nullptr; int64_t smp_cap = 0;

  // small outputs
  double* d_scal = nullptr;    // [0]=logdet [1]=quad
  int* d_info = nullptr;
  double* d_mean = nullptr; double* d_var = nullptr; int64_t out_cap = 0;  // d_var aliases d_mean + m
  double* h_pinned = nullptr; int64_t pinned_cap = 0;  // pinned staging for small results
  // All small results live in ONE device block per draw so that one copy brings them to the host:
  //   d_out = [ scal[8] | info (4 ints) | mean[m] | var[m] ] x batch (stride s_out);  d_scal / d_info / d_mean / d_var
  //   point into draw 0's block.
  double* d_out = nullptr;
  // pinned ring for hyper-parameter uploads (a pageable hipMemcpyAsync costs ~28 us of host time on this stack)
  double* h_up = nullptr; int64_t up_slot = 0; int up_next = 0; hipEvent_t up_ev[8] = {}; bool up_used[8] = {};

  // hand-over events of the split-panel schedule
  std::vector<hipEvent_t> ev_panel, ev_trail;
  hipEvent_t ev_join = nullptr, ev_join2 = nullptr;
  std::vector<hipEvent_t> ev_far, ev_w;
  double* d_mini = nullptr; int64_t mini_cap = 0;   // per draw: 4 x (nb x nb) scratch of the split-panel schedule (S1 = L_{k+1,k} and S2 = L_{k+2,k} of the chain, alternating)

  DistState dist;

  // profiling
  int prof_mode = 0; bool prof_open = false;
  StageProf prof[GPX_N_STAGES];
  struct PendingEv { int stage; hipEvent_t a, b; };
  std::vector<PendingEv> pending;
  std::vector<hipEvent_t> ev_pool;
};

// the handle's primary factor buffers as a matrix set, with the current batch and strides
static inline Mats primary_mats(const gpx_handle* h, int64_t rows_total) {
  Mats M;
  M.A = h->d_A; M.L = h->d_L; M.W = h->d_W; M.npad = h->npad; M.ld = h->ld; M.rows_total = rows_total;
  M.nb = h->nb; M.batch = h->batch; M.sA = h->s_mat; M.sW = h->s_w;
  return M;
}

// schedule.hip: right-looking blocked Cholesky of every matrix of the set (rows below npad ride along as border rows);
// tri_base >= 0: the border rows from tri_base on are an identity that is activated block by block (gpx_lml_grad)
int factor_loop(gpx_handle* h, const Mats& M, int64_t tri_base = -1);
int ensure_events(gpx_handle* h, size_t nblk, int nb, int batch);
template <typename T>
int ensure(gpx_handle* h, T** p, int64_t* cap, int64_t count) {
  if (count <= *cap && *p) return GPX_OK;
  if (*p) { (void)hipFree(*p); *p = nullptr; *cap = 0; }
  if (count <= 0) count = 1;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), (size_t)count * sizeof(T));
  if (e != hipSuccess) {
    gpx_set_error(h, std::string("hipMalloc failed: ") + hipGetErrorString(e));
    *p = nullptr;
    return GPX_ERR_NOMEM;
  }
  *cap = count;
  return GPX_OK;
}

// profiling helpers (api.hip)
void prof_begin(gpx_handle* h, hipStream_t s, int stage, double flops, double bytes);
void prof_end(gpx_handle* h, hipStream_t s);
// bracket by the kernel's OWN dispatch timestamps (hipExtLaunchKernelGGL start/stop events) instead of record packets
// around it: returns false (and null events) when the stage is not being profiled
bool prof_pair(gpx_handle* h, int stage, double flops, double bytes, hipEvent_t* a, hipEvent_t* b);
