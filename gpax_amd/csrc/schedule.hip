// schedule.hip -- host-side drivers of the bordered right-looking blocked Cholesky (gfx950).
//
// What `jnp.linalg.cholesky` inside numpyro's MultivariateNormal (gpax/gp.py:147-151) and
// `cho_factor` / `cho_solve` (gpax/gp.py:296-300) are to the reference: here one factorisation of the
// bordered matrix [K ; k_cross ; r] (the border rows ride along as "rows below the panel", so the
// triangular solves are absorbed in the panel GEMMs).  Two drivers:
//   * factor_loop_single: everything on one stream (one or two block columns, or option "lookahead" 0);
//   * SplitSchedule: the split-panel schedule on four streams with in-kernel hand-overs on the chain.
// Every launch carries `batch` independent hyper-parameter draws (gpax's vmap over hp samples,
// gpax/gp.py:354-364,384-390): draw b works on its own matrices / flags at base + b * stride, in
// blockIdx.z (GEMMs, gram) or blockIdx.x (potrf_diag).  The schedule -- which kernel updates which tile
// at which step -- does not depend on the batch size, so a draw's bits are those of a single-draw call.

#include <stdlib.h>

#include <algorithm>

#include "gpx_internal.h"

// Events that only order work between the handle's own streams on one device: a device-scope
// release is all they need; the default system-scope fence of an event record costs the NEXT kernel of the
// recording stream ~8 us (measured: 2.24 -> 2.12 ms per N=4096 step with hipEventDisableSystemFence).  Results
// reach the host through stream-ordered copies, which carry their own fences.
static unsigned sync_event_flags() { return hipEventDisableTiming | hipEventDisableSystemFence; }

int ensure_events(gpx_handle* h, size_t nblk, int nb, int batch) {
  const unsigned fl = sync_event_flags();
  if (!h->ev_join) GPX_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_join, fl));
  if (!h->ev_join2) GPX_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_join2, fl));
  if (!h->ev_chain2) GPX_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_chain2, fl));
  if (!h->ev_build) GPX_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_build, fl));
  if (!h->ev_near) GPX_HIP_CHECK(h, hipEventCreateWithFlags(&h->ev_near, fl));
  h->s_flags = GPX_N_FLAG_ARRAYS * (int64_t)std::max<size_t>(nblk, 64);     // flagsU, flagsW, flagsS, flagsF, flagsS2, flagsD
  h->s_mini = 4 * (int64_t)nb * nb;                                         // S1 (two, alternating), S2 (two, alternating)
  GPX_RET(ensure(h, &h->d_flags, &h->flags_cap, h->s_flags * batch));
  GPX_RET(ensure(h, &h->d_mini, &h->mini_cap, h->s_mini * batch));
  while (h->ev_far.size() < nblk) {
    hipEvent_t a, b;
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&a, fl));
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&b, fl));
    h->ev_far.push_back(a);
    h->ev_w.push_back(b);
  }
  while (h->ev_panel.size() < nblk) {
    hipEvent_t a, b;
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&a, fl));
    GPX_HIP_CHECK(h, hipEventCreateWithFlags(&b, fl));
    h->ev_panel.push_back(a);
    h->ev_trail.push_back(b);
  }
  return GPX_OK;
}

// ---- launch builders shared by the two drivers: one place that knows the batch strides
namespace {

struct Ctx {
  gpx_handle* h;
  Mats M;
  int64_t rows_total, tri_base;
  int64_t nb, npad, ld, nblk;
  int64_t s_info;   // ints between the info words of consecutive draws (the info word lives in the result block)
  // tri_base >= 0: the border rows from `tri_base` on are an identity (gpx_lml_grad: they become U = L^-T); border
  // block j is still all-zero in the panels k < j, so step k only carries rows below tri_base + (k+1)*nb -- the
  // triangular inverse costs N^3/3 flops instead of N^3.
  int64_t rows_at(int64_t k) const {
    return tri_base >= 0 ? std::min<int64_t>(rows_total, tri_base + (k + 1) * nb) : rows_total;
  }
  double* A(int64_t r, int64_t c) const { return M.A + r * ld + c; }
  double* L(int64_t r, int64_t c) const { return M.L + r * ld + c; }
  double* W(int64_t k) const { return M.W + k * nb * nb; }
  void batch(GemmLaunch& g, bool a_mat, bool b_mat, bool c_mat) const {   // operands inside the factor matrices
    g.batch = M.batch;
    g.sA = a_mat ? M.sA : 0; g.sB = b_mat ? M.sA : 0; g.sC = c_mat ? M.sA : 0;
    g.s_flag = h->s_flags; g.s_info = s_info;
  }
  // panel solve  C = A_{rows, k} W_k^T  (B = W_k, lower triangular)
  GemmLaunch panel(int64_t k, int64_t r0, int64_t mrows, double* C, int64_t ldc, int64_t sC) const {
    GemmLaunch g;
    g.A = A(r0, k * nb); g.lda = ld;
    g.B = W(k); g.ldb = nb;
    g.C = C; g.ldc = ldc;
    g.M = mrows; g.N = nb; g.K = nb; g.alpha = 1.0; g.beta = 0.0;
    g.lower = 0; g.row_off = 0; g.col_off = 0; g.b_lower_tri = 1; g.small_tile = 1;
    g.batch = M.batch; g.sA = M.sA; g.sB = M.sW; g.sC = sC; g.s_flag = h->s_flags; g.s_info = s_info;
    return g;
  }
  // trailing update  A[rr.., cfirst..] -= L[rr.., kfirst..] L[cfirst.., kfirst..]^T over npan panels
  GemmLaunch update(int64_t rr, int64_t mr, int64_t cfirst, int64_t ncolblk, int64_t kfirst, int64_t npan, bool small) const {
    GemmLaunch t;
    t.A = L(rr, kfirst * nb); t.lda = ld;
    t.B = L(cfirst * nb, kfirst * nb); t.ldb = ld;
    t.C = A(rr, cfirst * nb); t.ldc = ld;
    t.M = mr; t.N = ncolblk * nb; t.K = npan * nb; t.alpha = -1.0; t.beta = 1.0;
    t.lower = 1; t.row_off = rr; t.col_off = cfirst * nb; t.b_lower_tri = 0; t.small_tile = small ? 1 : 0;
    batch(t, true, true, true);
    return t;
  }
  PotrfLaunch potrf(int64_t k) const {
    PotrfLaunch p;
    const int64_t c0 = k * nb;
    p.nb = (int)nb; p.A = A(c0, c0); p.lda = ld; p.L = L(c0, c0); p.ldl = ld; p.W = W(k);
    p.info = h->d_info; p.col0 = (int)c0;
    p.batch = M.batch; p.sA = M.sA; p.sL = M.sA; p.sW = M.sW; p.s_info = s_info; p.s_flag = h->s_flags;
    return p;
  }
};

Ctx make_ctx(gpx_handle* h, const Mats& M, int64_t tri_base) {
  Ctx c;
  c.h = h; c.M = M; c.rows_total = M.rows_total; c.tri_base = tri_base;
  c.nb = M.nb; c.npad = M.npad; c.ld = M.ld; c.nblk = M.npad / M.nb;
  c.s_info = 2 * h->s_out;
  return c;
}

// lower trapezoid of `nc` columns and `mr` rows (mr >= nc): entries updated by one pass
double trapezoid_entries(int64_t mr, int64_t nc) { return (double)nc * (nc + 1) / 2.0 + (double)(mr - nc) * nc; }

}  // namespace

// ---------------------------------------------------------------------------------------- one stream
// step k:  potrf_diag  A_kk -> L_kk, W_k = inv(L_kk)            (one workgroup per draw, LDS resident)
//          panel       L_ik = A_ik W_k^T      for all rows below  (MFMA GEMM, 32x32 one-pass tiles)
//          column k+1 -= panel k                                  (32x32 one-pass tiles)
//          columns >= k+2 -= panel k                              (64x64 tiles)
// One or two block columns (the multi-stream machinery costs ~20 us of hand-overs on a ~60 us step; measured
// 0.136 -> 0.115 ms per N = 256 step), and the reference schedule of option "lookahead" 0.
static int factor_loop_single(gpx_handle* h, const Ctx& c) {
  hipStream_t P = h->s_main;
  const int64_t nb = c.nb, nblk = c.nblk;
  const double B = (double)c.M.batch;
  for (int64_t k = 0; k < nblk; ++k) {
    const int64_t c0 = k * nb, r0 = c0 + nb, rows = c.rows_at(k);
    prof_begin(h, P, GPX_STAGE_POTRF_DIAG, B * (double)nb * nb * nb / 3.0 * 2.0, B * 3.0 * 8.0 * nb * nb);
    GPX_RET(launch_potrf_diag(h, P, c.potrf(k)));
    prof_end(h, P);
    const int64_t mrows = rows - r0;
    if (mrows <= 0) continue;
    prof_begin(h, P, GPX_STAGE_TRSM_PANEL, B * (double)mrows * nb * nb, B * 8.0 * (2.0 * mrows * nb + nb * nb));
    GPX_RET(launch_gemm_nt(h, P, c.panel(k, r0, mrows, c.L(r0, c0), c.ld, c.M.sA)));
    prof_end(h, P);
    if (k + 1 >= nblk) continue;
    auto upd = [&](int64_t cfirst, int64_t ncolblk, bool small) -> int {
      const int64_t rr = cfirst * nb, mr = rows - rr, nc = ncolblk * nb;
      const double entries = trapezoid_entries(mr, nc);
      prof_begin(h, P, small ? GPX_STAGE_LOOKAHEAD : GPX_STAGE_TRAILING, B * 2.0 * entries * nb,
                 B * 8.0 * (2.0 * entries + (double)(mr + nc) * nb));
      int rc = launch_gemm_nt(h, P, c.update(rr, mr, cfirst, ncolblk, k, 1, small));
      prof_end(h, P);
      return rc;
    };
    GPX_RET(upd(k + 1, 1, true));
    if (nblk - (k + 2) > 0) GPX_RET(upd(k + 2, nblk - (k + 2), false));
  }
  return GPX_OK;
}

// ------------------------------------------------------------------------------- split-panel schedule
// The chain stream P carries ONLY what the next diagonal block needs:
//   P:  potrf(k) -> S = A_{k+1,k} W_k^T (one block row of the panel, 16 workgroups, into scratch)
//                -> A_{k+1,k+1} -= S S^T (10 workgroups) -> potrf(k+1) ...
//   Q:  the full panel k (all rows, incl. block row k+1 again: the official L) -> update of columns k+1
//       (below its diagonal block) and k+2 by panel k
//   T:  bulk update of the columns >= k+3
// so the big panel GEMM (which runs 2-3x slower while the bulk kernel owns the CUs) and one
// cross-stream hand-over leave the critical path: per block column it is potrf + two ~6 us kernels.
// Hand-overs (each an event): W_k P->Q; panel k Q->T; columns k+1,k+2 Q->P (needed by S of step k+1:
// a whole diagonal block later); column k+2 T->Q (bulk of step k-1).  One writer per block at a time:
// block (k+1,k+1) belongs to P at step k (Q's column-k+1 update starts below it).
//
// Grouped bulk updates while the factorisation is bulk-bound: T applies the g most recent panels in ONE
// K = g nb pass every g-th step, cutting the read-modify-write passes over the trailing matrix g-fold (the bulk
// kernel spends about as long on a tile's C traffic as on 128 columns of MFMA work; N = 16384: 42.0 ms with
// g = 1, 36.4 with g = 2).  g is chosen per group from the block columns still to go: 4 above `quad_min`,
// 2 above `pair_min`, else 1 -- in the chain-bound tail a g-fold kernel would sit one step before its
// consumer (N = 4096 with g = 2 throughout: 1.99 -> 2.11 ms).  Invariant after T's step j: every column
// >= j+3 is current through panel j; Q brings column k+2 up to date with the panels since then, so a change
// of g at a group boundary needs no fix-up.
//
// In-kernel hand-overs on the chain ("resident" 1 | 2): potrf_diag(k+1) and the band kernel of step k are launched
// a step early, become resident and sleep on device flags raised by their producers (potrf.hip, gemm.hip): the ~11 us
// from a kernel's last instruction to the start of the next kernel of its stream shrink to a flag round trip.  They
// need the chain's kernels to run concurrently: under per-launch bracketing (profile mode 1) and after a wait ever
// gave up (info = -7: the handle switches "resident" off) the hand-overs are kernel boundaries.
namespace {

struct SplitSchedule {
  gpx_handle* h;
  Ctx c;
  int64_t nb, nblk;
  double B;                       // draws per launch (profiling: flops and bytes scale with it)
  hipStream_t P, P2, Q, T;
  bool resident, chain2, fuse_band, kinv, attach, use_dib, persistent, lp_flag = false;
  unsigned n_signal;
  unsigned *flagsU, *flagsW, *flagsS, *flagsF, *flagsS2, *flagsD, *flagsB;
  // "diagonal block in the band kernel" (fused band kernel, ungrouped steps).  The column update of step j used to
  // cover column j+2 from its diagonal block down, behind the wait for the bulk launch that last wrote that column, and
  // the band kernel of step j+1 waited for all of it -- on alternate block columns of the chain-bound middle of the
  // factorisation that update (slowed 2x by the bulk launch it shares the CUs with) finished 10-15 us after
  // potrf_diag(j+1).  Now the panel stream's update of column j+2 starts BELOW the diagonal block, the hand-over
  // event is recorded behind the update of column j+1 alone, and the band kernel of step j+1 applies panel j to block
  // (j+2,j+2) itself, in a phase 0 that runs while it waits, resident, for potrf_diag(j+1) -- same tile kernel, same
  // order of the two subtractions, bit-identical result.
  // dib[k]: the band kernel of step k has that phase 0 (for panel k-1 on block (k+1,k+1)); band_t[k]: the bulk launch
  // (its step) that last wrote that block and must have finished, or -1.  Both follow from k alone (the grouping is a
  // function of k), so the chain and the side streams agree without talking.
  std::vector<char> dib, far_flag;
  std::vector<int64_t> band_t;
  std::vector<hipEvent_t> tev;        // hand-over event of the bulk launch of step k (set when it is launched)
  std::vector<char> t_pending;
  std::vector<unsigned> t_pub;        // > 0: the bulk launch of step k publishes its first block column (flagsB[k] counts to this)
  int64_t pair_min, quad_min;
  int64_t last_t = -1;        // last group end (panels <= last_t are in every column >= last_t+3: by Q's near block and T's bulk launch)
  int64_t group_end = -1;     // the step at which T acts next
  int64_t q_waited_t = -1;    // the T launch (its step) Q has already waited for
  int64_t t_idx = -1;         // step of the most recent bulk launch on T
  int64_t t_from;             // first block column that launch wrote: Q waits for it only before touching a column >= t_from
  int64_t kinv_last = -1;     // last panel folded into K^-1 (gradient path)
  bool far_attached = false;  // per step: ev_far[k] rides on a kernel's own dispatch packet
  int64_t force_timeout_at = -1;   // test hook ("debug_force_timeout"): this step's potrf_diag waits for an impossible count
  int64_t near_step = -1;     // step of a near-block launch on Q the chain has not been ordered behind yet (h->ev_near)

  int64_t group_size(int64_t left) const {
    return (quad_min > 0 && left > quad_min) ? 4 : ((pair_min > 0 && left > pair_min) ? 2 : 1);
  }

  int init(gpx_handle* hh, const Mats& M, int64_t tri_base) {
    h = hh; c = make_ctx(h, M, tri_base);
    nb = c.nb; nblk = c.nblk; B = (double)M.batch;
    P = h->s_main; Q = h->s_far; T = h->s_trail;
    // Batches of more than four draws hand over at kernel boundaries: a resident consumer holds its CU slots (and LDS)
    // while it sleeps, and its grid grows with the batch -- 16 band workgroups per draw park 34 KB of LDS each on as many
    // CUs, and from 16 draws on no CU is left with the 144 KB potrf_diag needs (measured: the wait times out, info = -7).
    // Up to four draws the chain still matters (N = 4096, two draws: 1.57 ms per draw with kernel boundaries, 1.33 resident)
    // and 64 sleeping band workgroups + 4 potrf_diag leave the GPU to their producers; the last panel's in-kernel wait
    // (border tiles x draws waiting workgroups) stays a single-draw device.  The bits do not depend on the hand-over mode.
    resident = h->resident != 0 && h->prof_mode != 1 && nblk >= 3 && M.batch <= 4;
    P2 = resident ? h->s_main2 : P;
    chain2 = resident && h->resident >= 2;
    fuse_band = chain2;
    persistent = chain2 && h->resident >= 3;   // ONE potrf_diag workgroup per draw for the whole factorisation (potrf.hip)
    kinv = h->kinv_U != nullptr;
    // completion events ride on the kernels' own dispatch packets (hipExtLaunchKernelGGL stopEvent) except under
    // per-launch profiling, where they are recorded behind the kernel as usual
    attach = h->prof_mode != 1;
    // the diagonal block (k+1,k+1) gets panel k-1 from the chain itself, in EVERY hand-over mode (phase 0 of the fused band
    // kernel, or a launch of its own in front of the band kernels): the arithmetic -- which tile routine touches which
    // tile, in which order -- must not depend on the mode, or a batched call (kernel-boundary hand-overs) and a
    // single-draw call (in-kernel hand-overs) would differ in the last bits
    use_dib = !kinv;
    n_signal = (unsigned)((nb / 32) * (nb / 32 + 1) / 2);   // lower 32x32 tiles of a diagonal block
    flagsU = h->d_flags;              // [k]: tiles of A_{k+1,k+1} -= S S^T published (potrf_diag(k+1) waits)
    flagsW = h->d_flags + nblk;       // [k]: W_k published (S = A W_k^T of step k waits)
    flagsS = h->d_flags + 2 * nblk;   // [k]: tiles of S published (fused band kernel: its second phase waits)
    flagsF = h->d_flags + 3 * nblk;   // [k]: first block rows of the panel stream's column updates of step k published (row 2 of the band kernel of step k+1 waits)
    flagsS2 = h->d_flags + 4 * nblk;  // [k]: tiles of S2 published (row 2 of the fused band kernel)
    flagsB = h->d_flags + 6 * nblk;   // [k]: tiles of the first block column of the bulk launch of step k published (Q's next column update, the band kernel)
    flagsD = h->d_flags + 5 * nblk;   // [k]: tiles of A_{k+2,k+1} published by row 2 (the first block row of Q's panel k+1 waits)
    // the gradient path doubles the bulk stream's work (identity border rows, K^-1 accumulation): it is bound by
    // that stream almost to the end, so its bulk updates are grouped down to 8 remaining block columns
    pair_min = kinv ? std::min<int64_t>(h->pair_min, 8) : h->pair_min; quad_min = h->quad_min;
    // more test points than training points (config C5: 10 000 candidates against N = 8192): the border rows are most of every
    // bulk launch and the step is bound by that stream to the end, like the gradient path's -- grouped down to 8 / 16 remaining
    // block columns: 17.5 -> 16.9-17.0 ms per step at C5's shape (tools/c5_opts.py); at M = N it is a tie
    if (!kinv && tri_base < 0 && M.rows_total - M.npad >= M.npad) { pair_min = std::min<int64_t>(pair_min, 8); quad_min = std::min<int64_t>(quad_min, 16); }
    // Batches of four or more draws are throughput-bound from the first step to the last: the chain's latency hides behind
    // the other draws' work, so the bulk updates are grouped throughout (K = g nb passes run the bulk kernel at 48-53
    // instead of 40 TFLOP/s and cut the passes over the trailing matrices g-fold).  Measured at N = 4096, 16 draws:
    // 40.5 -> 45.7 TFLOP/s aggregate.  The grouping is what a single-draw call does with "pair_min" = g / 4,
    // "quad_min" = g / 2, bit for bit; "batch_group" = 0 keeps the single-draw thresholds.
    if (M.batch >= 4 && h->batch_group >= 2 && !kinv) {
      pair_min = 1;
      quad_min = h->batch_group >= 4 ? 2 : 0;
    }
    t_from = nblk;
    dib.assign((size_t)nblk + 2, 0); band_t.assign((size_t)nblk + 2, -1); far_flag.assign((size_t)nblk + 2, 0);
    tev.assign((size_t)nblk, nullptr); t_pending.assign((size_t)nblk, 0); t_pub.assign((size_t)nblk, 0u);
    if (resident && h->debug_force_timeout) { force_timeout_at = 1; h->debug_force_timeout = 0; }
    if (resident && !h->flags_clean) {
      if (M.batch == 1) GPX_HIP_CHECK(h, hipMemsetAsync(h->d_flags, 0, (size_t)(GPX_N_FLAG_ARRAYS * nblk) * sizeof(unsigned), P));
      else GPX_HIP_CHECK(h, hipMemsetAsync(h->d_flags, 0, (size_t)(h->s_flags * M.batch) * sizeof(unsigned), P));
    }
    h->flags_clean = false;
    GPX_HIP_CHECK(h, hipEventRecord(h->ev_join, P));
    GPX_HIP_CHECK(h, hipStreamWaitEvent(Q, h->ev_join, 0));
    GPX_HIP_CHECK(h, hipStreamWaitEvent(T, h->ev_join, 0));
    if (resident) GPX_HIP_CHECK(h, hipStreamWaitEvent(P2, h->ev_join, 0));
    // gradient path: K^-1 is accumulated on T while the chain runs (see kinv_update below); T is idle until the
    // first panel is complete, so clearing the accumulator there costs nothing
    if (kinv) GPX_HIP_CHECK(h, hipMemsetAsync(h->d_kinv, 0, (size_t)(M.batch > 1 ? h->s_kinv * M.batch : h->kinv_rows * c.npad) * sizeof(double), T));
    plan_band();
    return GPX_OK;
  }

  // lp_flag: the last border panel (chain stream) is ordered behind the panel stream's last column update by a device
  // counter that update's tiles raise (flagsS[nblk-1], otherwise unused) instead of a stream-level event wait in front
  // of the last potrf_diag, which kept that kernel from starting resident (+17 us at the tail of every step)
  void plan_band() {
    if (kinv) return;
    int64_t lt = -1, ge = -1, ti = -1, tf = nblk;
    for (int64_t j = 0; j < nblk; ++j) {
      if (j > ge) ge = j + group_size(nblk - j) - 1;
      const int64_t rows = c.rows_at(j);
      if (rows - (j + 1) * nb <= 0 || j + 1 >= nblk) continue;
      int64_t near = 0;
      if (j == ge) {
        const int64_t gn = group_size(nblk - (j + 1));
        if (gn >= 2) near = std::max<int64_t>(0, std::min<int64_t>(gn, nblk - (j + 3)));
      }
      const int64_t mr = rows - (j + 2) * nb, ncb = std::min<int64_t>(2, nblk - (j + 1));
      if (chain2 && c.M.batch == 1 && j == nblk - 2 && j - lt == 1 && mr > 0 && (h->la_rows <= 0 || mr < h->la_rows) && (nb == 128 || nb == 64)) lp_flag = true;
      if (use_dib && mr > 0 && ncb > 1 && j - lt == 1) {
        dib[(size_t)(j + 1)] = 1;
        band_t[(size_t)(j + 1)] = (ti >= 0 && j + 2 >= tf) ? ti : -1;
        // far_flag[j] ("row 2", round 3): the band kernel of step j owns a window of TWO block rows -- it also forms
        // S2 = A_{j+2,j} W_j^T and applies A_{j+2,j+1} -= S2 S1^T, the block that is the NEXT band kernel's operand, and its S2 is
        // the next band kernel's phase-0 operand -- so that step j+1's critical phases need nothing from the panel stream.  The
        // panel stream's column updates of step j then start one block lower (block row j+3) and publish their first block rows
        // (flagsF[j], both the one-pass and the 64x64 look-ahead kernel can): row 2 of step j+1 waits for them inside the kernel,
        // with a whole potrf_diag of slack.  (Round 2 handed A_{j+2,j+1} itself over through flagsF, and only where the update
        // ran on the one-pass kernel; measured, that publication arrived within 0-6 us of potrf_diag(j+1)'s end in the middle of
        // the factorisation, and phase 0 + S then ran behind it: gaps of 25-32 us between two potrf_diag against 11.)
        if (fuse_band && j + 2 < nblk && (nb == 128 || nb == 64)) far_flag[(size_t)j] = 1;
      }
      if (j == ge) {
        if (nblk - (j + 3 + near) > 0) { ti = j; tf = j + 3 + near; }
        lt = j;
      }
    }
  }

  // ---- the chain at step k: potrf_diag(k), then the diagonal band of step k
  int chain(int64_t k) {
    // resident == 1: consecutive steps alternate between the two chain streams; resident == 2: potrf_diag on
    // s_main, the band kernels on P2
    hipStream_t PD = chain2 ? h->s_main : ((resident && (k & 1)) ? P2 : h->s_main);
    hipStream_t PB = chain2 ? P2 : PD;   // stream of the two band kernels
    double* S = h->d_mini + (k & 1) * nb * nb;
    const int64_t c0 = k * nb, r0 = c0 + nb, rows = c.rows_at(k);
    // ev_w[k] = completion of potrf_diag(k): attached to the kernel's own dispatch when possible (no barrier
    // packet behind it), else recorded right after it -- NOT after the two band kernels (measured 1.91 -> 2.12 ms
    // per N = 4096 step when the panel stream learns of W_k 12 us later)
    const bool need_w = rows - r0 > 0;
    if (need_w && k + 1 >= nblk && !kinv && k >= 1 && !lp_flag) {
      // The last panel (below) reads the border rows of column k on THIS stream; their last writer is Q's column update
      // of step k-1.  Nothing else orders the two (potrf_diag(k) only waits for the band kernel of step k-1, which
      // waited for Q's step k-2): wait for it here, in front of potrf_diag, where it costs nothing -- that update is
      // normally long finished.  (Without it the race was real at two block columns once the border was built on
      // the panel stream: one wrong posterior in ~200 runs of the test suite.)
      GPX_HIP_CHECK(h, hipStreamWaitEvent(PD, h->ev_far[(size_t)(k - 1)], 0));
    }
    if (persistent) {
      if (k == 0) {
        PotrfChainLaunch q;
        q.nb = (int)nb; q.A = c.A(0, 0); q.lda = c.ld; q.L = c.L(0, 0); q.ldl = c.ld; q.W = c.W(0); q.info = h->d_info;
        q.flagsU = flagsU; q.flagsW = flagsW; q.k0 = 0; q.k1 = (int)nblk; q.raise_until = (int)nblk; q.force_k = (int)force_timeout_at;
        q.wait_count = n_signal;
        q.batch = c.M.batch; q.sA = c.M.sA; q.sL = c.M.sA; q.sW = c.M.sW; q.s_info = c.s_info; q.s_flag = h->s_flags;
        prof_begin(h, PD, GPX_STAGE_POTRF_DIAG, B * (double)nblk * nb * nb * nb / 3.0 * 2.0, B * (double)nblk * 3.0 * 8.0 * nb * nb);
        GPX_RET(launch_potrf_chain(h, PD, q));
        prof_end(h, PD);
      }
    } else {
      PotrfLaunch p = c.potrf(k);
      if (need_w && attach) p.stop_event = h->ev_w[(size_t)k];
      if (resident && k >= 1) { p.wait_flag = flagsU + (k - 1); p.wait_count = n_signal; }
      if (chain2 && need_w && k + 1 < nblk) p.done_flag = flagsW + k;
      if (k == force_timeout_at && p.wait_flag) { p.wait_count = 0xffffffffu; p.wait_ticks = 2000ull; }
      prof_begin(h, PD, GPX_STAGE_POTRF_DIAG, B * (double)nb * nb * nb / 3.0 * 2.0, B * 3.0 * 8.0 * nb * nb);
      GPX_RET(launch_potrf_diag(h, PD, p));
      prof_end(h, PD);
    }
    if (!need_w) return GPX_OK;
    if (k + 1 >= nblk && !kinv) {
      // the last panel is the border rows only and nothing follows it but the epilogue: it runs on the chain stream,
      // behind potrf_diag, instead of hopping to the panel stream and back (two ~8 us event hand-overs)
      GemmLaunch g = c.panel(k, r0, rows - r0, c.L(r0, c0), c.ld, c.M.sA);
      if (lp_flag && k >= 1) {   // its input rows were last written by the panel stream's column update of step k-1: wait for that update's tiles
        g.wait_flag = flagsS + k; g.wait_count = (unsigned)((nb / 32) * ((rows - r0) / 32)); g.wait_info = h->d_info;
      }
      prof_begin(h, PD, GPX_STAGE_TRSM_PANEL, B * (double)g.M * nb * nb, B * 8.0 * (2.0 * g.M * nb + nb * nb));
      GPX_RET(launch_gemm_nt(h, PD, g));
      prof_end(h, PD);
      return GPX_OK;   // (the gradient path keeps its last panel on Q: the K^-1 update behind it belongs to T's sequence)
    }
    if (!attach && !persistent) GPX_HIP_CHECK(h, hipEventRecord(h->ev_w[(size_t)k], PD));
    if (k + 1 >= nblk) return GPX_OK;
    const bool far_by_flag = k >= 1 && far_flag[(size_t)(k - 1)] != 0;
    if (k >= 1 && !far_by_flag) GPX_HIP_CHECK(h, hipStreamWaitEvent(PB, h->ev_far[(size_t)(k - 1)], 0));   // columns k, k+1 were brought up to date by Q at step k-1
    // The near block of group end j (columns j+3 .. j+2+near, diagonal blocks included) is written by Q behind ev_far[j].
    // The first band kernel that touches one of those columns is the one of step j+2; where it has no stream-level wait
    // for Q (far_by_flag: its critical phases need nothing else from Q since round 3 -- round 2's in-kernel wait for Q's
    // step-(k-1) updates ordered it implicitly) it waits for the near block's own event.
    if (near_step >= 0 && k >= near_step + 2) {
      if (far_by_flag) GPX_HIP_CHECK(h, hipStreamWaitEvent(PB, h->ev_near, 0));
      near_step = -1;   // (otherwise ev_far[k-1], k-1 > near_step, was recorded behind it)
    }
    const bool phase0 = dib[(size_t)k] != 0;   // block (k+1,k+1) -= L_{k+1,k-1} L_{k+1,k-1}^T (panel k-1), see dib[] above
    // the bulk launch that last wrote the block: its completion event -- or, where that launch publishes its first block
    // column (which holds the block), the band kernel's phase 0 waits for the counter itself
    const int64_t bt = phase0 ? band_t[(size_t)k] : -1;
    const bool bulk_by_flag = bt >= 0 && fuse_band && t_pub[(size_t)bt] > 0;
    if (bt >= 0 && !bulk_by_flag) GPX_HIP_CHECK(h, hipStreamWaitEvent(PB, tev[(size_t)bt], 0));
    if (fuse_band) {   // the whole band (phase 0, S1, the diagonal block, and row 2) in one launch
      BandLaunch b;
      b.nb = (int)nb; b.W = c.W(k); b.ld = c.ld;
      b.A1 = c.A(r0, c0); b.D11 = c.A(r0, r0);
      b.row2 = far_flag[(size_t)k] != 0;
      if (b.row2) { b.A2 = c.A(r0 + nb, c0); b.D21 = c.A(r0 + nb, r0); b.big = col_small(k) ? 0 : 1; }
      b.S1 = S; b.S2 = h->d_mini + (2 + (k & 1)) * nb * nb;
      if (phase0) {
        // its operand L_{k+1,k-1}: the S2 the previous band kernel left in scratch (same stream: ordered), or -- where that
        // kernel had no row 2 -- the official copy of the panel stream (ordered by the event wait above)
        if (far_by_flag) { b.P0 = h->d_mini + (2 + ((k - 1) & 1)) * nb * nb; b.ldp = nb; b.sP0 = h->s_mini; }
        else { b.P0 = c.L(r0, (k - 1) * nb); b.ldp = c.ld; b.sP0 = c.M.sA; }
      }
      b.flagW = flagsW + k; b.flagS = flagsS + k; b.flagS2 = flagsS2 + k; b.flagU = flagsU + k; b.flagD = flagsD + k; b.info = h->d_info;
      if (b.row2 && far_by_flag) { b.flagF = flagsF + (k - 1); b.countF = far_count(k - 1); }
      if (bulk_by_flag) { b.flagB = flagsB + bt; b.countB = t_pub[(size_t)bt]; }
      b.batch = c.M.batch; b.sA = c.M.sA; b.sW = c.M.sW; b.s_mini = h->s_mini; b.s_flag = h->s_flags; b.s_info = c.s_info;
      prof_begin(h, PB, GPX_STAGE_LOOKAHEAD, B * ((phase0 ? 3.0 : 2.0) + (b.row2 ? 2.0 : 0.0)) * (double)nb * nb * nb, B * 8.0 * 6.0 * nb * nb);
      GPX_RET(launch_chain_band(h, PB, b));
      prof_end(h, PB);
      return GPX_OK;
    }
    GemmLaunch g = c.panel(k, r0, nb, S, nb, h->s_mini);                                    // S = A_{k+1,k} W_k^T
    GemmLaunch u;                                                                           // A_{k+1,k+1} -= S S^T
    u.A = S; u.lda = nb; u.B = S; u.ldb = nb;
    u.C = c.A(r0, r0); u.ldc = c.ld;
    u.M = nb; u.N = nb; u.K = nb; u.alpha = -1.0; u.beta = 1.0;
    u.lower = 1; u.row_off = 0; u.col_off = 0; u.b_lower_tri = 0; u.small_tile = 1;
    u.batch = c.M.batch; u.sA = h->s_mini; u.sB = h->s_mini; u.sC = c.M.sA; u.s_flag = h->s_flags; u.s_info = c.s_info;
    if (resident) u.signal = flagsU + k;   // potrf_diag(k+1) is already resident and polls this counter
    GemmLaunch z;
    if (phase0) {
      z = c.update(r0, nb, k + 1, 1, k - 1, 1, true);
      z.row_off = 0; z.col_off = 0;
    }
    if (phase0) {
      prof_begin(h, PB, GPX_STAGE_LOOKAHEAD, B * (double)nb * nb * nb, B * 8.0 * 3.0 * nb * nb);
      GPX_RET(launch_gemm_nt(h, PB, z));
      prof_end(h, PB);
    }
    prof_begin(h, PB, GPX_STAGE_LOOKAHEAD, B * (double)nb * nb * nb, B * 8.0 * 3.0 * nb * nb);
    GPX_RET(launch_gemm_nt(h, PB, g));
    prof_end(h, PB);
    prof_begin(h, PB, GPX_STAGE_LOOKAHEAD, B * (double)nb * nb * nb, B * 8.0 * 3.0 * nb * nb);
    GPX_RET(launch_gemm_nt(h, PB, u));
    prof_end(h, PB);
    return GPX_OK;
  }

  // Grouped phases: the bulk launch at a group end leaves the NEXT group's columns (the "near block", as many block
  // columns as that group has panels) to Q, which applies the same panels to them in one K = g nb launch of its own.
  // The chain of the next group then depends on that short launch only and runs WHILE the bulk launch does; with
  // the bulk launch covering everything from column k+3 the chain stalled at the first step of every group until
  // the bulk launch had finished and the bulk stream then idled for the 2-4 chain steps that produce its next
  // panels (N = 16384: ~270 us of every ~2.3 ms).
  int64_t near_cols_at(int64_t k) const {
    if (k != group_end || kinv) return 0;
    const int64_t g_next = group_size(nblk - (k + 1));
    return g_next >= 2 ? std::max<int64_t>(0, std::min<int64_t>(g_next, nblk - (k + 3))) : 0;
  }

  // ---- Q: the whole panel k (all rows below the diagonal block, border rows included)
  int panel(int64_t k, int64_t near_cols) {
    const int64_t c0 = k * nb, r0 = c0 + nb, rows = c.rows_at(k), mrows = rows - r0;
    // (Starting the panel kernel resident too -- it would sleep on potrf_diag's flag instead of waiting ~8 us for the
    // cross-stream event -- crowds the bulk update with sleeping workgroups: 1.76 -> 1.81 ms, not used.)
    // W_k: the completion event of potrf_diag(k), or -- where one workgroup walks the whole diagonal -- a one-wave relay
    // kernel that sleeps on flagsW[k]; the kernel boundary behind it releases the panel
    if (persistent) GPX_RET(launch_chain_relay(h, Q, flagsW + k, 1u, h->d_info, c.M.batch, h->s_flags, c.s_info));
    else GPX_HIP_CHECK(h, hipStreamWaitEvent(Q, h->ev_w[(size_t)k], 0));
    GemmLaunch g = c.panel(k, r0, mrows, c.L(r0, c0), c.ld, c.M.sA);
    if (k >= 1 && far_flag[(size_t)(k - 1)]) {
      // its first block row, A_{k+1,k}, was last written by row 2 of the band kernel of step k-1 (another stream): those
      // tiles are published and counted; this panel starts behind potrf_diag(k), so the count is there and one poll passes
      g.wait_flag = flagsD + (k - 1); g.wait_count = (unsigned)((nb / 32) * (nb / 32)); g.wait_m = nb; g.wait_info = h->d_info;
    }
    // T gets its own event on the panel: the bulk update must not wait for the column kernels below (with one
    // shared event after them the N = 4096 step went from 1.90 to 2.17 ms: every bulk update starts late)
    const bool t_needs = (k + 1 < nblk && k == group_end && nblk - (k + 3 + near_cols) > 0) || (kinv && (k == group_end || k + 1 >= nblk));
    if (t_needs && attach) g.stop_event = h->ev_panel[(size_t)k];
    prof_begin(h, Q, GPX_STAGE_TRSM_PANEL, B * (double)mrows * nb * nb, B * 8.0 * (2.0 * mrows * nb + nb * nb));
    GPX_RET(launch_gemm_nt(h, Q, g));
    prof_end(h, Q);
    if (t_needs && !attach) GPX_HIP_CHECK(h, hipEventRecord(h->ev_panel[(size_t)k], Q));
    // every block column but the last is final now (V^T, z and the diagonal of L): their share of the epilogue's
    // reductions runs here, on the panel stream, beside the last potrf_diag instead of behind the last panel
    if (h->epi && k == nblk - 2 && nblk >= 3 && h->d_epart) {
      GPX_RET(launch_epilogue_partial(h, Q, *h->epi, (nblk - 1) * nb, h->d_epart));
      h->epi_split_col = (nblk - 1) * nb;
    }
    return GPX_OK;
  }

  // K^-1 += U_{:,j} U_{:,j}^T for the panels j = kinv_last+1 .. k that are complete now (gradient path): rank-(npan nb)
  // update of the leading (k+1) nb square, lower tiles, on T.  Small while the bulk updates are big and growing as
  // they shrink, it fills the chain-bound second half of the factorisation with the N^3/3 flops that used to
  // follow it as one U U^T product (2.9 -> 2.2 ms per gradient call at N = 4096).
  int kinv_update(int64_t k) {
    const int64_t kfirst = kinv_last + 1, npan = k - kinv_last;
    kinv_last = k;
    const int64_t msq = std::min<int64_t>(h->kinv_rows, (k + 1) * nb);
    GemmLaunch t;
    t.A = h->kinv_U + kfirst * nb; t.lda = c.ld;
    t.B = h->kinv_U + kfirst * nb; t.ldb = c.ld;
    t.C = h->d_kinv; t.ldc = c.npad;
    t.M = msq; t.N = msq; t.K = npan * nb; t.alpha = 1.0; t.beta = 1.0;
    t.lower = 1; t.row_off = 0; t.col_off = 0; t.b_lower_tri = 0; t.small_tile = 0;
    t.batch = c.M.batch; t.sA = c.M.sA; t.sB = c.M.sA; t.sC = h->s_kinv; t.s_flag = h->s_flags; t.s_info = c.s_info;   // U lives in the factor matrices
    const double entries = (double)msq * (msq + 1) / 2.0;
    prof_begin(h, T, GPX_STAGE_TRAILING, B * 2.0 * entries * t.K, B * 8.0 * (2.0 * entries + 2.0 * (double)msq * t.K));
    int rc = launch_gemm_nt(h, T, t);
    prof_end(h, T);
    return rc;
  }

  // tall columns (the bulk-bound phase of a large factorisation: no hurry, the machine is full) on 64x64 tiles,
  // which move half the operand bytes per flop; the one-pass 32x32 tiles are for latency.  The choice follows the
  // STEP (rows below block row k+2), not the launch: a column whose update starts one block lower in one hand-over
  // mode ("diagonal block in the band kernel") must get the same tile routine -- the same bits -- in every mode.
  bool col_small(int64_t k) const { return !(h->la_rows > 0 && c.rows_at(k) - (k + 2) * nb >= h->la_rows); }
  // what the two column updates of step k (far_flag[k]: both start at block row k+3) add to flagsF[k]: the tiles of
  // their first block row, 32x32 or 64x64 ones
  unsigned far_count(int64_t k) const {
    const int64_t mr = c.rows_at(k) - (k + 3) * nb, tw = col_small(k) ? 32 : 64;
    if (mr <= 0) return 0u;
    return (unsigned)(2 * ((std::min<int64_t>(nb, mr) + tw - 1) / tw) * (nb / tw));
  }

  // Q: columns [cfirst, cfirst+ncol) blocks, rows from block row k+2 (+ skip_blocks) down, -= panels [kfirst, kfirst+npan)
  int col_update(int64_t k, int64_t cfirst, int64_t ncol, int64_t kfirst, int64_t npan, bool last, int64_t skip_blocks = 0, bool raise_far = false) {
    const int64_t rows = c.rows_at(k), rr = (k + 2 + skip_blocks) * nb, mr = rows - rr;
    if (mr <= 0) return GPX_OK;
    GemmLaunch t = c.update(rr, mr, cfirst, ncol, kfirst, npan, col_small(k));
    if (last && attach) { t.stop_event = h->ev_far[(size_t)k]; far_attached = true; }
    if (lp_flag && k == nblk - 2 && cfirst == nblk - 1) t.signal = flagsS + (nblk - 1);   // the last border panel waits for these tiles (chain stream)
    if (raise_far) { t.signal = flagsF + k; t.signal_m = nb; }   // the band kernel of step k+1 waits for the tiles of the first block row
    t.role = 1;
    double entries = 0.0;
    for (int64_t cc = cfirst; cc < cfirst + ncol; ++cc)
      entries += (cc * nb >= rr) ? (double)nb * (nb + 1) / 2.0 + (double)(mr - nb) * nb : (double)mr * nb;
    prof_begin(h, Q, GPX_STAGE_LOOKAHEAD, B * 2.0 * entries * t.K, B * 8.0 * (2.0 * entries + (double)(mr + ncol * nb) * t.K));
    int rc = launch_gemm_nt(h, Q, t);
    prof_end(h, Q);
    return rc;
  }

  int q_wait_for_bulk() {
    if (t_idx >= 0 && t_idx != q_waited_t && t_pending[(size_t)t_idx]) {
      GPX_HIP_CHECK(h, hipStreamWaitEvent(Q, tev[(size_t)t_idx], 0));
      q_waited_t = t_idx;
    }
    return GPX_OK;
  }

  // ---- Q keeps two columns current ahead of the chain: column k+1 (below its diagonal block) gets panel k, column
  // k+2 every panel since T's last step; rows from block row k+2 down.  (A third window column, which takes T
  // off everything P waits for, was tried with and without grouped updates: 1.94-2.02 vs 1.88 ms at N = 4096.)
  int columns(int64_t k) {
    const int64_t rows = c.rows_at(k), mr = rows - (k + 2) * nb;
    const int64_t ncb = std::min<int64_t>(2, nblk - (k + 1));
    far_attached = false;
    if (mr > 0) {
      const int64_t npan2 = k - last_t;   // panels column k+2 still lacks: last_t+1 .. k
      // column k+2 was last written by the bulk launch of step t_idx -- if it lies in its range
      const bool col2_in_bulk = k + 2 >= t_from;
      if (dib[(size_t)(k + 1)]) {
        // column k+1 alone carries the hand-over to the chain; column k+2 from BELOW its diagonal block, behind the
        // wait for the bulk launch (the band kernel of step k+1 applies panel k to that block itself)
        // far_flag[k]: block (k+2,k+1) belongs to row 2 of this step's band kernel; both updates start at block row k+3 and
        // publish their first block row to row 2 of the NEXT band kernel
        const bool rf = far_flag[(size_t)k] != 0;
        GPX_RET(col_update(k, k + 1, 1, k, 1, true, rf ? 1 : 0, rf));
        if (!far_attached) GPX_HIP_CHECK(h, hipEventRecord(h->ev_far[(size_t)k], Q));
        far_attached = true;   // recorded (or attached) already: not again behind the second column
        // column k+2 is the first block column of the most recent bulk launch: wait for THAT (a one-wave relay kernel on
        // the launch's counter) where the launch publishes it, not for the launch to end -- round 2's event wait here closed
        // a loop  bulk(k-1) ends -> this update -> panel(k+1) -> bulk(k+1)  that was longer than a bulk launch (T idled
        // ~10 us in front of every launch of the bulk-bound phase, profiles/r03_critical_path_persist.txt)
        if (col2_in_bulk) {
          if (t_idx >= 0 && t_pub[(size_t)t_idx] > 0 && t_from == k + 2) {
            if (t_idx != q_waited_t) GPX_RET(launch_chain_relay(h, Q, flagsB + t_idx, t_pub[(size_t)t_idx], h->d_info, c.M.batch, h->s_flags, c.s_info));
          } else GPX_RET(q_wait_for_bulk());
        }
        GPX_RET(col_update(k, k + 2, 1, k, 1, false, 1, rf));
      } else if (ncb > 1 && npan2 > 1) {
        // grouped phases: column k+1 (whose last bulk update Q waited for a step ago) needs no bulk update to finish,
        // only column k+2 does
        GPX_RET(col_update(k, k + 1, 1, k, 1, false));
        if (col2_in_bulk) GPX_RET(q_wait_for_bulk());
        GPX_RET(col_update(k, k + 2, 1, last_t + 1, npan2, true));
      } else {
        if (col2_in_bulk) GPX_RET(q_wait_for_bulk());
        GPX_RET(col_update(k, k + 1, ncb, k, 1, true));
      }
    }
    if (!far_attached) GPX_HIP_CHECK(h, hipEventRecord(h->ev_far[(size_t)k], Q));
    return GPX_OK;
  }

  // ---- Q: the near block (columns k+3 .. k+2+near_cols, panels last_t+1 .. k), behind the event the chain waits for
  int near_block(int64_t k, int64_t near_cols) {
    GPX_RET(q_wait_for_bulk());   // those columns were last written by the previous bulk launch
    const int64_t rows = c.rows_at(k), kfirst = last_t + 1, npan = k - last_t;
    const int64_t br = (k + 3) * nb, bm = rows - br, nc = near_cols * nb;
    GemmLaunch t = c.update(br, bm, k + 3, near_cols, kfirst, npan, false);
    t.role = 1;
    const double entries = trapezoid_entries(bm, nc);
    prof_begin(h, Q, GPX_STAGE_LOOKAHEAD, B * 2.0 * entries * t.K, B * 8.0 * (2.0 * entries + (double)(bm + nc) * t.K));
    GPX_RET(launch_gemm_nt(h, Q, t));
    prof_end(h, Q);
    GPX_HIP_CHECK(h, hipEventRecord(h->ev_near, Q));
    near_step = k;
    return GPX_OK;
  }

  // ---- T: bulk, columns >= k+3 (+ the near block), panels last_t+1 .. k in one pass
  int bulk(int64_t k, int64_t near_cols) {
    const int64_t rows = c.rows_at(k), nbulk = nblk - (k + 3 + near_cols);
    if (nbulk > 0) {
      GPX_HIP_CHECK(h, hipStreamWaitEvent(T, h->ev_panel[(size_t)k], 0));
      const int64_t kfirst = last_t + 1, npan = k - last_t;
      const int64_t cf = k + 3 + near_cols, br = cf * nb, bm = rows - br, nc = nbulk * nb;
      GemmLaunch t = c.update(br, bm, cf, nbulk, kfirst, npan, false);
      // ungrouped steps of the fused chain: the first block column (the only part of this launch the next step's column
      // update and band kernel need) is enumerated first and publishes itself
      if (fuse_band && !kinv && npan == 1 && near_cols == 0 && nbulk >= 2 && nb % 64 == 0 && h->bulk_first_col) {
        const int64_t fc = nb / 64, nti = bm / 64;
        t.first_col = (int)fc; t.signal = flagsB + k;
        t_pub[(size_t)k] = (unsigned)(fc * nti - fc * (fc - 1) / 2);
      }
      const double entries = trapezoid_entries(bm, nc);
      // profiled: the kernel's own dispatch timestamps (what rocprofv3 reports), the hand-over event recorded behind it
      const bool timed = prof_pair(h, GPX_STAGE_TRAILING, B * 2.0 * entries * t.K, B * 8.0 * (2.0 * entries + (double)(bm + nc) * t.K),
                                   &t.start_event, &t.stop_event);
      // timed (profile mode 3): the timing stop event, attached to the kernel's own dispatch, IS the hand-over event -- a
      // second record behind the launch delayed every consumer of the bulk update and stretched the measured step by 10 %
      if (attach && !timed) t.stop_event = h->ev_trail[(size_t)k];
      tev[(size_t)k] = (attach && timed) ? t.stop_event : h->ev_trail[(size_t)k];
      GPX_RET(launch_gemm_nt(h, T, t));
      if (!attach) GPX_HIP_CHECK(h, hipEventRecord(h->ev_trail[(size_t)k], T));
      t_pending[(size_t)k] = 1;
      t_idx = k; t_from = cf;
    }
    // behind the bulk update (Q's next column update waits for that one, not for this), two panels at a time:
    // nothing waits for K^-1 before the end and K = 2 nb passes run at 48 instead of 40 TFLOP/s (groups of 4 or 8
    // leave a long last pass exposed behind the chain: 2.58 / 2.68 / 2.85 ms per call at N = 4096)
    if (kinv && (k - kinv_last >= 2 || k + 2 >= nblk)) {
      if (nbulk <= 0) GPX_HIP_CHECK(h, hipStreamWaitEvent(T, h->ev_panel[(size_t)k], 0));
      GPX_RET(kinv_update(k));
    }
    last_t = k;
    return GPX_OK;
  }

  // ---- the side streams at step k
  int side(int64_t k) {
    if (k > group_end) group_end = k + group_size(nblk - k) - 1;
    const int64_t rows = c.rows_at(k), mrows = rows - (k + 1) * nb;
    if (mrows <= 0) return GPX_OK;
    if (k + 1 >= nblk && !kinv) return GPX_OK;   // the last panel (border rows only) is launched by the chain itself
    const int64_t near_cols = near_cols_at(k);
    GPX_RET(panel(k, near_cols));
    if (k + 1 >= nblk) {   // border rows only: nothing left to update
      if (kinv) {
        GPX_HIP_CHECK(h, hipStreamWaitEvent(T, h->ev_panel[(size_t)k], 0));
        GPX_RET(kinv_update(k));
      }
      return GPX_OK;
    }
    GPX_RET(columns(k));
    if (near_cols > 0) GPX_RET(near_block(k, near_cols));
    if (k == group_end) GPX_RET(bulk(k, near_cols));
    return GPX_OK;
  }

  int finish() {
    if (resident) {
      GPX_HIP_CHECK(h, hipEventRecord(h->ev_chain2, P2));
      GPX_HIP_CHECK(h, hipStreamWaitEvent(P, h->ev_chain2, 0));
    }
    GPX_HIP_CHECK(h, hipEventRecord(h->ev_join, T));
    GPX_HIP_CHECK(h, hipStreamWaitEvent(P, h->ev_join, 0));
    GPX_HIP_CHECK(h, hipEventRecord(h->ev_join2, Q));
    GPX_HIP_CHECK(h, hipStreamWaitEvent(P, h->ev_join2, 0));
    return GPX_OK;
  }
};

}  // namespace

#ifdef GPX_TIMELINE
#include <chrono>
// diagnostic build only: host time (us since the loop began) at which step k's launches were enqueued: [2k] chain, [2k+1] side
double g_tl_host[1024];
int g_tl_host_n = 0;
extern "C" int gpx_timeline_host(double* out, int n) {
  const int m = n < g_tl_host_n ? n : g_tl_host_n;
  for (int i = 0; i < m; ++i) out[i] = g_tl_host[i];
  return m;
}
#define TL_HOST(i) do { if ((i) < 1024) { g_tl_host[(i)] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tl_t0).count(); g_tl_host_n = (i) + 1; } } while (0)
#else
#define TL_HOST(i) do { } while (0)
#endif

int factor_loop(gpx_handle* h, const Mats& M, int64_t tri_base) {
  const int64_t nblk = M.npad / M.nb;
  if (h->lookahead == 0 || nblk < h->split_min_blocks) return factor_loop_single(h, make_ctx(h, M, tri_base));
  SplitSchedule s;
#ifdef GPX_TIMELINE
  const auto tl_t0 = std::chrono::steady_clock::now();
#endif
  GPX_RET(s.init(h, M, tri_base));
  for (int64_t k = 0; k < nblk; ++k) {
    TL_HOST(2 * k);
    GPX_RET(s.chain(k));
    TL_HOST(2 * k + 1);
    GPX_RET(s.side(k));
  }
  TL_HOST(2 * nblk);
  return s.finish();
}
