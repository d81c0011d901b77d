// gemm.hip -- fp64 MFMA "NT" GEMM for gfx950:  C[M,N] = beta*C + alpha * A[M,K] * B[N,K]^T
//
// The one GEMM shape the factorisation needs (both operands K-contiguous, row-major):
//   panel solve      L_ik = A_ik * inv(L_kk)^T          (replaces the TRSM of cho_factor, gpax/gp.py:296)
//   trailing update  A_ij -= L_ik * L_jk^T              (SYRK/GEMM of the right-looking Cholesky)
//   posterior        S_j -= X_k * L_jk^T, cov -= V^T V  (cho_solve + matmul of gpax/gp.py:297-299)
//
// Built on v_mfma_f64_16x16x4_f64 (__builtin_amdgcn_mfma_f64_16x16x4f64):
//   A operand: lane l holds A[i = l&15][k = l>>4];  B operand: lane l holds B[k = l>>4][j = l&15];
//   C/D: 4 doubles per lane, col = l&15, row = (l>>4) + 4*reg   (the f64 map, NOT the f32 one).
// One f64 MFMA is 2048 flop and occupies the SIMD's matrix pipe for 64 cycles, so the
// kernel is MFMA-bound with very little operand traffic: a 2x2 register block per wave
// needs only 4 ds_read_b64 per 4 MFMAs (256 cycles).  Tiles are staged global -> registers
// -> LDS (one register set, written after the barrier; the next tile's global loads are in
// flight during the MFMAs).  Fragments are read 16 B at a time (ds_read_b128, 256 B/clk; a
// ds_read2_b64 runs at half that and made this kernel LDS-bound once the MFMAs issued every 64
// cycles): lane (i, g) takes k = 8s+2g, 8s+2g+1 of row i and feeds them to two consecutive MFMA
// steps -- a permutation of the k index applied to A and B alike, so the sum over k is unchanged.
// (Occupancy is not the limiter: padding LDS to leave 3 workgroups per CU instead of 4 changes nothing, 2 cost 13 %;
// at K = 128 the kernel is bound by moving its C tiles: 16 B per 256 flop.)
// LDS rows are padded to 36 doubles: row stride = 72 dwords = 8 (mod 64) makes the ds_read_b128
// lane groups ({0-3,12-15,20-27}, ...) hit 16 distinct 4-bank slots.

#include "gpx_internal.h"

struct GemmArgs {
  const double* A; const double* B; double* C;
  int64_t lda, ldb, ldc;
  int64_t M, N, K;
  double alpha, beta;
  int64_t row_off, col_off;
  int lower, b_lower_tri;
  int a_upper_tri;   // A[m][k] == 0 for k < m (rows of an upper-triangular factor): K below the tile's first row is dead
  // map_S < 0: 1-D row-major enumeration of the lower trapezoid's tiles (row_off == col_off), map_ntj tile columns
  int map_S; int64_t map_ntj;
  // batched hyper-parameter draws: blockIdx.z = draw, per-draw strides (0: shared)
  int64_t sA, sB, sC, s_flag, s_info;
  // block-cyclic row distribution (distributed factorisation): local row m is global row
  // ((m / cyc_nb) * cyc_P + cyc_r) * cyc_nb + m % cyc_nb;  cyc_P <= 1: rows are global already
  int cyc_P, cyc_r, cyc_nb;
  unsigned* signal;   // one-pass path: publish the stored tile to an already-resident consumer (see GemmLaunch::signal)
  int64_t signal_m;   // > 0: only tiles with m0 < signal_m publish
  const unsigned* wait_flag; unsigned wait_count;   // one-pass path: resident start (see GemmLaunch::wait_flag)
  const unsigned* wait_flag2; unsigned wait_count2;
  int* wait_info;     // set to -7 when the bounded wait gives up
  int tl_id;          // diagnostic timeline build only: launch number (index of this launch's end-time cell)
};

// first global row of the tile whose first (local) row is row_off + m0
__device__ __forceinline__ int64_t tile_global_row(const GemmArgs& g, int64_t m0) {
  int64_t r = g.row_off + m0;
  if (g.cyc_P > 1) r = ((r / g.cyc_nb) * g.cyc_P + g.cyc_r) * g.cyc_nb + r % g.cyc_nb;
  return r;
}

#define GEMM_BK 32
#ifndef GEMM_MAIN_WAVES
#define GEMM_MAIN_WAVES 3   // waves per SIMD the register budget of gemm_nt_kernel is sized for (168 VGPRs: three workgroups per CU)
#endif
#define GEMM_LDT (GEMM_BK + 4)

// Diagnostic build only (make TIMELINE=1 -> libgpx_hip_tl.so, tools/timeline2.py): the first thread of every
// kernel logs (kind, s_memrealtime, M, N) into a device buffer -- an unperturbed timeline of kernel STARTS.
#ifdef GPX_TIMELINE
__device__ unsigned long long* g_tl_gemm;
__device__ unsigned long long* g_tl_gemm_end;   // [launch number] = latest time any workgroup of that launch passed GEMM_TL_END
static int g_tl_launch = 0;
void tl_set_gemm(unsigned long long* p, unsigned long long* ends) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_gemm), &p, sizeof(p));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_gemm_end), &ends, sizeof(ends));
  g_tl_launch = 0;
}
#define GEMM_TL(kind, a, b)                                                                                  \
  do {                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && g_tl_gemm) {                               \
      const unsigned long long slot = atomicAdd(g_tl_gemm, 1ull);                                            \
      if (slot < 16000) {                                                                                    \
        unsigned long long* r = g_tl_gemm + 4 + 4 * slot;                                                    \
        r[0] = (unsigned long long)(kind) | ((unsigned long long)(g.tl_id + 1) << 8); r[1] = __builtin_amdgcn_s_memrealtime(); \
        r[2] = (unsigned long long)(a); r[3] = (unsigned long long)(b);                                      \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)
#define GEMM_TL_END()                                                                                        \
  do {                                                                                                       \
    if (threadIdx.x == 0 && g_tl_gemm_end && g.tl_id >= 0 && g.tl_id < 8192)                                 \
      atomicMax(&g_tl_gemm_end[g.tl_id], (unsigned long long)__builtin_amdgcn_s_memrealtime());              \
  } while (0)
#else
#define GEMM_TL(kind, a, b) do { } while (0)
#define GEMM_TL_END() do { } while (0)
#endif

// Diagnostic build only (tools/gemm_bench): s_memtime stamps of one workgroup's phases.
#ifdef GPX_GEMM_STAMPS
__device__ unsigned long long* g_gemm_stamps;
#define GEMM_STAMP(slot)                                                                         \
  do {                                                                                           \
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0) g_gemm_stamps[slot] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define GEMM_STAMP(slot) do { } while (0)
#endif

// draw b of a batched launch works on its own matrices / flags
__device__ __forceinline__ void batch_offset(GemmArgs& g) {
  const int64_t z = blockIdx.z;
  if (z == 0) return;
  g.A += z * g.sA; g.B += z * g.sB; g.C += z * g.sC;
  if (g.signal) g.signal += z * g.s_flag;
  if (g.wait_flag) g.wait_flag += z * g.s_flag;
  if (g.wait_flag2) g.wait_flag2 += z * g.s_flag;
  if (g.wait_info) g.wait_info += z * g.s_info;
}

// ROLE only names the instantiation (0: the bulk trailing update, 1: look-ahead column updates and near blocks on the panel
// stream): profilers then report the roofline kernel's launches apart from the small ones that share its code
template <int BM, int BN, bool HASC, int ROLE = 0>   // HASC: beta != 0 (compile-time, so that the C requests are straight-line code)
// (256, 2): a register budget of 256 makes hipcc keep the accumulators in VGPRs.  With the default budget it
// puts them in AGPRs, and v_mfma_f64_16x16x4_f64 with an AGPR accumulator issues every ~100 cycles instead of
// 64 (tools/peaks: 49 vs 78 TFLOP/s) -- the single largest factor in this kernel's throughput.
__global__ __launch_bounds__(256, GEMM_MAIN_WAVES) void gemm_nt_kernel(GemmArgs g) {
  constexpr int TM = BM / 32, TN = BN / 32;        // 16x16 MFMA tiles per wave (waves as 2 x 2)
  constexpr int NA = BM * (GEMM_BK / 2) / 256;      // 16-B chunks per thread, A tile
  constexpr int NBc = BN * (GEMM_BK / 2) / 256;
  __shared__ __attribute__((aligned(16))) double As[BM * GEMM_LDT];
  __shared__ __attribute__((aligned(16))) double Bs[BN * GEMM_LDT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  GEMM_TL(4, g.M, g.N * 1000000 + g.K);
  batch_offset(g);
  int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  if (g.map_S < 0) {
    // 1-D row-major enumeration of the lower trapezoid's tiles (row_off == col_off): no empty workgroups, and the
    // round-robin deal of consecutive workgroup ids over the 8 XCDs stays balanced whatever the grid width (with
    // the 2-D grid and a width that is a multiple of 8 tiles, XCD x owns the tile columns = x mod 8 and XCD 0
    // gets 25 % more of the triangle than XCD 7: tools/gemm_shape, 33 vs 42 TFLOP/s at 4096 vs 3968 rows)
    const int64_t id = blockIdx.x, ntj = g.map_ntj, ntri = ntj * (ntj + 1) / 2;
    int64_t ti, tj;
    if (id < ntri) {
      ti = (int64_t)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
      while (ti * (ti + 1) / 2 > id) --ti;
      while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
      tj = id - ti * (ti + 1) / 2;
    } else {
      ti = ntj + (id - ntri) / ntj; tj = (id - ntri) % ntj;
    }
    m0 = ti * BM; n0 = tj * BN;
  } else if (g.lower && (tile_global_row(g, m0) + BM - 1 < g.col_off + n0)) {
    return;  // tile strictly above the diagonal
  }

  int64_t kend = g.K;
  if (g.b_lower_tri) {  // B[n][k] == 0 for k > n: K beyond the tile's last row is dead
    int64_t kk = (n0 + BN + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    if (kk < kend) kend = kk;
  }

  const int fr = lane & 15, fk = lane >> 4;
  v4d acc[TM][TN], cin[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) { acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0}; cin[i][j] = (v4d){0.0, 0.0, 0.0, 0.0}; }

  v2d ra[NA], rb[NBc];
  const double* Ag = g.A + m0 * g.lda;
  const double* Bg = g.B + n0 * g.ldb;

  auto load_tiles = [&](int64_t k0) {
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int c = tid + it * 256;
      ra[it] = *reinterpret_cast<const v2d*>(Ag + (int64_t)(c >> 4) * g.lda + k0 + 2 * (c & 15));
    }
#pragma unroll
    for (int it = 0; it < NBc; ++it) {
      const int c = tid + it * 256;
      rb[it] = *reinterpret_cast<const v2d*>(Bg + (int64_t)(c >> 4) * g.ldb + k0 + 2 * (c & 15));
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int c = tid + it * 256;
      *reinterpret_cast<v2d*>(&As[(c >> 4) * GEMM_LDT + 2 * (c & 15)]) = ra[it];
    }
#pragma unroll
    for (int it = 0; it < NBc; ++it) {
      const int c = tid + it * 256;
      *reinterpret_cast<v2d*>(&Bs[(c >> 4) * GEMM_LDT + 2 * (c & 15)]) = rb[it];
    }
  };

  const double* Ap = &As[(wm * (BM / 2) + fr) * GEMM_LDT + 2 * fk];
  const double* Bp = &Bs[(wn * (BN / 2) + fr) * GEMM_LDT + 2 * fk];

  const int64_t kbeg = g.a_upper_tri ? (m0 / GEMM_BK) * GEMM_BK : 0;
  GEMM_STAMP(0);
  // The C tile is streamed (read once per launch, from HBM or the Infinity Cache: ~2 us under load) and consumed only
  // in the epilogue.  Loads return IN ORDER, so whatever is requested before an operand chunk has to arrive before
  // that chunk can be used: requested up front (behind the first chunk), the C tile is waited for at the second
  // chunk and only the first chunk's MFMAs (0.85 us) cover its latency.  It is therefore requested LAST -- behind the
  // operands of the last chunk, while the second-to-last chunk is still to be multiplied -- so that two chunks of
  // MFMAs cover it and no operand load queues behind it.  (The code is peeled so that this is straight-line: inside
  // a loop with a condition the compiler's wait-count pass falls back to vmcnt(0) before the last chunk's LDS store.)
  // C addresses = a wave-uniform base (SGPR pair per tile row: the wave index goes through readfirstlane) + ONE 32-bit
  // per-lane offset: sixteen 64-bit address pairs in VGPRs cost the kernel its last registers (it spilled two, and a
  // spill's scratch traffic waits on vmcnt(0), i.e. for every operand load in flight)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  double* const cw = g.C + (m0 + (wave_u >> 1) * (BM / 2)) * g.ldc + n0 + (wave_u & 1) * (BN / 2);
  const unsigned c_lane = (unsigned)(fk * (int)g.ldc + fr);
  auto load_c = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
#pragma unroll
        // (non-temporal C loads: 71 -> 61 us standalone on some boxes, but 1.765 -> 1.81 ms per N = 4096 step in situ --
        // the next kernels re-read these tiles from the caches -- so the default policy stays)
        for (int r = 0; r < 4; ++r) cin[i][j][r] = (cw + (int64_t)(i * 16 + 4 * r) * g.ldc + j * 16)[c_lane];
      }
  };
  auto chunk = [&](int64_t k0, const bool prefetch, const bool request_c, const bool last) __attribute__((always_inline)) {
#if defined(GPX_ABLATE) && (GPX_ABLATE & 2)
    if (k0 == kbeg)
#endif
    store_tiles();
    GEMM_STAMP(1 + 3 * (int)(k0 / GEMM_BK));
    __syncthreads();
    GEMM_STAMP(2 + 3 * (int)(k0 / GEMM_BK));
#if defined(GPX_ABLATE) && (GPX_ABLATE & 1)
#else
    if (prefetch) load_tiles(k0 + GEMM_BK);  // in flight during the MFMAs below
#endif
    if (request_c) load_c();
#pragma unroll
    for (int s = 0; s < GEMM_BK / 8; ++s) {
      v2d a[TM], b[TN];
#if defined(GPX_ABLATE) && (GPX_ABLATE & 8)
      if (s == 0 && k0 == kbeg)
#endif
      {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const v2d*>(Ap + i * 16 * GEMM_LDT + 8 * s);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const v2d*>(Bp + j * 16 * GEMM_LDT + 8 * s);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
    }
    GEMM_STAMP(3 + 3 * (int)(k0 / GEMM_BK));
#if defined(GPX_ABLATE) && (GPX_ABLATE & 4)
#else
    if (!last) __syncthreads();
#endif
  };
  load_tiles(kbeg);
  if (kbeg + GEMM_BK >= kend) {          // a single chunk
    if (HASC) load_c();
    chunk(kbeg, false, false, true);
  } else {
    int64_t k0 = kbeg;
#ifdef GEMM_EARLY_C
    if (HASC) load_c();
    for (; k0 + GEMM_BK < kend; k0 += GEMM_BK) chunk(k0, true, false, false);
    chunk(k0, false, false, true);
#else
    for (; k0 + 2 * GEMM_BK < kend; k0 += GEMM_BK) chunk(k0, true, false, false);
    chunk(k0, true, HASC, false);         // second to last
    chunk(k0 + GEMM_BK, false, false, true);
#endif
  }
  GEMM_STAMP(13);

  // epilogue: lane (col = l&15, row group = l>>4), reg r -> row = (l>>4) + 4r
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + wn * (BN / 2) + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * (BM / 2) + i * 16 + fk + 4 * r;
        // (non-temporal and agent-scope write-through stores were A/B-tested here: no effect on the step time)
        g.C[row * g.ldc + col] = HASC ? fma(g.beta, cin[i][j][r], g.alpha * acc[i][j][r]) : g.alpha * acc[i][j][r];
      }
    }
  GEMM_STAMP(14);
  GEMM_TL_END();
}

// (A streaming variant -- 768 resident workgroups taking tiles from a device counter, the next tile's first operand
// chunk requested during the last chunk of the current one so that only a workgroup's first tile starts cold -- was
// written and measured: 41.9 vs 46.1 TFLOP/s at M=4736 N=3712 K=128, 1.91 vs 1.76 ms per N = 4096 step.  Like a
// tile-pipelined persistent kernel (removed in round 2) and the 128x128 tiles, it loses to plain one-tile workgroups: the hardware dispatcher
// de-correlates the phases of the three workgroups of a CU, resident workgroups stay in lock-step.  Removed.)
// (A K = 128 specialisation with every global load of the tile in flight at once -- both K-halves and the C tile requested up
// front, the halves through a 69.6 KB LDS image one after the other, three barriers per tile instead of eight, two
// workgroups per CU -- was written and measured: 40.5 vs 46.2 TFLOP/s at M=4736 N=3712, 1.67 vs 1.62 ms per N = 4096 step.
// Fifth variant to lose to three 4-wave workgroups per CU on the chunked pipeline.  Removed.)
// (A 128x128-tile kernel -- 64x64 per wave, 128 accumulator VGPRs, accumulators initialised from C, K in chunks of 16
// through a double-buffered LDS image, two workgroups per CU -- halves the operand bytes per flop but measured slower
// than the 64x64 kernel at every shape: 38 vs 44 TFLOP/s at M=4736 N=3712 K=128, 60.5 vs 65.9 at 24576^2 K=512.  With
// eight waves per CU the per-tile C latency is exposed; the 64x64 kernel's three workgroups per CU cover it.  Removed.)

// ---- latency variant for K = nb (64 or 128): one pass, one barrier ---------------------------
// The panel solve and the look-ahead update sit on the critical path of the factorisation and
// are small (M x nb x nb), so what matters is the latency of ONE tile, not throughput: the whole
// K range of the 32-row A and B tiles is fetched with every 16-B load of the workgroup in flight
// at once (coalesced row segments), staged through LDS in two halves, and each wave then runs its
// K/4 MFMAs on two accumulators (a dependent f64 MFMA issues every ~190 cycles, an
// independent one every ~100).  One global round trip and one barrier per tile instead of K/32.
// (A register-only variant with per-lane contiguous K chunks was tried and measured slower:
// its loads touch 64 different cache lines per instruction.)
template <int K>
__device__ __forceinline__ void onepass_body(const GemmArgs& g, const int64_t m0, const int64_t n0, double* __restrict__ As,
                                             double* __restrict__ Bs) {
  // LDS holds HALF of K at a time (K/2 + 4 doubles per row): 34 KB at K = 128, so four workgroups fit a CU and even the widest
  // panel of the factorisation is resident in a single round.  All global loads of BOTH halves are in
  // flight together; the second half waits in registers while the first is in the MFMAs.
  constexpr int KH = K / 2;
  constexpr int LD = KH + 4;   // stride 8 (mod 64) dwords: conflict-free ds_read_b128 fragment reads
  constexpr int NCH = 32 * (KH / 2) / 256;   // 16-B chunks per thread, operand and half
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  if (g.wait_flag || g.wait_flag2) {   // resident start: one lane polls (bounded: 0.25 s), one agent-scope acquire, the barrier releases the loads
    GEMM_TL(6, g.M, g.N * 1000000 + (g.beta != 0.0 ? 1 : 0));   // (timeline build) became resident
    if (tid == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while ((g.wait_flag && __hip_atomic_load(g.wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.wait_count) ||
             (g.wait_flag2 && __hip_atomic_load(g.wait_flag2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.wait_count2)) {
        __builtin_amdgcn_s_sleep(4);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 25000000ull) { if (g.wait_info) atomicMin(g.wait_info, -7); break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  GEMM_TL(3, g.M, g.N * 1000000 + (g.beta != 0.0 ? 1 : 0));
  const double* Ag = g.A + m0 * g.lda;
  const double* Bg = g.B + n0 * g.ldb;
  // B lower triangular (the panel solve against W_k): B[n][k] = 0 for k > n, so the tile columns left of K/2 need
  // the first half of K only -- a quarter of the panel solve's loads and MFMAs
  const int nhalf = (g.b_lower_tri && n0 + 32 <= KH) ? 1 : 2;
  v2d ra[2][NCH], rb[2][NCH];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    if (hf < nhalf) {
#pragma unroll
      for (int it = 0; it < NCH; ++it) {
        const int c = tid + it * 256;
        ra[hf][it] = *reinterpret_cast<const v2d*>(Ag + (int64_t)(c / (KH / 2)) * g.lda + hf * KH + 2 * (c % (KH / 2)));
        rb[hf][it] = *reinterpret_cast<const v2d*>(Bg + (int64_t)(c / (KH / 2)) * g.ldb + hf * KH + 2 * (c % (KH / 2)));
      }
    }
  }
  const int fr = lane & 15, fk = lane >> 4;
  double* cp = g.C + (m0 + wm * 16 + fk) * g.ldc + n0 + wn * 16 + fr;   // row = fk + 4r, col = fr
  v4d cin = {0.0, 0.0, 0.0, 0.0};
  if (g.beta != 0.0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) cin[r] = cp[(int64_t)(4 * r) * g.ldc];
  }
  const double* Ap = &As[(wm * 16 + fr) * LD + 2 * fk];
  const double* Bp = &Bs[(wn * 16 + fr) * LD + 2 * fk];
  v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    if (hf >= nhalf) break;
    if (hf) __syncthreads();   // everyone is done reading the first half
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
      const int c = tid + it * 256;
      *reinterpret_cast<v2d*>(&As[(c / (KH / 2)) * LD + 2 * (c % (KH / 2))]) = ra[hf][it];
      *reinterpret_cast<v2d*>(&Bs[(c / (KH / 2)) * LD + 2 * (c % (KH / 2))]) = rb[hf][it];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KH / 8; ++s) {   // 16-B fragment reads: k = 8s + 2g, 8s + 2g + 1 (same permutation for A and B)
      const v2d a = *reinterpret_cast<const v2d*>(Ap + 8 * s);
      const v2d b = *reinterpret_cast<const v2d*>(Bp + 8 * s);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, acc1, 0, 0, 0);
    }
  }
  if (g.signal && (g.signal_m <= 0 || m0 < g.signal_m)) {
    // publish: write-through (sc1) stores, each wave drains its own, the barrier collects the four waves, ONE
    // lane adds to the agent-scope counter the resident consumer polls
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = g.alpha * (acc0[r] + acc1[r]);
      if (g.beta != 0.0) v += g.beta * cin[r];
      __hip_atomic_store(&cp[(int64_t)(4 * r) * g.ldc], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(g.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    GEMM_TL_END();
    return;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double v = g.alpha * (acc0[r] + acc1[r]);
    if (g.beta != 0.0) v += g.beta * cin[r];
    cp[(int64_t)(4 * r) * g.ldc] = v;
  }
  GEMM_TL_END();
}

template <int K>
__global__ __launch_bounds__(256, 2) void gemm_nt_onepass_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double As[32 * (K / 2 + 4)];
  __shared__ __attribute__((aligned(16))) double Bs[32 * (K / 2 + 4)];
  const int64_t m0 = (int64_t)blockIdx.y * 32, n0 = (int64_t)blockIdx.x * 32;
  if (g.lower && (tile_global_row(g, m0) + 31 < g.col_off + n0)) return;
  batch_offset(g);
  onepass_body<K>(g, m0, n0, As, Bs);
}

// The two band kernels of the chain in ONE launch (split-panel schedule with in-kernel hand-overs): every workgroup
// first forms its 32x32 tile of S = A_{k+1,k} W_k^T (gs: sleeps on potrf_diag(k)'s flag, publishes the tile and counts
// into gs.signal), then the workgroups of the lower tiles wait for all (nb/32)^2 tiles of S (gu.wait_flag = gs.signal)
// and apply A_{k+1,k+1} -= S S^T to their tile (gu: publishes to the counter potrf_diag(k+1) polls).  The second kernel's
// launch -- ~2 us between the two even back to back on one stream -- becomes a flag round trip inside a launch whose
// 16 workgroups are all resident (they never wait for a workgroup that is not: phase 1 waits only for potrf_diag).
// Phase 0 (gz.K > 0, lower tiles only): one more update of the SAME diagonal block, whose operands are ready before W_k
// is -- the previous panel's contribution -- applied while the kernel would otherwise sleep.  The thread that stores an
// entry in phase 0 is the thread that loads it in phase 2 (same tile, same lane map): program order, no fence.
template <int K>
__global__ __launch_bounds__(256, 2) void chain_band_kernel(GemmArgs gz, GemmArgs gs, GemmArgs gu) {
  __shared__ __attribute__((aligned(16))) double As[32 * (K / 2 + 4)];
  __shared__ __attribute__((aligned(16))) double Bs[32 * (K / 2 + 4)];
  const int64_t m0 = (int64_t)blockIdx.y * 32, n0 = (int64_t)blockIdx.x * 32;
  batch_offset(gz); batch_offset(gs); batch_offset(gu);
  if (gz.K > 0 && n0 <= m0) {
    onepass_body<K>(gz, m0, n0, As, Bs);
    __syncthreads();   // the LDS image is reused by phase 1
  }
  onepass_body<K>(gs, m0, n0, As, Bs);
  if (n0 > m0) return;   // strictly upper tile of the diagonal block
  onepass_body<K>(gu, m0, n0, As, Bs);
}

static void fill_args(GemmArgs& g, const GemmLaunch& L) {
  g.A = L.A; g.B = L.B; g.C = L.C; g.lda = L.lda; g.ldb = L.ldb; g.ldc = L.ldc;
  g.M = L.M; g.N = L.N; g.K = L.K; g.alpha = L.alpha; g.beta = L.beta;
  g.row_off = L.row_off; g.col_off = L.col_off; g.lower = L.lower; g.b_lower_tri = L.b_lower_tri;
  g.cyc_P = L.cyc_P; g.cyc_r = L.cyc_r; g.cyc_nb = L.cyc_nb;
  g.a_upper_tri = L.a_upper_tri;
  g.map_S = 0; g.map_ntj = 0;
  g.sA = L.sA; g.sB = L.sB; g.sC = L.sC; g.s_flag = L.s_flag; g.s_info = L.s_info;
  g.signal = L.signal; g.signal_m = L.signal_m; g.wait_flag = L.wait_flag; g.wait_count = L.wait_count; g.wait_info = L.wait_info;
  g.wait_flag2 = L.wait_flag2; g.wait_count2 = L.wait_count2;
  g.tl_id = -1;
}

// S (nb x nb, L.N == L.M == L.K == nb) and the update of the diagonal block in one launch; both launches as the caller
// would have issued them separately, S.signal / U.wait_flag pointing at the same zeroed counter
int launch_chain_band(gpx_handle* h, hipStream_t s, const GemmLaunch* Z, const GemmLaunch& S, const GemmLaunch& U) {
  const int64_t nb = S.K;
  if ((nb != 128 && nb != 64) || S.M != nb || S.N != nb || U.M != nb || U.N != nb || U.K != nb || !S.signal || U.wait_flag != S.signal ||
      U.wait_count != (unsigned)((nb / 32) * (nb / 32))) {
    gpx_set_error(h, "chain_band: contract violated");
    return GPX_ERR_ARG;
  }
  GemmArgs gz, gs, gu;
  fill_args(gs, S); fill_args(gu, U);
  if (Z) {
    if (Z->M != nb || Z->N != nb || Z->K != nb || Z->C != U.C || Z->signal || Z->wait_flag) { gpx_set_error(h, "chain_band: phase-0 contract violated"); return GPX_ERR_ARG; }
    fill_args(gz, *Z);
  } else {
    fill_args(gz, U); gz.K = 0; gz.signal = nullptr; gz.wait_flag = nullptr;
  }
#ifdef GPX_TIMELINE
  gs.tl_id = g_tl_launch++; gu.tl_id = g_tl_launch++;
#endif
  if (S.batch != U.batch || (Z && Z->batch != S.batch)) { gpx_set_error(h, "chain_band: batch mismatch"); return GPX_ERR_ARG; }
  dim3 grid((unsigned)(nb / 32), (unsigned)(nb / 32), (unsigned)S.batch);
  if (nb == 128) hipExtLaunchKernelGGL((chain_band_kernel<128>), grid, dim3(256), 0, s, nullptr, U.stop_event, 0, gz, gs, gu);
  else hipExtLaunchKernelGGL((chain_band_kernel<64>), grid, dim3(256), 0, s, nullptr, U.stop_event, 0, gz, gs, gu);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

int launch_gemm_nt(gpx_handle* h, hipStream_t s, const GemmLaunch& L) {
  if (L.M <= 0 || L.N <= 0) return GPX_OK;
  const int bm = L.small_tile ? 32 : 64;
  if (L.M % bm || L.N % bm || L.K % GEMM_BK || L.K <= 0 || (L.lda & 1) || (L.ldb & 1) ||
      (reinterpret_cast<uintptr_t>(L.A) & 15) || (reinterpret_cast<uintptr_t>(L.B) & 15)) {
    gpx_set_error(h, "gemm_nt: shape/alignment contract violated (M,N % tile, K % 32, even ld, 16-B base)");
    return GPX_ERR_ARG;
  }
  GemmArgs g;
  fill_args(g, L);
#ifdef GPX_TIMELINE
  g.tl_id = g_tl_launch++;
#else
  g.tl_id = -1;
#endif
  if ((L.signal || L.wait_flag) && !(L.small_tile && (L.K == 128 || L.K == 64))) {
    gpx_set_error(h, "gemm_nt: signal is served by the one-pass kernel only");
    return GPX_ERR_ARG;
  }
  if (L.a_upper_tri && L.small_tile) {
    gpx_set_error(h, "gemm_nt: a_upper_tri is served by the 64x64 kernel only");
    return GPX_ERR_ARG;
  }
  if (L.batch < 1 || L.batch > 65535) { gpx_set_error(h, "gemm_nt: batch out of range"); return GPX_ERR_ARG; }
  dim3 grid((unsigned)(L.N / bm), (unsigned)(L.M / bm), (unsigned)L.batch);
  // lower-trapezoid updates on 64x64 tiles: 1-D row-major enumeration of the tiles (no empty workgroups, and the
  // round-robin deal of consecutive workgroup ids over the 8 XCDs stays balanced whatever the grid width)
  if (!L.small_tile && L.lower && L.row_off == L.col_off && L.cyc_P <= 1 && L.M >= L.N) {
    const int64_t nti = L.M / 64, ntj = L.N / 64;
    g.map_S = -1; g.map_ntj = ntj;
    const int64_t ntiles = ntj * (ntj + 1) / 2 + (nti - ntj) * ntj;
    const dim3 tg((unsigned)ntiles, 1u, (unsigned)L.batch);
    if (L.beta != 0.0 && L.role == 1) hipExtLaunchKernelGGL((gemm_nt_kernel<64, 64, true, 1>), tg, dim3(256), 0, s, L.start_event, L.stop_event, 0, g);
    else if (L.beta != 0.0) hipExtLaunchKernelGGL((gemm_nt_kernel<64, 64, true>), tg, dim3(256), 0, s, L.start_event, L.stop_event, 0, g);
    else hipExtLaunchKernelGGL((gemm_nt_kernel<64, 64, false>), tg, dim3(256), 0, s, L.start_event, L.stop_event, 0, g);
    GPX_HIP_CHECK(h, hipGetLastError());
    return GPX_OK;
  }
  hipEvent_t se = L.stop_event, sb = L.start_event;
  if (L.small_tile && L.K == 128) hipExtLaunchKernelGGL((gemm_nt_onepass_kernel<128>), grid, dim3(256), 0, s, sb, se, 0, g);
  else if (L.small_tile && L.K == 64) hipExtLaunchKernelGGL((gemm_nt_onepass_kernel<64>), grid, dim3(256), 0, s, sb, se, 0, g);
  else if (L.small_tile && L.beta != 0.0) hipExtLaunchKernelGGL((gemm_nt_kernel<32, 32, true>), grid, dim3(256), 0, s, sb, se, 0, g);
  else if (L.small_tile) hipExtLaunchKernelGGL((gemm_nt_kernel<32, 32, false>), grid, dim3(256), 0, s, sb, se, 0, g);
  else if (L.beta != 0.0 && L.role == 1) hipExtLaunchKernelGGL((gemm_nt_kernel<64, 64, true, 1>), grid, dim3(256), 0, s, sb, se, 0, g);
  else if (L.beta != 0.0) hipExtLaunchKernelGGL((gemm_nt_kernel<64, 64, true>), grid, dim3(256), 0, s, sb, se, 0, g);
  else hipExtLaunchKernelGGL((gemm_nt_kernel<64, 64, false>), grid, dim3(256), 0, s, sb, se, 0, g);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}
