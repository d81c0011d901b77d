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
  // first_col > 0 (bulk instantiation, 1-D map): the tiles of the first `first_col` tile columns are enumerated FIRST and
  // publish themselves (write-through stores + one count into `signal`): the consumers of a bulk update's first block column
  // -- the panel stream's next column update, the chain's band kernel -- need not wait for the whole launch to end
  int first_col; int64_t map_nti;
  // batched hyper-parameter draws: blockIdx.z = draw, per-draw strides (0: shared)
  int64_t sA, sB, sC, s_flag, s_info;
  // block-cyclic row distribution (distributed factorisation): local row m is global row
  // ((m / cyc_nb) * cyc_P + cyc_r) * cyc_nb + m % cyc_nb;  cyc_P <= 1: rows are global already
  int cyc_P, cyc_r, cyc_nb;
  unsigned* signal;   // one-pass path: publish the stored tile to an already-resident consumer (see GemmLaunch::signal)
  int64_t signal_m;   // > 0: only tiles with m0 < signal_m publish
  const unsigned* wait_flag; unsigned wait_count;   // one-pass path: resident start (see GemmLaunch::wait_flag)
  const unsigned* wait_flag2; unsigned wait_count2;
  int64_t wait_m;     // > 0: only tiles with m0 < wait_m wait
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
// Wave priorities (s_setprio): the chain's band kernel and the panel stream's kernels share their SIMDs with bulk-update
// waves, which are throughput work -- their waves issue ahead of those.  Same box, alternated: N = 2048 0.574-0.578 vs
// 0.578-0.582 ms per step (one run 0.676), N = 4096 1.455-1.477 vs 1.471-1.486, N = 8192 a tie; (2, 3) instead of (1, 2) no
// better (profiles/r03_wave_priority_ab.txt).
#ifndef GEMM_Q_PRIO
#define GEMM_Q_PRIO 1      // the panel stream's kernels: panel solve, look-ahead column updates
#endif
#ifndef GEMM_BAND_PRIO
#define GEMM_BAND_PRIO 2   // the chain's band kernel
#endif
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
// [8192 + launch number] = latest time a workgroup of that launch raised its `signal` counter
#define GEMM_TL_PUB()                                                                                        \
  do {                                                                                                       \
    if (threadIdx.x == 0 && g_tl_gemm_end && g.tl_id >= 0 && g.tl_id < 8192)                                 \
      atomicMax(&g_tl_gemm_end[8192 + g.tl_id], (unsigned long long)__builtin_amdgcn_s_memrealtime());       \
  } while (0)
#else
#define GEMM_TL(kind, a, b) do { } while (0)
#define GEMM_TL_END() do { } while (0)
#define GEMM_TL_PUB() do { } while (0)
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
  GEMM_TL(ROLE == 1 ? 8 : 4, g.M, g.N * 1000000 + g.K);
  if constexpr (ROLE == 1) __builtin_amdgcn_s_setprio(GEMM_Q_PRIO);
  batch_offset(g);
  int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  bool first_part = false;   // bulk instantiation: this tile lies in the first block column and publishes itself
  if (g.map_S < 0) {
    // 1-D row-major enumeration of the lower trapezoid's tiles (row_off == col_off): no empty workgroups, and the
    // round-robin deal of consecutive workgroup ids over the 8 XCDs stays balanced whatever the grid width (with
    // the 2-D grid and a width that is a multiple of 8 tiles, XCD x owns the tile columns = x mod 8 and XCD 0
    // gets 25 % more of the triangle than XCD 7: tools/gemm_shape, 33 vs 42 TFLOP/s at 4096 vs 3968 rows)
    int64_t id = blockIdx.x, ntj = g.map_ntj, shift = 0;
    int64_t ti = 0, tj = 0;
    bool mapped = false;
    if (ROLE == 0 && g.first_col > 0) {
      const int64_t fc = g.first_col, tri1 = fc * (fc + 1) / 2, n1 = fc * g.map_nti - fc * (fc - 1) / 2;
      if (id < n1) {   // row-major over the first fc tile columns: row r holds min(r + 1, fc) tiles
        if (id < tri1) {
          ti = (int64_t)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
          while (ti * (ti + 1) / 2 > id) --ti;
          while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
          tj = id - ti * (ti + 1) / 2;
        } else {
          ti = fc + (id - tri1) / fc; tj = (id - tri1) % fc;
        }
        mapped = true; first_part = true;
      } else {   // the rest: the same trapezoid without its first fc tile columns (and rows)
        id -= n1; ntj -= fc; shift = fc;
      }
    }
    if (!mapped) {
      const int64_t ntri = ntj * (ntj + 1) / 2;
      if (id < ntri) {
        ti = (int64_t)((sqrt(8.0 * (double)id + 1.0) - 1.0) * 0.5);
        while (ti * (ti + 1) / 2 > id) --ti;
        while ((ti + 1) * (ti + 2) / 2 <= id) ++ti;
        tj = id - ti * (ti + 1) / 2;
      } else {
        ti = ntj + (id - ntri) / ntj; tj = (id - ntri) % ntj;
      }
      ti += shift; tj += shift;
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
#ifdef GEMM_LATE_C
    // experiment: the C tile requested in the LAST chunk, when no prefetch register is live (128 VGPRs: a SIMD then holds three of
    // these waves AND a wave of the one-pass / band kernels, which the 160-VGPR form cannot)
    for (; k0 + GEMM_BK < kend; k0 += GEMM_BK) chunk(k0, true, false, false);
    chunk(k0, false, HASC, true);
#else
    for (; k0 + 2 * GEMM_BK < kend; k0 += GEMM_BK) chunk(k0, true, false, false);
    chunk(k0, true, HASC, false);         // second to last
    chunk(k0 + GEMM_BK, false, false, true);
#endif
#endif
  }
  GEMM_STAMP(13);

  // The look-ahead instantiation (ROLE 1) can hand its first block row to an already-resident consumer -- the band kernel of
  // the next step, asleep on `signal` -- the way the one-pass kernel does: write-through stores, each wave drains its own,
  // the barrier collects the four waves, ONE lane adds to the agent-scope counter.  Same arithmetic, same bits; the bulk
  // instantiation (ROLE 0) carries none of this.
  if constexpr (BM == 64 && HASC) {
    if (g.signal && (ROLE == 1 ? m0 < g.signal_m : first_part)) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int64_t col = n0 + wn * (BN / 2) + j * 16 + fr;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t row = m0 + wm * (BM / 2) + i * 16 + fk + 4 * r;
            const double v = HASC ? fma(g.beta, cin[i][j][r], g.alpha * acc[i][j][r]) : g.alpha * acc[i][j][r];
            __hip_atomic_store(&g.C[row * g.ldc + col], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(g.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      GEMM_TL_PUB();
      GEMM_TL_END();
      return;
    }
  }
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
#ifdef GEMM_WT_C
        // experiment: write-through C stores, so that the L2s never hold megabytes of dirty C tiles that every kernel boundary
        // and release of the OTHER streams has to write back first
        __hip_atomic_store(&g.C[row * g.ldc + col], HASC ? fma(g.beta, cin[i][j][r], g.alpha * acc[i][j][r]) : g.alpha * acc[i][j][r],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        g.C[row * g.ldc + col] = HASC ? fma(g.beta, cin[i][j][r], g.alpha * acc[i][j][r]) : g.alpha * acc[i][j][r];
#endif
      }
    }
  GEMM_STAMP(14);
  GEMM_TL_END();
}

// (The STRIDED lazy update -- the bulk launch of step k visits only the block columns k+3, k+3+G, k+3+2G, ... and applies
// the panels max(0, k-G+1) .. k to them in one K = G nb pass, so that every hand-over and the per-step distribution of the
// work are those of the ungrouped schedule while the tiles are K = G nb tiles and each C tile is read and written once per
// G panels; with the first column visited one step earlier too, so that its published tiles stay K = nb tiles -- was written
// in round 3, passed the parity tests and measured SLOWER: N = 4096 G = 4: 1.54-1.58 vs 1.49-1.52 ms per step, the bulk
// kernel's summed time 1.05 vs 0.89 ms per step (30.9 vs 35.6 TFLOP/s in situ); G = 2: 1.48-1.49 (a tie); G = 8: 1.88;
// N = 8192 G = 4: 6.03-6.15 vs 5.91-5.95 with the quad grouping (42.5 vs 45.5 TFLOP/s), profiles/r03_bulk_strided_lazy.txt.
// A launch with a quarter of the tiles, each living four times as long, leaves the chip's 768 slots partly empty at N = 4096
// (240-500 tiles per launch) and runs 3 instead of 2.5 rounds at N = 8192; the K = 512 rate of 64-67 TFLOP/s needs launches
// of many rounds.  Removed.)
// (The same idea for the border rows alone -- the launch leaves the test points' rows, 42-57 % of its tiles at N = 4096 / M = 1000,
// out of the trapezoid beyond its first block column and updates them in a rectangular segment over every G-th block column
// with the G panels those columns lack, inside the SAME launch, so that no launch gets shorter of tiles -- was measured too:
// G = 4 at N = 4096: the bulk kernel's summed time 1.01 vs 0.89 ms per step (31.3 vs 35.3 TFLOP/s), the step 1.50-1.60 vs
// 1.47-1.60; G = 2 a tie; N = 8192 41-42 vs 42-43 TFLOP/s (profiles/r03_bulk_lazy_border.txt).  In these launches a K = 512 tile
// is simply a tile that holds its slot four times as long; the per-chunk stalls of a lone wave (5.4k cycles per 32 columns of
// K against 2k of MFMA work) are what three resident workgroups per CU hide from one another, whatever K is.  Removed.)
// (TWO operand chunks in flight -- a second register set taking every other chunk, the first two chunks requested together at
// the start of the tile, the C tile at the second-to-last chunk, so that every load but the first has two chunks of MFMAs to
// arrive in -- was measured stand-alone in round 3: at two workgroups per CU (232 VGPRs unconstrained) 41.9 vs 45.7 TFLOP/s at
// M=4736 N=3712 K=128, 54.0 vs 58.3 at K=512, a tie on a 117-tile launch where every wave is alone on its SIMD; at three per CU
// it spills 60 registers (32.7 TFLOP/s): profiles/r03_gemm_deep_prefetch.txt.  The lone wave's 1.7k cycles per chunk beside its
// 2k of MFMA work are the LDS round trips and the two barriers of a chunk, not the operands' arrival.  Its by-product --
// operand addresses as a wave-uniform base + one 32-bit lane offset, 160 -> 132 VGPRs at the same stand-alone speed -- lost in
// situ like the 128-VGPR form of round 2 (N = 4096: 1.48-1.54 vs 1.478-1.50 ms, N = 8192 +0.5 %,
// profiles/r03_gemm_uniform_addressing_ab.txt: a bulk wave that leaves room on its SIMD gets a neighbour).  Both removed.)
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
// One 32 x 32 tile, the pieces: a bounded resident wait, the tile product, the publication of its result.
struct TileOp {
  const double* A; const double* B; double* C;
  int64_t lda, ldb, ldc;
  double alpha, beta;
  int b_lower_tri;
};

// resident start: one lane polls (bounded: 0.25 s), one agent-scope acquire, the barrier releases the loads
// (A pipelined poll -- four flag loads in flight ~64 cycles apart, each tested as it returns, instead of load / compare / sleep /
// clock read with one load in flight -- was measured in round 3 at all three global-flag waits of the chain: 0.307-0.312 vs
// 0.307-0.312 ms at N = 1024, 0.584-0.607 vs 0.580-0.582 at N = 2048, 1.507-1.545 vs 1.498-1.532 at N = 4096
// (profiles/r03_pipelined_poll_ab.txt): the time a hand-over takes is the producer's write-through stores and the consumer's
// operand loads, not the detection of the flag.  Not kept.)
__device__ __forceinline__ void onepass_wait(const unsigned* f1, unsigned c1, const unsigned* f2, unsigned c2, int* info) {
  if (!f1 && !f2) return;
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((f1 && __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c1) ||
           (f2 && __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < c2)) {
      __builtin_amdgcn_s_sleep(4);
      if (__builtin_amdgcn_s_memrealtime() - t0 > 25000000ull) { if (info) atomicMin(info, -7); break; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// after a published tile (write-through stores): each wave drains its own, the barrier collects the four waves, ONE lane
// adds to the agent-scope counter the resident consumer polls
__device__ __forceinline__ void onepass_signal(unsigned* counter) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// BIG: the summation order and the final rounding of gemm_nt_kernel<64,64,true> (ONE accumulator, fma(beta, C, alpha acc)),
// so that a tile of a look-ahead column update gets the same bits whichever kernel computes it (the chain's band kernel
// takes over one block of it, schedule.hip); otherwise two accumulators (a dependent f64 MFMA issues every ~190 cycles,
// an independent one every ~100).  PUBLISH: write-through (sc1) stores for a consumer that is already resident.
template <int K, bool BIG>
__device__ __forceinline__ void onepass_tile(const TileOp& g, const int64_t m0, const int64_t n0, double* __restrict__ As,
                                             double* __restrict__ Bs, const bool publish) {
  // LDS holds HALF of K at a time (K/2 + 4 doubles per row): 34 KB at K = 128, so four workgroups fit a CU and even the widest
  // panel of the factorisation is resident in a single round.  All global loads of BOTH halves are in
  // flight together; the second half waits in registers while the first is in the MFMAs.
  constexpr int KH = K / 2;
  constexpr int LD = KH + 4;   // stride 8 (mod 64) dwords: conflict-free ds_read_b128 fragment reads
  constexpr int NCH = 32 * (KH / 2) / 256;   // 16-B chunks per thread, operand and half
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
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
      if constexpr (BIG) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, acc0, 0, 0, 0);
      else acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double v;
    if constexpr (BIG) {
      v = (g.beta != 0.0) ? fma(g.beta, cin[r], g.alpha * acc0[r]) : g.alpha * acc0[r];
    } else {
      v = g.alpha * (acc0[r] + acc1[r]);
      if (g.beta != 0.0) v += g.beta * cin[r];
    }
    if (publish) __hip_atomic_store(&cp[(int64_t)(4 * r) * g.ldc], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else cp[(int64_t)(4 * r) * g.ldc] = v;
  }
}

template <int K>
__device__ __forceinline__ void onepass_body(const GemmArgs& g, const int64_t m0, const int64_t n0, double* __restrict__ As,
                                             double* __restrict__ Bs) {
  if ((g.wait_flag || g.wait_flag2) && (g.wait_m <= 0 || m0 < g.wait_m)) {
    GEMM_TL(6, g.M, g.N * 1000000 + (g.beta != 0.0 ? 1 : 0));   // (timeline build) became resident
    onepass_wait(g.wait_flag, g.wait_count, g.wait_flag2, g.wait_count2, g.wait_info);
  }
  GEMM_TL(3, g.M, g.N * 1000000 + (g.beta != 0.0 ? 1 : 0));
  TileOp t;
  t.A = g.A; t.B = g.B; t.C = g.C; t.lda = g.lda; t.ldb = g.ldb; t.ldc = g.ldc; t.alpha = g.alpha; t.beta = g.beta;
  t.b_lower_tri = g.b_lower_tri;
  const bool publish = g.signal && (g.signal_m <= 0 || m0 < g.signal_m);
  onepass_tile<K, false>(t, m0, n0, As, Bs, publish);
  if (publish) {
    onepass_signal(g.signal);
    GEMM_TL_PUB();
  }
  GEMM_TL_END();
}

template <int K>
__global__ __launch_bounds__(256, 2) void gemm_nt_onepass_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) double As[32 * (K / 2 + 4)];
  __shared__ __attribute__((aligned(16))) double Bs[32 * (K / 2 + 4)];
  const int64_t m0 = (int64_t)blockIdx.y * 32, n0 = (int64_t)blockIdx.x * 32;
  if (g.lower && (tile_global_row(g, m0) + 31 < g.col_off + n0)) return;
  __builtin_amdgcn_s_setprio(GEMM_Q_PRIO);
  batch_offset(g);
  onepass_body<K>(g, m0, n0, As, Bs);
}

// ---- the chain's band kernel (split-panel schedule with in-kernel hand-overs) ---------------------------------------------
// Between potrf_diag(k) and potrf_diag(k+1) the chain owns a window of TWO block rows below the diagonal block (k,k):
//   row 1 (nb/32 x nb/32 workgroups, the critical ones):
//     phase 0   A_{k+1,k+1} -= P P^T, P = L_{k+1,k-1}: the previous panel's contribution, applied while the kernel would
//               otherwise sleep waiting for potrf_diag(k)  (lower tiles; the thread that stores an entry here is the thread
//               that loads it in phase U: program order, no fence)
//     phase S   S1 = A_{k+1,k} W_k^T       (sleeps on potrf_diag(k)'s flag; publishes its tile, counts into flagS)
//     phase U   A_{k+1,k+1} -= S1 S1^T     (lower tiles; waits for all of S1; publishes to the counter potrf_diag(k+1) polls)
//   row 2 (as many workgroups again, off the critical path: their deadline is the END of potrf_diag(k+1)):
//     S2 = A_{k+2,k} W_k^T  (waits for W_k and for the panel stream's column updates of step k-1, flagF; into scratch: it is
//               the next step's phase-0 operand P),  then  A_{k+2,k+1} -= S2 S1^T  -- the block that is the NEXT step's
//               phase-S operand (published, counted into flagD: the panel stream's panel of step k+1 reads it too).
// Round 2 had the panel stream update A_{k+2,k+1} and hand it to the band kernel of step k+1 through flagF: measured
// (profiles/r03_critical_path_before.txt), that publication arrived within 0-6 us of potrf_diag(k+1)'s end in the chain-bound
// middle of the factorisation and phase 0 + phase S then ran one after the other BEHIND it (gaps of 25-32 us between two
// potrf_diag against 11 where it arrived early).  With row 2 the chain depends on the panel stream only two potrf_diag
// later.  Same tile routines, same summation orders as before (row 2's update in the order of the kernel that used to do it:
// `big`), so every output keeps its bits.
struct BandArgs {
  const double* W;                       // W_k (K x K, ld K)
  const double* A1; const double* A2;    // A_{k+1,k}, A_{k+2,k} (ld)
  double* D11; double* D21;              // A_{k+1,k+1}, A_{k+2,k+1} (ld)
  const double* P0; int64_t ldp, sP0;    // phase-0 operand (nullptr: no phase 0), its leading dimension and per-draw stride
  double* S1; double* S2;                // scratch K x K (ld K)
  int64_t ld;
  const unsigned* flagW; const unsigned* flagF; unsigned countF;
  unsigned* flagS; unsigned* flagS2; unsigned* flagU; unsigned* flagD;
  const unsigned* flagB; unsigned countB;   // the bulk launch that last wrote A_{k+1,k+1}: its first block column published (or nullptr)
  int* info;
  int row2, big;
  int64_t sA, sW, s_mini, s_flag, s_info;
  int tl_id;
};

#ifdef GPX_TIMELINE
#define BAND_TL(kind, beta)                                                                                  \
  do {                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && g_tl_gemm) {            \
      const unsigned long long slot = atomicAdd(g_tl_gemm, 1ull);                                            \
      if (slot < 16000) {                                                                                    \
        unsigned long long* r = g_tl_gemm + 4 + 4 * slot;                                                    \
        r[0] = (unsigned long long)(kind) | ((unsigned long long)(b.tl_id + 1) << 8); r[1] = __builtin_amdgcn_s_memrealtime(); \
        r[2] = (unsigned long long)K; r[3] = (unsigned long long)K * 1000000ull + (beta);                    \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)
#else
#define BAND_TL(kind, beta) do { } while (0)
#endif

template <int K>
__global__ __launch_bounds__(256, 2) void chain_band_kernel(BandArgs b) {
  __shared__ __attribute__((aligned(16))) double As[32 * (K / 2 + 4)];
  __shared__ __attribute__((aligned(16))) double Bs[32 * (K / 2 + 4)];
  constexpr int NT = K / 32;
  __builtin_amdgcn_s_setprio(GEMM_BAND_PRIO);
  const int row = (int)blockIdx.y / NT;
  const int64_t m0 = (int64_t)((int)blockIdx.y % NT) * 32, n0 = (int64_t)blockIdx.x * 32;
  const int64_t z = blockIdx.z;
  if (z) {
    b.W += z * b.sW; b.A1 += z * b.sA; b.A2 += z * b.sA; b.D11 += z * b.sA; b.D21 += z * b.sA;
    if (b.P0) b.P0 += z * b.sP0;
    b.S1 += z * b.s_mini; b.S2 += z * b.s_mini;
    b.flagW += z * b.s_flag; if (b.flagF) b.flagF += z * b.s_flag; if (b.flagB) b.flagB += z * b.s_flag;
    b.flagS += z * b.s_flag; b.flagS2 += z * b.s_flag; b.flagU += z * b.s_flag; b.flagD += z * b.s_flag;
    b.info += z * b.s_info;
  }
  TileOp t;
  if (row == 0) {
    if (b.P0 && n0 <= m0) {
      onepass_wait(b.flagB, b.countB, nullptr, 0u, b.info);
      BAND_TL(3, 1);
      t.A = b.P0; t.B = b.P0; t.C = b.D11; t.lda = b.ldp; t.ldb = b.ldp; t.ldc = b.ld; t.alpha = -1.0; t.beta = 1.0; t.b_lower_tri = 0;
      onepass_tile<K, false>(t, m0, n0, As, Bs, false);
      __syncthreads();   // the LDS image is reused by phase S
    }
    BAND_TL(6, 0);
    onepass_wait(b.flagW, 1u, nullptr, 0u, b.info);
    BAND_TL(3, 0);
    t.A = b.A1; t.B = b.W; t.C = b.S1; t.lda = b.ld; t.ldb = K; t.ldc = K; t.alpha = 1.0; t.beta = 0.0; t.b_lower_tri = 1;
    onepass_tile<K, false>(t, m0, n0, As, Bs, true);
    onepass_signal(b.flagS);
    if (n0 > m0) return;   // strictly upper tile of the diagonal block
    BAND_TL(6, 1);
    onepass_wait(b.flagS, (unsigned)(NT * NT), nullptr, 0u, b.info);
    BAND_TL(3, 1);
    t.A = b.S1; t.B = b.S1; t.C = b.D11; t.lda = K; t.ldb = K; t.ldc = b.ld; t.alpha = -1.0; t.beta = 1.0; t.b_lower_tri = 0;
    onepass_tile<K, false>(t, m0, n0, As, Bs, true);
    onepass_signal(b.flagU);
    return;
  }
  // row 2
  onepass_wait(b.flagW, 1u, b.flagF, b.countF, b.info);
  t.A = b.A2; t.B = b.W; t.C = b.S2; t.lda = b.ld; t.ldb = K; t.ldc = K; t.alpha = 1.0; t.beta = 0.0; t.b_lower_tri = 1;
  onepass_tile<K, false>(t, m0, n0, As, Bs, true);
  onepass_signal(b.flagS2);
  onepass_wait(b.flagS, (unsigned)(NT * NT), b.flagS2, (unsigned)(NT * NT), b.info);
  t.A = b.S2; t.B = b.S1; t.C = b.D21; t.lda = K; t.ldb = K; t.ldc = b.ld; t.alpha = -1.0; t.beta = 1.0; t.b_lower_tri = 0;
  // published: the next reader outside this stream is the panel stream's panel of step k+1, whose first block row polls flagD
  if (b.big) onepass_tile<K, true>(t, m0, n0, As, Bs, true);
  else onepass_tile<K, false>(t, m0, n0, As, Bs, true);
  onepass_signal(b.flagD);
}

static void fill_args(GemmArgs& g, const GemmLaunch& L) {
  g.A = L.A; g.B = L.B; g.C = L.C; g.lda = L.lda; g.ldb = L.ldb; g.ldc = L.ldc;
  g.M = L.M; g.N = L.N; g.K = L.K; g.alpha = L.alpha; g.beta = L.beta;
  g.row_off = L.row_off; g.col_off = L.col_off; g.lower = L.lower; g.b_lower_tri = L.b_lower_tri;
  g.cyc_P = L.cyc_P; g.cyc_r = L.cyc_r; g.cyc_nb = L.cyc_nb;
  g.a_upper_tri = L.a_upper_tri;
  g.map_S = 0; g.map_ntj = 0; g.first_col = 0; g.map_nti = 0;
  g.sA = L.sA; g.sB = L.sB; g.sC = L.sC; g.s_flag = L.s_flag; g.s_info = L.s_info;
  g.signal = L.signal; g.signal_m = L.signal_m; g.wait_flag = L.wait_flag; g.wait_count = L.wait_count; g.wait_info = L.wait_info;
  g.wait_flag2 = L.wait_flag2; g.wait_count2 = L.wait_count2; g.wait_m = L.wait_m;
  g.tl_id = -1;
}

int launch_chain_band(gpx_handle* h, hipStream_t s, const BandLaunch& L) {
  const int64_t nb = L.nb;
  if ((nb != 128 && nb != 64) || !L.W || !L.A1 || !L.D11 || !L.S1 || !L.S2 || !L.flagW || !L.flagS || !L.flagS2 || !L.flagU || !L.flagD || !L.info ||
      (L.row2 && (!L.A2 || !L.D21)) || (L.flagF && !L.row2) || L.batch < 1) {
    gpx_set_error(h, "chain_band: contract violated");
    return GPX_ERR_ARG;
  }
  BandArgs b;
  b.W = L.W; b.A1 = L.A1; b.A2 = L.row2 ? L.A2 : L.A1; b.D11 = L.D11; b.D21 = L.row2 ? L.D21 : L.D11;
  b.P0 = L.P0; b.ldp = L.ldp; b.sP0 = L.sP0; b.S1 = L.S1; b.S2 = L.S2; b.ld = L.ld;
  b.flagW = L.flagW; b.flagF = L.flagF; b.countF = L.countF; b.flagS = L.flagS; b.flagS2 = L.flagS2; b.flagU = L.flagU; b.flagD = L.flagD; b.flagB = L.flagB; b.countB = L.countB;
  b.info = L.info; b.row2 = L.row2; b.big = L.big;
  b.sA = L.sA; b.sW = L.sW; b.s_mini = L.s_mini; b.s_flag = L.s_flag; b.s_info = L.s_info;
#ifdef GPX_TIMELINE
  b.tl_id = g_tl_launch++;
#else
  b.tl_id = -1;
#endif
  dim3 grid((unsigned)(nb / 32), (unsigned)((L.row2 ? 2 : 1) * (nb / 32)), (unsigned)L.batch);
  if (nb == 128) hipExtLaunchKernelGGL((chain_band_kernel<128>), grid, dim3(256), 0, s, nullptr, L.stop_event, 0, b);
  else hipExtLaunchKernelGGL((chain_band_kernel<64>), grid, dim3(256), 0, s, nullptr, L.stop_event, 0, b);
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
  // hand-overs inside a launch: the one-pass kernel (publish and wait), or the 64x64 look-ahead instantiation (publish its
  // first `signal_m` rows only)
  const bool big_signal = !L.small_tile && L.beta != 0.0 && L.signal && !L.wait_flag &&
                          ((L.role == 1 && L.signal_m > 0 && L.signal_m % 64 == 0) ||
                           (L.role == 0 && L.first_col > 0 && L.lower && L.row_off == L.col_off && L.cyc_P <= 1 && L.M >= L.N && L.N / 64 > L.first_col));
  if ((L.signal || L.wait_flag) && !(L.small_tile && (L.K == 128 || L.K == 64)) && !big_signal) {
    gpx_set_error(h, "gemm_nt: signal is served by the one-pass kernel and the 64x64 look-ahead kernel only");
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
    g.map_S = -1; g.map_ntj = ntj; g.map_nti = nti;
    if (L.first_col > 0 && L.role == 0 && L.beta != 0.0 && L.signal && ntj > L.first_col) g.first_col = L.first_col;
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
