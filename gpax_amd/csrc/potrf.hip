// potrf.hip -- LDS-resident factorisation of one diagonal block (gfx950), as a dataflow kernel.
//
// One workgroup takes the NB x NB diagonal block A_kk of the right-looking blocked
// Cholesky (the `cho_factor` of gpax/gp.py:296 / the Cholesky inside
// dist.MultivariateNormal, gpax/gp.py:147-151), keeps it in LDS as 16 x 16 tiles, and
// produces BOTH
//     L_kk          (lower Cholesky factor of the block), and
//     W_k = L_kk^-1 (lower triangular),
// so that every other step of the factorisation (panel solve, posterior solves) is a
// plain MFMA GEMM against W_k instead of a triangular solve.
//
// The 16 x 16 leaf factors a diagonal tile with one matrix ROW per lane: the pivot column / pivot row are
// broadcast with DPP `row_newbcast` (no LDS, no readlane).  One wave issues one fp64 instruction every ~7.2 cycles
// whatever the EXEC mask (tools/dpp_probe), so the leaf is issue-bound at ~3.0-3.6k cycles and everything else in this
// file is arranged to keep the NEXT leaf's input ready when the current one ends (see "Dataflow form" below).
// Off-diagonal tiles live in LDS with the XOR swizzle  elem(r,c) -> r*16 + (c ^ (r & 14)), which makes both the
// row-fragment and the column-fragment ds_read_b64 patterns of the f64 MFMA conflict-free.
// (Round 2 retired the barrier-phased form of this kernel -- leaf | sub-panel solve by inv(L_jj) | rank-16 update, two
// workgroup barriers per 16 columns, 29 us per 128 x 128 block -- for the dataflow form, 22 us.)

#include <type_traits>

#include "gpx_internal.h"

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// broadcast lane L of every 16-lane row to the whole row (DPP row_newbcast, one v_mov_b64_dpp)
template <int L>
__device__ __forceinline__ double row_bcast64(double x) {
  return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + L, 0xf, 0xf, false);
}

// The DPP instructions below are written as `asm volatile` (hipcc neither pads hazards inside an asm string nor knows
// they are DPP reads): they keep the program order written here, and the one hazard that order leaves open -- a DPP
// read needs 2 wait states after a VALU write of its source VGPR -- is closed by dpp_fence() before each group instead
// of an s_nop per instruction (an s_nop 1 costs ~8 cycles here).  v_fmac_f64_dpp: 7.2 cycles issued from one wave, the
// same as a plain v_fma_f64; a v_mov_b64_dpp + v_fma_f64 pair costs 13.6 (tools/dpp_probe).
// Orders every earlier write of `x` before the statement and gives 2 wait states after it.
__device__ __forceinline__ void dpp_fence(double& x) { asm volatile("s_nop 1" : "+v"(x)); }

__device__ __forceinline__ int tix(int ib, int jb) { return ib * (ib + 1) / 2 + jb; }
__device__ __forceinline__ int eix(int r, int c) { return r * 16 + (c ^ (r & 14)); }

// Diagnostic build only (tools/potrf_stamps): s_memtime stamps per phase, written to a buffer of
// their own; in the product build GPX_STAMP expands to nothing.
#ifdef GPX_POTRF_STAMPS
__device__ unsigned long long* g_stamps;
#define GPX_STAMP(slot)                                                              \
  do {                                                                               \
    if (threadIdx.x == 0) g_stamps[slot] = __builtin_amdgcn_s_memtime();             \
  } while (0)
#define GPX_STAMP_T(slot, t)                                                         \
  do {                                                                               \
    if (threadIdx.x == (t)) g_stamps[slot] = __builtin_amdgcn_s_memtime();           \
  } while (0)
#else
#define GPX_STAMP(slot) do { } while (0)
#define GPX_STAMP_T(slot, t) do { } while (0)
#endif

// ---- tile-level building blocks (one wave each; all operands are LDS tiles) ----

// X = A * inv(L_jj)^T  in place:  A[i][k], B[k][j] = Wd[j][k]; two accumulators halve the
// dependent-MFMA chain (a dependent f64 MFMA issues every ~190 cycles, an independent one every ~100)
__device__ __forceinline__ void tile_trsm(double* At, const double* Wd, int fr, int fk) {
  v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(At[eix(fr, fk)], Wd[eix(fr, fk)], acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(At[eix(fr, 4 + fk)], Wd[eix(fr, 4 + fk)], acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(At[eix(fr, 8 + fk)], Wd[eix(fr, 8 + fk)], acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(At[eix(fr, 12 + fk)], Wd[eix(fr, 12 + fk)], acc1, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) At[eix(fk + 4 * r, fr)] = acc0[r] + acc1[r];
}

// readers consume the leaf's columns in groups {0-3} {4-7} {8-11} {12,13} {14} {15}: one LDS round trip per
// group, short groups at the end so that they finish right behind the leaf
__host__ __device__ constexpr int leaf_group_start(int j) { return j < 12 ? (j & ~3) : (j < 14 ? 12 : j); }
__host__ __device__ constexpr int leaf_group_size(int j) { return j < 12 ? 4 : (j < 14 ? 2 : 1); }
__host__ __device__ constexpr bool leaf_group_end(int j) { return j == leaf_group_start(j) + leaf_group_size(j) - 1; }

template <int L>
__device__ __forceinline__ double pivot_bcast(double x) {   // volatile: keeps its place among the fmacs
  double r;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "i"(L));
  return r;
}
template <int L>
__device__ __forceinline__ double pivot_bcast_padded(double x) {   // with the 2 wait states inside
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "i"(L));
  return r;
}
// acc -= row_newbcast<L>(src) * mul   (neg modifier on src1: no separate negation)
template <int L>
__device__ __forceinline__ void fnmac_bcast(double& acc, double src, double mul) {
  asm volatile("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(src), "v"(mul), "i"(L));
}

// one wave writes LDS tile (ib, jb) of `tiles` to the global matrix `dst` (leading dimension ld);
// for a diagonal tile of L the don't-care entries above the diagonal become zeros
// W_k is published with plain stores and ONE agent-scope release per wave in front of the flag (write-through stores
// of every W tile lengthened the kernel by a quarter).
template <bool WT>
__device__ __forceinline__ void store_tile(const double* tiles, double* dst, int64_t ld, int ib, int jb, int lane,
                                           bool zero_upper_of_diag) {
  const int r = lane >> 2, c4 = (lane & 3) * 4;
  const double* src = tiles + tix(ib, jb) * 256;
  double* out = dst + (int64_t)(ib * 16 + r) * ld + jb * 16 + c4;
  double v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] = src[eix(r, c4 + q)];
    if (zero_upper_of_diag && ib == jb && c4 + q > r) v[q] = 0.0;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (WT) __hip_atomic_store(&out[q], v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else out[q] = v[q];
  }
}

#ifdef GPX_TIMELINE
__device__ unsigned long long* g_tl_potrf;
void tl_set_potrf(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_potrf), &p, sizeof(p)); }
#define POTRF_TL(kind, a)                                                                       \
  do {                                                                                          \
    if (threadIdx.x == 0 && g_tl_potrf) {                                                       \
      const unsigned long long slot = atomicAdd(g_tl_potrf, 1ull);                              \
      if (slot < 16000) {                                                                       \
        unsigned long long* r = g_tl_potrf + 4 + 4 * slot;                                      \
        r[0] = (kind); r[1] = __builtin_amdgcn_s_memrealtime(); r[2] = (unsigned long long)(a); r[3] = 0; \
      }                                                                                         \
    }                                                                                           \
  } while (0)
#else
#define POTRF_TL(kind, a) do { } while (0)
#endif


// =====================================================================================================
// Dataflow form: no workgroup barrier inside the
// block, no phase structure.  Wave r OWNS block row r of the 16 x 16 tiling (NT = NB/16 waves):
//
//   stage jb < r   follower: while wave jb runs the leaf of tile (jb,jb) and publishes its columns, wave r
//                  forward-substitutes its tile (r,jb) behind it column by column (stream_solve: X = A L_jj^-T, one
//                  tile COLUMN per lane, the 16 rows split 4 x 4 over the DPP rows, so a column costs one multiply
//                  and four v_fmac_f64_dpp): the sub-panel is final a few hundred cycles after the leaf's last
//                  column and nobody waits for inv(L_jj).  The tile it will solve NEXT, (r,jb+1), is brought up to date
//                  left-looking: the terms of columns < jb before the solve (they are final), the term of column jb
//                  after it -- every row does jb+2 MFMA terms at stage jb, the work is the same for all rows.
//   stage r-1      the same solve, but the update of the diagonal tile (r,r) is fed to the MFMA four columns at a
//                  time while the solve is still streaming: the next leaf's input is complete one LDS round trip
//                  after the previous leaf's last column.
//   stage r        leader: flow_leaf on (r,r); then inv(L_rr) by the same streaming solve on the identity.
//   stage jb > r   the row is finished: the wave owns COLUMN r of W = L^-1 and produces W(jb,r) =
//                  -inv(L_jj) * sum_kb L(jb,kb) W(kb,r)  as  MFMA sum + streaming solve.
//
// LDS: the strictly lower tiles of L (XOR-swizzled as above), the lower tiles of W, and one DIAGONAL AREA per stage:
// the diagonal tile stored column-major and symmetric (so the leaf reads its rows and the followers read its columns with
// consecutive lanes), the 16 reciprocals 1/l_jj, an always-zero slot and the stage's epoch (columns published so far).
// The leaf publishes IN PLACE: column j of L_jj overwrites column j of the tile.  Every stage has its own area, so a
// follower may lag as far as it likes.  Synchronisation is by LDS words only: the stage epochs and one "tile final" flag
// per L tile; every wait is bounded (info = -8 on exhaustion: a logic error, never seen).
#define FLOW_SPIN_MAX (1 << 22)
#define DT_RINV 256
#define DT_ZERO 272
#define DT_EPOCH 273
#define DT_STRIDE 280

typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) int lds_int;
// LDS accesses by byte offset.  Plain (not volatile: hipcc follows every volatile access with an s_waitcnt, which
// serialises a group of loads into one round trip each); the order that matters -- epoch before data on the reading
// side, data before epoch on the writing side, a fresh read per poll -- is kept with lds_order(), a compiler-level
// barrier (the LDS itself executes one wave's instructions in order).
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ double lds_ld(unsigned a) { return *(lds_double*)(size_t)a; }
__device__ __forceinline__ int lds_ldi(unsigned a) { return *(lds_int*)(size_t)a; }
__device__ __forceinline__ void lds_sti(unsigned a, int v) { *(lds_int*)(size_t)a = v; }
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(uintptr_t)p; }

__device__ __forceinline__ int tixs(int ib, int jb) { return ib * (ib - 1) / 2 + jb; }   // strictly lower tiles

__device__ __forceinline__ void flow_wait(const int* flag, int want, int* info) {
  const unsigned a = lds_off(flag);
  lds_order();
  int v = lds_ldi(a);
  for (int spin = 0; v < want && spin < FLOW_SPIN_MAX; ++spin) {
    __builtin_amdgcn_s_sleep(2);
    lds_order();
    v = lds_ldi(a);
  }
  if (v < want) atomicMin(info, -8);
  lds_order();
}
// after the wave's LDS writes of the data (LDS executes one wave's DS instructions in order)
__device__ __forceinline__ void flow_raise(int* flag, int value) {
  lds_order();
  lds_sti(lds_off(flag), value);
  lds_order();
}

// The 16 x 16 leaf: L_jj, one matrix row per lane (the 4 DPP rows of the wave work redundantly).  The rows are read from the
// symmetric column-major tile of the stage's diagonal area; per column: 1/sqrt(pivot) by v_rsq_f64 + one cubic (Halley) step
// (error O(e^3) ~ 1e-22, ~1 ulp after rounding; no denormal rescaling: pivots here are O(1e-6 .. 1e3)), the column scaled,
// published in place the moment it exists (l_ij by lane i, 1/l_jj, the epoch at the end of a reader group), then the
// rank-1 update of the columns to its right with the pivot row broadcast inside v_fmac_f64_dpp.  No predication at all:
// finished rows (i <= j) only ever corrupt their own above-diagonal entries, which are don't-care.  LAPACK-style info: the
// first non-positive (or NaN) pivot shows as the first diagonal entry that is not a positive number.
__device__ __forceinline__ void flow_leaf(double* dt, int lane, int* info, int gcol0) {
  const int i = lane & 15;
  double a[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = dt[c * 16 + i];
  const unsigned ep = lds_off(dt + DT_EPOCH);
  double d = row_bcast64<0>(a[0]);                   // pivot a_00
  static_for<0, 16>([&](auto J) {
    constexpr int j = decltype(J)::value;
    const double y = __builtin_amdgcn_rsq(d);
    const double t = d * y;
    const double e = fma(-t, y, 1.0);
    const double cc = fma(0.375, e, 0.5);
    const double ye = y * e;
    const double rinv = fma(ye, cc, y);
    double l = a[j] * rinv;                          // column j of L; lane j: d * rinv = l_jj
    a[j] = l;
    dt[j * 16 + i] = l;
    dt[DT_RINV + j] = rinv;
    if constexpr (leaf_group_end(j)) {
      lds_order();
      lds_sti(ep, j + 1);
      lds_order();
    }
    dpp_fence(l);                                    // l was just written by a VALU multiply
    if constexpr (j + 1 < 16) fnmac_bcast<j + 1>(a[j + 1], l, l);
    if constexpr (j + 2 < 16) fnmac_bcast<j + 2>(a[j + 2], l, l);
    if constexpr (j + 3 < 16) fnmac_bcast<j + 3>(a[j + 3], l, l);
    if constexpr (j + 3 < 16) d = pivot_bcast<j + 1>(a[j + 1]);
    else if constexpr (j + 1 < 16) d = pivot_bcast_padded<j + 1>(a[j + 1]);
    static_for<j + 4, 16>([&](auto C) {
      constexpr int c = decltype(C)::value;
      fnmac_bcast<c>(a[c], l, l);
    });
  });
  const unsigned long long bad = __ballot(!(a[i] > 0.0)) & 0xffffull;
  // first failing pivot wins, and a negative code (-7 / -8, raised with atomicMin by another wave) is never overwritten
  if (bad && lane == 0) atomicCAS(info, 0, gcol0 + (int)__ffsll((long long)bad));
}

// Forward substitution  L_jj y = rhs  behind the leaf that is producing L_jj, NR right-hand sides at once.
// Lane (k = lane & 15, g = lane >> 4) holds rows k of the right-hand sides, columns g, g+4, g+8, g+12 (u[n][q]:
// column g + 4q) -- for a follower tile that is the MFMA D-layout of the TRANSPOSED tile, so X = A L_jj^-T comes out in
// the D-layout of X.  Unscaled rows:  u_k -= (l_kj / l_jj) u_j  (k > j),  y_k = u_k / l_kk.  Finished rows (k <= j) read
// l_kj from the always-zero slot, so their multiplier is exactly 0 with no select.
// Columns are consumed in groups {0-3} {4-7} {8-11} {12,13} {14} {15}; the NEXT group's loads are issued before the
// current group's arithmetic (when the leaf is far ahead they hit, when the solve is right behind the leaf they are
// simply repeated).  on_final(S, rk) is called as soon as lanes 4S..4S+3 hold their final (unscaled) rows; rk = 1/l_kk.
struct FlowGroup { int epoch; double l[4], r[4], rk; };
template <int J0, int CNT>
__device__ __forceinline__ void flow_fetch(FlowGroup& g, unsigned dt, const unsigned (&la)[16], unsigned rk_addr) {
  lds_order();
  g.epoch = lds_ldi(dt + DT_EPOCH * 8);
  lds_order();
  if constexpr (J0 < 15) {   // column 15 eliminates nothing: only its epoch and the lanes' own 1/l_kk are needed
#pragma unroll
    for (int c = 0; c < CNT; ++c) { g.l[c] = lds_ld(la[J0 + c]); g.r[c] = lds_ld(dt + (DT_RINV + J0 + c) * 8); }
  }
  g.rk = lds_ld(rk_addr);
  lds_order();
}

// EAGER (the wave that leads next): a poll re-reads the whole group, so the data arrive with the epoch; otherwise a
// poll reads the epoch word only (seven followers polling whole groups would keep the LDS busy) and the group is
// fetched once it is there.
template <int NR, bool EAGER, typename F>
__device__ __forceinline__ void stream_solve(double* dtp, double (&u)[NR][4], int lane, int* info, F&& on_final) {
  const int k = lane & 15;
  const unsigned dt = lds_off(dtp);
  const unsigned rk_addr = dt + (DT_RINV + k) * 8;
  unsigned la[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) la[j] = dt + ((k > j) ? (j * 16 + k) * 8 : DT_ZERO * 8);
  // the right-hand sides may have been produced by VALU instructions that hipcc is free to sink down to their first
  // use, a DPP read inside an asm: pin them two wait states ahead
#pragma unroll
  for (int n = 0; n < NR; ++n) asm volatile("s_nop 1" : "+v"(u[n][0]), "+v"(u[n][1]), "+v"(u[n][2]), "+v"(u[n][3]));
  // every group is requested up front: when the leaf is (partly) ahead, one LDS round trip serves all that is published;
  // a group whose epoch was not there yet is requested again when its turn comes
  FlowGroup g[6];
  flow_fetch<0, 4>(g[0], dt, la, rk_addr);
  flow_fetch<4, 4>(g[1], dt, la, rk_addr);
  flow_fetch<8, 4>(g[2], dt, la, rk_addr);
  flow_fetch<12, 2>(g[3], dt, la, rk_addr);
  flow_fetch<14, 1>(g[4], dt, la, rk_addr);
  flow_fetch<15, 1>(g[5], dt, la, rk_addr);
  static_for<0, 16>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int j0 = leaf_group_start(j), cnt = leaf_group_size(j);
    constexpr int gi = j < 12 ? j / 4 : (j < 14 ? 3 : j - 10);
    if constexpr (j == j0) {
      if (__any(g[gi].epoch < j0 + cnt)) {
        int spin = 0;
        if constexpr (EAGER) {
#pragma unroll 1
          do {
            if (++spin == FLOW_SPIN_MAX) { atomicMin(info, -8); break; }
            __builtin_amdgcn_s_sleep(1);
            flow_fetch<j0, cnt>(g[gi], dt, la, rk_addr);
          } while (__any(g[gi].epoch < j0 + cnt));
        } else {
          int e;
#pragma unroll 1
          do {
            if (++spin == FLOW_SPIN_MAX) { atomicMin(info, -8); break; }
            __builtin_amdgcn_s_sleep(2);
            lds_order();
            e = lds_ldi(dt + DT_EPOCH * 8);
            lds_order();
          } while (__any(e < j0 + cnt));
          flow_fetch<j0, cnt>(g[gi], dt, la, rk_addr);
        }
      }
    }
    if constexpr (j < 15) {   // column 15 has no row below it
      const double m = g[gi].l[j - j0] * g[gi].r[j - j0];      // l_kj / l_jj, exactly 0 for k <= j
#pragma unroll
      for (int n = 0; n < NR; ++n) {
        fnmac_bcast<j>(u[n][0], u[n][0], m);
        fnmac_bcast<j>(u[n][1], u[n][1], m);
        fnmac_bcast<j>(u[n][2], u[n][2], m);
        fnmac_bcast<j>(u[n][3], u[n][3], m);
      }
    }
    if constexpr (j == 2 || j == 6 || j == 10) on_final(std::integral_constant<int, j / 4>{}, g[gi].rk);
    if constexpr (j == 15) on_final(std::integral_constant<int, 3>{}, g[gi].rk);
  });
}

// store the factorised diagonal tile (column-major in its diagonal area) to the global L block; zeros above the diagonal
__device__ __forceinline__ void flow_store_diag(const double* dt, double* dst, int64_t ld, int w, int lane) {
  const int r = lane >> 2, c4 = (lane & 3) * 4;
  double* out = dst + (int64_t)(w * 16 + r) * ld + w * 16 + c4;
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = (c4 + q <= r) ? dt[(c4 + q) * 16 + r] : 0.0;
}
// one wave writes strictly-lower LDS tile (ib, jb) of L to the global matrix
__device__ __forceinline__ void flow_store_lower(const double* Lt, double* dst, int64_t ld, int ib, int jb, int lane) {
  const int r = lane >> 2, c4 = (lane & 3) * 4;
  const double* src = Lt + tixs(ib, jb) * 256;
  double* out = dst + (int64_t)(ib * 16 + r) * ld + jb * 16 + c4;
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = src[eix(r, c4 + q)];
}

// WT: W_k is published to an already-resident consumer (done_flag): its stores are write-through (agent scope), so that
// the flag can go up behind a plain `s_waitcnt vmcnt(0)`.  Round 2 used plain stores and ONE agent-scope release for the
// block; measured in situ (profiles/r03_critical_path_*.txt) that release -- a write-back of the XCD's whole L2 -- takes
// ~10 us while a bulk update is filling that L2 with dirty C tiles, i.e. in every step of the first two thirds of a
// factorisation (the band kernel started 12-14 us after potrf_diag's last instruction there, 2-3 us in the last steps).
template <int NB, bool WT>
__device__ __forceinline__ void potrf_flow(const double* A, int64_t lda, double* __restrict__ L, int64_t ldl,
                                           double* __restrict__ W, int* info, int col0, const unsigned* wait_flag,
                                           unsigned wait_count, unsigned* done_flag, unsigned long long wait_ticks, double* sm) {
  constexpr int NT = NB / 16, NLOW = NT * (NT - 1) / 2, NTRI = NT * (NT + 1) / 2, NW = NT;
  double* Lt = sm;                                             // strictly lower tiles of L
  double* Wt = sm + NLOW * 256;                                // lower tiles of W
  double* DT = Wt + NTRI * 256;                                // diagonal areas, one per stage
  int* fin = reinterpret_cast<int*>(DT + NT * DT_STRIDE);      // [NLOW] strictly lower tile of L final in LDS
  int* winv = fin + NLOW;                                      // [NT] inv(L_jj) is in its W tile
  static_assert(NT >= 4, "dataflow potrf_diag: the stage-0 inverse is streamed by the last row, which must not be row 2");

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  // 1/l_jj, zero slots, epochs and flags start at zero (the tiles themselves are written by their owners) -- cleared BEFORE the
  // resident wait: a block that starts asleep has nothing else to do, and behind the flag only the loads are left
  for (int t = tid; t < NT * (DT_STRIDE - DT_RINV); t += NW * 64)
    DT[(t / (DT_STRIDE - DT_RINV)) * DT_STRIDE + DT_RINV + t % (DT_STRIDE - DT_RINV)] = 0.0;
  for (int t = tid; t < (NLOW + NT + 1) / 2 + 1; t += NW * 64) reinterpret_cast<double*>(fin)[t] = 0.0;
  if (wait_flag) {
    POTRF_TL(5, col0);
    if (tid == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_count) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > wait_ticks) { atomicMin(info, -7); break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
  POTRF_TL(1, col0);
  GPX_STAMP(0);

  // ---- wave w loads block row w (tiles (w, 0..w)); the loads are in flight across the barrier
  const int r4 = lane >> 2, c4 = (lane & 3) * 4;
  double* const dtw = DT + w * DT_STRIDE;
  {
    double v[NT][4];
#pragma unroll
    for (int jb = 0; jb < NT; ++jb) {
      if (jb <= w) {
        const double* src = A + (int64_t)(w * 16 + r4) * lda + jb * 16 + c4;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[jb][q] = src[q];
      }
    }
    __syncthreads();   // diagonal areas' control words and the flags are zero
    GPX_STAMP(1);
#pragma unroll
    for (int jb = 0; jb < NT; ++jb) {
      if (jb < w) {
        double* dst = Lt + tixs(w, jb) * 256;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[eix(r4, c4 + q)] = v[jb][q];
      } else if (jb == w) {
        // only the lower triangle of A_kk is maintained by the factorisation: mirror it
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c4 + q;
          if (c <= r4) { dtw[c * 16 + r4] = v[jb][q]; dtw[r4 * 16 + c] = v[jb][q]; }
        }
      }
    }
  }
  // ---- follower stages
#pragma unroll 1
  for (int jb = 0; jb < w; ++jb) {
    double* dts = DT + jb * DT_STRIDE;
    double* Xt = Lt + tixs(w, jb) * 256;
    GPX_STAMP_T(64 + w * 32 + jb * 4, 64 * w);
    if (jb == w - 1) {
      // last stage before this wave leads: the update of (w,w) rides along, four columns at a time
      __builtin_amdgcn_s_setprio(2);
      double u[1][4];
      v4d acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) { u[0][q] = Xt[eix(fk + 4 * q, fr)]; acc[q] = dtw[(fk + 4 * q) * 16 + fr]; }
      double xp = 0.0;   // the MFMA of a column group is issued one group later: its operand's LDS round trip hides
      stream_solve<1, true>(dts, u, lane, info, [&](auto S, double rk) {
        constexpr int s = decltype(S)::value;
        if constexpr (s > 0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xp, xp, acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) Xt[eix(fk + 4 * q, fr)] = u[0][q] * rk;
        xp = Xt[eix(fr, 4 * s + fk)];
      });
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xp, xp, acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) dtw[(fk + 4 * q) * 16 + fr] = acc[q];
      GPX_STAMP_T(64 + w * 32 + jb * 4 + 1, 64 * w);
      flow_raise(&fin[tixs(w, jb)], 1);
      continue;
    }
    // Invariant at the start of stage jb: (w,jb) is complete, (w,jb+1) lacks the terms of columns jb-1 and jb.
    // (P1) column jb-1 into (w,jb+1); it stays in registers until (U1)
    double* C1 = Lt + tixs(w, jb + 1) * 256;
    v4d a0, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) a0[q] = C1[eix(fk + 4 * q, fr)];
    if (jb >= 1) {
      flow_wait(&fin[tixs(jb + 1, jb - 1)], 1, info);
      const double* Ap = Lt + tixs(w, jb - 1) * 256;
      const double* Bp = Lt + tixs(jb + 1, jb - 1) * 256;
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, fk)], Bp[eix(fr, fk)], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, 4 + fk)], Bp[eix(fr, 4 + fk)], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, 8 + fk)], Bp[eix(fr, 8 + fk)], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, 12 + fk)], Bp[eix(fr, 12 + fk)], a1, 0, 0, 0);
      // (P2) the columns < jb into (w,jb+2), two stages ahead of its solve (row jb+2 raises its flags in column order)
      if (jb + 2 < w) {
        double* C2 = Lt + tixs(w, jb + 2) * 256;
        v4d c0, c1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q) c0[q] = C2[eix(fk + 4 * q, fr)];
        flow_wait(&fin[tixs(jb + 2, jb - 1)], 1, info);
        for (int kb = 0; kb < jb; ++kb) {
          const double* Ak = Lt + tixs(w, kb) * 256;
          const double* Bk = Lt + tixs(jb + 2, kb) * 256;
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ak[eix(fr, fk)], Bk[eix(fr, fk)], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ak[eix(fr, 4 + fk)], Bk[eix(fr, 4 + fk)], c1, 0, 0, 0);
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ak[eix(fr, 8 + fk)], Bk[eix(fr, 8 + fk)], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ak[eix(fr, 12 + fk)], Bk[eix(fr, 12 + fk)], c1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) C2[eix(fk + 4 * q, fr)] = c0[q] + c1[q];
      }
    }
    GPX_STAMP_T(64 + w * 32 + jb * 4 + 3, 64 * w);
    // (S) the solve of (w,jb): the row that leads after the next one follows the leaf column by column; the rows
    // further down have time to wait for inv(L_jj) and multiply (stage 0: the last row streams that inverse itself)
    if (jb == w - 2) {
      double u[1][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) u[0][q] = Xt[eix(fk + 4 * q, fr)];
      stream_solve<1, false>(dts, u, lane, info, [&](auto S, double rk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Xt[eix(fk + 4 * q, fr)] = u[0][q] * rk;
      });
    } else {
      double* Wd = Wt + tix(jb, jb) * 256;
      if (jb == 0 && w == NT - 1) {
        double u[1][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) u[0][q] = (fr == fk + 4 * q) ? 1.0 : 0.0;
        stream_solve<1, false>(dts, u, lane, info, [&](auto S, double rk) {
#pragma unroll
          for (int q = 0; q < 4; ++q) Wd[eix(fr, fk + 4 * q)] = u[0][q] * rk;
        });
        flow_raise(&winv[0], 1);
        store_tile<WT>(Wt, W, NB, 0, 0, lane, false);
      } else {
        flow_wait(&winv[jb], 1, info);
      }
      tile_trsm(Xt, Wd, fr, fk);
    }
    flow_raise(&fin[tixs(w, jb)], 1);
    GPX_STAMP_T(64 + w * 32 + jb * 4 + 1, 64 * w);
    // (U1) column jb into (w,jb+1): complete;  (U2) the same column on the diagonal tile
    double xa[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) xa[s] = -Xt[eix(fr, 4 * s + fk)];
    v4d d0, d1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) d0[q] = dtw[(fk + 4 * q) * 16 + fr];
    flow_wait(&fin[tixs(jb + 1, jb)], 1, info);
    const double* Bp = Lt + tixs(jb + 1, jb) * 256;
    double b[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = Bp[eix(fr, 4 * s + fk)];
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[0], b[0], a0, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[0], -xa[0], d0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[1], b[1], a1, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[1], -xa[1], d1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2], b[2], a0, 0, 0, 0);
    d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[2], -xa[2], d0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[3], b[3], a1, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[3], -xa[3], d1, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { C1[eix(fk + 4 * q, fr)] = a0[q] + a1[q]; dtw[(fk + 4 * q) * 16 + fr] = d0[q] + d1[q]; }
    GPX_STAMP_T(64 + w * 32 + jb * 4 + 2, 64 * w);
    flow_store_lower(Lt, L, ldl, w, jb, lane);
  }

  // ---- leader stage: the leaf of (w,w)
  {
    __builtin_amdgcn_s_setprio(3);
    GPX_STAMP_T(2 + 2 * w, 64 * w);
    flow_leaf(dtw, lane, info, col0 + 16 * w);
    GPX_STAMP_T(3 + 2 * w, 64 * w);
    __builtin_amdgcn_s_setprio(0);
    flow_store_diag(dtw, L, ldl, w, lane);
    if (w >= 1) flow_store_lower(Lt, L, ldl, w, w - 1, lane);
    // zeros above the diagonal of the L block and of W (nothing depends on them)
    for (int jb = w + 1; jb < NT; ++jb) {
      double* ldst = L + (int64_t)(w * 16 + r4) * ldl + jb * 16 + c4;
      double* wdst = W + (int64_t)(w * 16 + r4) * NB + jb * 16 + c4;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ldst[q] = 0.0;
        if constexpr (WT) __hip_atomic_store(&wdst[q], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else wdst[q] = 0.0;
      }
    }
  }

  // ---- the row is finished: column w of W, one tile per remaining stage.  The wave whose row finished last also
  // streams inv(L_jj) behind the leaf (it has a single term to sum) and announces it.
#pragma unroll 1
  for (int jb = w + 1; jb < NT; ++jb) {
    double* dts = DT + jb * DT_STRIDE;
    double* Tp = Wt + tix(jb, w) * 256;
    double* Wd = Wt + tix(jb, jb) * 256;
    // T = sum_kb L(jb,kb) W(kb,w)   (row jb raises its flags in column order)
    v4d a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
    flow_wait(&fin[tixs(jb, jb - 1)], 1, info);
    if (jb == w + 1) flow_wait(&winv[w], 1, info);   // W(w,w) was streamed by another wave
    for (int kb = w; kb < jb; ++kb) {
      const double* Lp = Lt + tixs(jb, kb) * 256;
      const double* Wp = Wt + tix(kb, w) * 256;
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, fk)], Wp[eix(fk, fr)], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, 4 + fk)], Wp[eix(4 + fk, fr)], a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, 8 + fk)], Wp[eix(8 + fk, fr)], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, 12 + fk)], Wp[eix(12 + fk, fr)], a1, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) Tp[eix(fk + 4 * q, fr)] = a0[q] + a1[q];
    if (w == jb - 1) {
      double u[1][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) u[0][q] = (fr == fk + 4 * q) ? 1.0 : 0.0;
      stream_solve<1, false>(dts, u, lane, info, [&](auto S, double rk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) Wd[eix(fr, fk + 4 * q)] = u[0][q] * rk;
      });
      flow_raise(&winv[jb], 1);
    } else {
      flow_wait(&winv[jb], 1, info);
    }
    // W(jb,w) = -inv(L_jj) * T   (same-wave LDS round trip of T as the B operand)
    v4d p0 = {0.0, 0.0, 0.0, 0.0}, p1 = {0.0, 0.0, 0.0, 0.0};
    p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, fk)], Tp[eix(fk, fr)], p0, 0, 0, 0);
    p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, 4 + fk)], Tp[eix(4 + fk, fr)], p1, 0, 0, 0);
    p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, 8 + fk)], Tp[eix(8 + fk, fr)], p0, 0, 0, 0);
    p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, 12 + fk)], Tp[eix(12 + fk, fr)], p1, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) Tp[eix(fk + 4 * q, fr)] = p0[q] + p1[q];
    store_tile<WT>(Wt, W, NB, jb, w, lane, false);
    if (w == jb - 1) store_tile<WT>(Wt, W, NB, jb, jb, lane, false);
  }

  GPX_STAMP_T(2 + 2 * NT, 64 * (NT - 1));
  POTRF_TL(2, col0);
  // publish W_k to an already-resident consumer (the band kernel, which reads W_k and nothing else of this block: L_kk
  // reaches its readers through this kernel's completion event): every wave drains its write-through stores, the barrier
  // collects the waves, one lane raises the flag
  if (done_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if constexpr (!WT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(done_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    POTRF_TL(7, col0);
  }
  GPX_STAMP_T(3 + 2 * NT, 64 * (NT - 1));
}

struct PotrfArgs {
  const double* A; int64_t lda; double* L; int64_t ldl; double* W; int* info; int col0;
  const unsigned* wait_flag; unsigned wait_count; unsigned* done_flag; unsigned long long wait_ticks;
  int64_t sA, sL, sW, s_info, s_flag;   // per-draw strides of a batched launch (blockIdx.x = draw)
};

template <int NB, int NWAVES, bool WT>
__global__ __launch_bounds__(NWAVES * 64) void potrf_diag_kernel(PotrfArgs p) {
  constexpr int NT = NB / 16, NTRI = NT * (NT + 1) / 2;
  static_assert(NWAVES == NT, "potrf_diag: one wave per block row");
  __shared__ double sm[NT * NT * 256 + NT * DT_STRIDE + (NTRI + NT) / 2 + 2];
  const int64_t z = blockIdx.x;
  potrf_flow<NB, WT>(p.A + z * p.sA, p.lda, p.L + z * p.sL, p.ldl, p.W + z * p.sW, p.info + z * p.s_info, p.col0,
                 p.wait_flag ? p.wait_flag + z * p.s_flag : nullptr, p.wait_count,
                 p.done_flag ? p.done_flag + z * p.s_flag : nullptr, p.wait_ticks, sm);
}

// ---- the whole chain of diagonal blocks in ONE launch ("resident" = 3) ---------------------------------------------------
// potrf_diag needs a whole CU (149 KB of LDS, 8 waves x 233 VGPRs).  Launched per block -- even a step early -- the next one
// has to be GIVEN a CU while a bulk update saturates the chip (three 37-KB workgroups per CU, refilled one by one as they
// finish): measured with the timeline build (profiles/r03_critical_path_row2.txt), potrf_diag(k+1) became resident 18-39 us
// after potrf_diag(k) had ended in the first half of an N = 4096 factorisation, 4 us in the last steps -- with every one of
// its inputs long there.  (hipExtAnyOrderLaunch, which would let it become resident beside its predecessor, is ignored on
// gfx950: tools/anyorder_probe.)  One workgroup per draw therefore keeps its CU for the whole factorisation and walks
// down the diagonal: block k waits for flagsU[k-1] (the band kernel's tiles of A_kk), factors, publishes W_k (write-through)
// and raises flagsW[k].  The band kernel polls flagsW[k] as before; the panel stream, which used to wait for this kernel's
// completion event, gets a one-wave relay kernel that sleeps on the same flag (chain_relay_kernel).  L_kk itself has no
// reader before the end of the factorisation.
struct PotrfChainArgs {
  const double* A; int64_t lda; double* L; int64_t ldl; double* W; int* info;
  const unsigned* flagsU; unsigned* flagsW;
  int k0, k1;             // blocks [k0, k1)
  int raise_until;        // flagsW[k] is raised for k < raise_until
  int force_k;            // test hook: block force_k waits for an impossible count (gives up at once: info = -7)
  unsigned wait_count;
  unsigned long long wait_ticks;
  int64_t sA, sL, sW, s_info, s_flag;
};

template <int NB, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void potrf_chain_kernel(PotrfChainArgs p) {
  constexpr int NT = NB / 16, NTRI = NT * (NT + 1) / 2;
  static_assert(NWAVES == NT, "potrf_diag: one wave per block row");
  __shared__ double sm[NT * NT * 256 + NT * DT_STRIDE + (NTRI + NT) / 2 + 2];
  __shared__ int stop;
  const int64_t z = blockIdx.x;
  int* info = p.info + z * p.s_info;
  for (int k = p.k0; k < p.k1; ++k) {
    const int64_t off = (int64_t)k * NB;
    const unsigned* wf = k >= 1 ? p.flagsU + (k - 1) + z * p.s_flag : nullptr;
    unsigned* df = k < p.raise_until ? p.flagsW + k + z * p.s_flag : nullptr;
    const bool forced = k == p.force_k && wf;
    potrf_flow<NB, true>(p.A + z * p.sA + off * p.lda + off, p.lda, p.L + z * p.sL + off * p.ldl + off, p.ldl,
                         p.W + z * p.sW + (int64_t)k * NB * NB, info, (int)off, wf, forced ? 0xffffffffu : p.wait_count, df,
                         forced ? 2000ull : p.wait_ticks, sm);
    // a wait that gave up (-7) or an internal error (-8): the remaining blocks would each wait their bound out
    if (threadIdx.x == 0) stop = __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0;
    __syncthreads();   // also: every wave is done with the LDS image before the next block re-initialises it
    if (stop) break;
    __syncthreads();   // `stop` is rewritten by the next round
  }
}

// One wave per draw: sleeps until *flag >= count, then ends -- the kernel boundary behind it is what releases the next
// kernel of its stream (the panel of step k behind W_k).
__global__ __launch_bounds__(64) void chain_relay_kernel(const unsigned* flag, unsigned count, int* info, int64_t s_flag, int64_t s_info,
                                                         unsigned long long wait_ticks) {
  flag += (int64_t)blockIdx.x * s_flag; info += (int64_t)blockIdx.x * s_info;
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < count) {
      __builtin_amdgcn_s_sleep(8);
      if (__builtin_amdgcn_s_memrealtime() - t0 > wait_ticks) { atomicMin(info, -7); break; }
    }
  }
}

int launch_chain_relay(gpx_handle* h, hipStream_t s, const unsigned* flag, unsigned count, int* info, int batch, int64_t s_flag,
                       int64_t s_info) {
  hipLaunchKernelGGL(chain_relay_kernel, dim3((unsigned)batch), dim3(64), 0, s, flag, count, info, s_flag, s_info, 25000000ull);
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

int launch_potrf_chain(gpx_handle* h, hipStream_t s, const PotrfChainLaunch& q) {
  PotrfChainArgs p;
  p.A = q.A; p.lda = q.lda; p.L = q.L; p.ldl = q.ldl; p.W = q.W; p.info = q.info;
  p.flagsU = q.flagsU; p.flagsW = q.flagsW; p.k0 = q.k0; p.k1 = q.k1; p.raise_until = q.raise_until; p.force_k = q.force_k;
  p.wait_count = q.wait_count; p.wait_ticks = q.wait_ticks;
  p.sA = q.sA; p.sL = q.sL; p.sW = q.sW; p.s_info = q.s_info; p.s_flag = q.s_flag;
  if (q.batch < 1 || q.k1 <= q.k0 || !q.flagsU || !q.flagsW) { gpx_set_error(h, "potrf_chain: contract violated"); return GPX_ERR_ARG; }
  const dim3 grid((unsigned)q.batch);
  if (q.nb == 128) hipLaunchKernelGGL((potrf_chain_kernel<128, 8>), grid, dim3(512), 0, s, p);
  else if (q.nb == 64) hipLaunchKernelGGL((potrf_chain_kernel<64, 4>), grid, dim3(256), 0, s, p);
  else { gpx_set_error(h, "potrf_chain: nb must be 64 or 128"); return GPX_ERR_ARG; }
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}

int launch_potrf_diag(gpx_handle* h, hipStream_t s, const PotrfLaunch& q) {
  PotrfArgs p;
  p.A = q.A; p.lda = q.lda; p.L = q.L; p.ldl = q.ldl; p.W = q.W; p.info = q.info; p.col0 = q.col0;
  p.wait_flag = q.wait_flag; p.wait_count = q.wait_count; p.done_flag = q.done_flag; p.wait_ticks = q.wait_ticks;
  p.sA = q.sA; p.sL = q.sL; p.sW = q.sW; p.s_info = q.s_info; p.s_flag = q.s_flag;
  if (q.batch < 1) { gpx_set_error(h, "potrf_diag: batch < 1"); return GPX_ERR_ARG; }
  const dim3 grid((unsigned)q.batch);
  switch (q.nb) {
    case 64:
      if (q.done_flag) hipExtLaunchKernelGGL((potrf_diag_kernel<64, 4, true>), grid, dim3(256), 0, s, nullptr, q.stop_event, 0, p);
      else hipExtLaunchKernelGGL((potrf_diag_kernel<64, 4, false>), grid, dim3(256), 0, s, nullptr, q.stop_event, 0, p);
      break;
    case 128:
      if (q.done_flag) hipExtLaunchKernelGGL((potrf_diag_kernel<128, 8, true>), grid, dim3(512), 0, s, nullptr, q.stop_event, 0, p);
      else hipExtLaunchKernelGGL((potrf_diag_kernel<128, 8, false>), grid, dim3(512), 0, s, nullptr, q.stop_event, 0, p);
      break;
    default:
      gpx_set_error(h, "potrf_diag: nb must be 64 or 128");
      return GPX_ERR_ARG;
  }
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}
