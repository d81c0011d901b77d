// potrf.hip -- LDS-resident factorisation of one diagonal block (gfx950).
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
// Per 16-column sub-panel jb:
//   leaf   one wave factors the 16x16 diagonal tile with one matrix ROW per lane: the
//          pivot column / pivot row are broadcast with DPP `row_newbcast` (no LDS, no
//          readlane), and the same sweep forward-substitutes the identity, giving
//          inv(L_jj) for free;
//   phase A  (all waves, one tile each)  L_ib,jb = A_ib,jb * inv(L_jj)^T        [4 MFMA]
//            and row jb of W:  W_jb,cb = -inv(L_jj) * sum_kb L_jb,kb W_kb,cb   [4(jb-cb)+4 MFMA]
//   phase B  trailing tiles  A_ib,lb -= L_ib,jb * L_lb,jb^T                     [4 MFMA]
// Tiles live un-padded in LDS (2 KiB each; 36 + 36 tiles = 144 KiB at NB = 128) with the
// XOR swizzle  elem(r,c) -> r*16 + (c ^ (r & 14)), which makes both the row-fragment
// and the column-fragment ds_read_b64 patterns of the f64 MFMA conflict-free.

#include <type_traits>

#include "gpx_internal.h"

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// broadcast lane L of every 16-lane row to the whole row (DPP row_newbcast, gfx90a+)
template <int L>
__device__ __forceinline__ double row_bcast(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + L, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + L, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// 64-bit form (one v_mov_b64_dpp)
template <int L>
__device__ __forceinline__ double row_bcast64(double x) {
  return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + L, 0xf, 0xf, false);
}

// acc += row_newbcast<L>(src) * mul  as ONE instruction (v_fmac_f64_dpp: 7.2 cycles issued from one
// wave, the same as a plain v_fma_f64; a v_mov_b64_dpp + v_fma_f64 pair costs 13.6 -- tools/dpp_probe).
// hipcc neither pads hazards inside an asm string nor knows this is a DPP read, so the statements
// are `volatile` (they keep the program order written here) and the one hazard that order leaves
// open -- a DPP read needs 2 wait states after a VALU write of its source VGPR -- is closed by
// dpp_fence() before each group instead of an s_nop per instruction (an s_nop 1 costs ~8 cycles here).
template <int L>
__device__ __forceinline__ void fmac_bcast(double& acc, double src, double mul) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(src), "v"(mul), "i"(L));
}
// Orders every earlier write of `x` before the statement and gives 2 wait states after it.
__device__ __forceinline__ void dpp_fence(double& x) { asm volatile("s_nop 1" : "+v"(x)); }

// 1/sqrt(d) and sqrt(d) from v_rsq_f64 (measured 2^-24.2 relative accuracy on gfx950,
// tools/rsq_probe) and ONE cubic (Halley) step:  y <- y (1 + e/2 + 3e^2/8),  e = 1 - d y^2,
// error O(e^3) ~ 1e-22 -> ~1 ulp after rounding.  Dependency depth 4 fp64 ops after the rsq:
// this chain is the critical path of the whole factorisation (one per matrix column).
// No denormal rescaling: pivots here are O(1e-6 .. 1e3).
__device__ __forceinline__ void sqrt_rsqrt(double d, double& s, double& rinv) {
  const double y = __builtin_amdgcn_rsq(d);
  const double t = d * y;
  const double e = fma(-t, y, 1.0);
  const double c = fma(0.375, e, 0.5);
  const double ye = y * e;
  rinv = fma(ye, c, y);
  s = d * rinv;
}

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

// C -= A * B^T :  A[i][k] = -Ap[i][k], B[k][j] = Bp[j][k]
__device__ __forceinline__ void tile_syrk(double* Cp, const double* Ap, const double* Bp, int fr, int fk) {
  v4d acc0, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) acc0[r] = Cp[eix(fk + 4 * r, fr)];
  acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, fk)], Bp[eix(fr, fk)], acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, 4 + fk)], Bp[eix(fr, 4 + fk)], acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, 8 + fk)], Bp[eix(fr, 8 + fk)], acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Ap[eix(fr, 12 + fk)], Bp[eix(fr, 12 + fk)], acc1, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) Cp[eix(fk + 4 * r, fr)] = acc0[r] + acc1[r];
}

// Two independent C -= A * B^T tile updates in one instruction stream: the LDS-read -> MFMA -> LDS-write
// chain of a single tile is latency bound (~1.5k cycles), two interleaved chains hide each other.
__device__ __forceinline__ void tile_syrk2(double* C0, const double* A0, const double* B0, double* C1, const double* A1,
                                           const double* B1, int fr, int fk) {
  v4d p0, p1, q0 = {0.0, 0.0, 0.0, 0.0}, q1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r) { p0[r] = C0[eix(fk + 4 * r, fr)]; p1[r] = C1[eix(fk + 4 * r, fr)]; }
  double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    a0[s] = -A0[eix(fr, 4 * s + fk)]; b0[s] = B0[eix(fr, 4 * s + fk)];
    a1[s] = -A1[eix(fr, 4 * s + fk)]; b1[s] = B1[eix(fr, 4 * s + fk)];
  }
  p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], b0[0], p0, 0, 0, 0);
  p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], b1[0], p1, 0, 0, 0);
  q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], b0[1], q0, 0, 0, 0);
  q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], b1[1], q1, 0, 0, 0);
  p0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[2], b0[2], p0, 0, 0, 0);
  p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[2], b1[2], p1, 0, 0, 0);
  q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[3], b0[3], q0, 0, 0, 0);
  q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[3], b1[3], q1, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) { C0[eix(fk + 4 * r, fr)] = p0[r] + q0[r]; C1[eix(fk + 4 * r, fr)] = p1[r] + q1[r]; }
}

// W tile (jb, cb), cb < jb:  T = sum_{kb=cb}^{jb-1} L_jb,kb * W_kb,cb ;  W_jb,cb = -inv(L_jj) * T
__device__ __forceinline__ void tile_winv(double* Lt, double* Wt, int jb, int cb, int fr, int fk) {
  double* Tp = Wt + tix(jb, cb) * 256;
  const double* Wd = Wt + tix(jb, jb) * 256;
  v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  for (int kb = cb; kb < jb; ++kb) {
    const double* Lp = Lt + tix(jb, kb) * 256;
    const double* Wp = Wt + tix(kb, cb) * 256;
    // A[i][k] = L_jb,kb[i][k] , B[k][j] = W_kb,cb[k][j]
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, fk)], Wp[eix(fk, fr)], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, 4 + fk)], Wp[eix(4 + fk, fr)], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, 8 + fk)], Wp[eix(8 + fk, fr)], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Lp[eix(fr, 12 + fk)], Wp[eix(12 + fk, fr)], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) Tp[eix(fk + 4 * r, fr)] = acc0[r] + acc1[r];
  // same-wave LDS round trip (LDS operations of one wave execute in order): T as the B operand
  v4d a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
  a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, fk)], Tp[eix(fk, fr)], a0, 0, 0, 0);
  a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, 4 + fk)], Tp[eix(4 + fk, fr)], a1, 0, 0, 0);
  a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, 8 + fk)], Tp[eix(8 + fk, fr)], a0, 0, 0, 0);
  a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[eix(fr, 12 + fk)], Tp[eix(12 + fk, fr)], a1, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) Tp[eix(fk + 4 * r, fr)] = a0[r] + a1[r];
}

// ---- 16x16 leaf, split over TWO waves that sit on different SIMDs -------------------------
// One matrix row per lane (the 4 DPP rows of a wave work redundantly); pivot column / pivot row
// are broadcast by DPP.  A single wave issues one fp64 instruction every ~7.2 cycles, so the leaf
// is issue-bound: wave 0 runs only the Cholesky recurrence (the critical path of the whole
// factorisation) and publishes each finished column through LDS; wave 1 follows behind and
// forward-substitutes the identity with them, giving inv(L_jj).
//
// Hand-off area (LDS, 16 columns x 40 doubles; the 4 redundant DPP rows store identical values to
// the same slots, so no store is predicated): per column j   [0..15] l_ij by row   [16..31] 1/l_jj
// (16 copies)   [32..39] 16 x int epoch; plus one always-zero double behind the columns.  LDS executes a wave's DS
// instructions in order, so a reader that sees a group's epoch sees the group's data.  Epochs increase monotonically over the
// kernel (leaf jb, column j -> 16 jb + j + 1): nothing is ever reset.
#define LEAF_COL_DOUBLES 40
#define LEAF_ZERO_SLOT (16 * LEAF_COL_DOUBLES)
#define LEAF_SHARE_DOUBLES (16 * LEAF_COL_DOUBLES + 2)

// wave 1 consumes the columns in groups {0-3} {4-7} {8-11} {12,13} {14} {15}: one LDS round trip per
// group, short groups at the end so that it finishes right behind wave 0
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

// wave 0: L_jj.  No predication at all: finished rows (i <= j) only ever corrupt their own
// above-diagonal entries, which are don't-care; lane j gets l_jj = d * rinv from the same multiply.
__device__ __forceinline__ void leaf_potrf(double* Ld, double* share, int lane, int base_epoch, int* info,
                                           int gcol0) {
  const int i = lane & 15;
  double a[16];
  // eix(i, c) keeps even/odd column pairs adjacent (the XOR mask is even): 8 x 16-B accesses per row
#pragma unroll
  for (int c = 0; c < 16; c += 2) {
    const v2d v = *reinterpret_cast<const v2d*>(Ld + eix(i, c));
    a[c] = v.x; a[c + 1] = v.y;
  }
  double d = row_bcast64<0>(a[0]);                   // pivot a_00
  static_for<0, 16>([&](auto J) {
    constexpr int j = decltype(J)::value;
    // 1/sqrt(d): v_rsq_f64 + one cubic step (see sqrt_rsqrt)
    const double y = __builtin_amdgcn_rsq(d);
    const double t = d * y;
    const double e = fma(-t, y, 1.0);
    const double cc = fma(0.375, e, 0.5);
    const double ye = y * e;
    const double rinv = fma(ye, cc, y);
    double l = a[j] * rinv;                          // column j of L; lane j: d * rinv = l_jj
    a[j] = l;
    // publish column j (l per lane, then 1/l_jj per lane); its epoch only at the end of a
    // consumer group (leaf_group_end)
    share[j * LEAF_COL_DOUBLES + i] = l;
    share[j * LEAF_COL_DOUBLES + 16 + i] = rinv;
    if constexpr (leaf_group_end(j)) {
      asm volatile("" ::: "memory");
      reinterpret_cast<int*>(share + j * LEAF_COL_DOUBLES + 32)[i] = base_epoch + j + 1;
    }
    dpp_fence(l);                                    // l was just written by a VALU multiply
    // rank-1 update a_ic -= l_ij * l_cj (c > j).  c = j+1 first: it is the next pivot, whose
    // broadcast is issued as soon as the DPP-read hazard allows (two fmacs later).
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
  // LAPACK-style info: the first non-positive (or NaN) pivot shows as the first diagonal entry
  // that is not a positive number (NaN propagates to every later pivot)
  const unsigned long long bad = __ballot(!(a[i] > 0.0)) & 0xffffull;
  if (bad && lane == 0 && *info == 0) *info = gcol0 + __ffsll((long long)bad);
  // entries above the diagonal are don't-care here; the write-out phase zeroes them
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      v2d v; v.x = a[c]; v.y = a[c + 1];
      *reinterpret_cast<v2d*>(Ld + eix(i, c)) = v;
    }
  }
}

// One LDS round trip for a group of CNT columns starting at J0: the group's epoch, and (l_i, 1/l_jj)
// of every column (loads and their wait in ONE asm statement: hipcc does not count loads inside asm).
// la[] are per-lane byte addresses of l_ij, already redirected to the always-zero slot for finished
// rows (i <= j), so the multiplier of a finished row is exactly 0 with no select.
template <int J0, int CNT>
__device__ __forceinline__ void fetch_group(unsigned lds_base, unsigned ep_base, const unsigned (&la)[16], int& epoch,
                                            double (&l)[4], double (&r)[4]) {
  constexpr int B = LEAF_COL_DOUBLES * 8;
  if constexpr (CNT == 4) {
    asm volatile(
        "ds_read_b32 %0, %10 offset:%15\n\t"
        "ds_read_b64 %1, %11\n\tds_read_b64 %2, %9 offset:%16\n\t"
        "ds_read_b64 %3, %12\n\tds_read_b64 %4, %9 offset:%17\n\t"
        "ds_read_b64 %5, %13\n\tds_read_b64 %6, %9 offset:%18\n\t"
        "ds_read_b64 %7, %14\n\tds_read_b64 %8, %9 offset:%19\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(epoch), "=&v"(l[0]), "=&v"(r[0]), "=&v"(l[1]), "=&v"(r[1]), "=&v"(l[2]), "=&v"(r[2]), "=&v"(l[3]), "=&v"(r[3])
        : "v"(lds_base), "v"(ep_base), "v"(la[J0]), "v"(la[J0 + 1]), "v"(la[J0 + 2]), "v"(la[J0 + 3]), "i"((J0 + 3) * B),
          "i"(J0 * B + 128), "i"((J0 + 1) * B + 128), "i"((J0 + 2) * B + 128), "i"((J0 + 3) * B + 128)
        : "memory");
  } else if constexpr (CNT == 2) {
    asm volatile(
        "ds_read_b32 %0, %6 offset:%9\n\t"
        "ds_read_b64 %1, %7\n\tds_read_b64 %2, %5 offset:%10\n\t"
        "ds_read_b64 %3, %8\n\tds_read_b64 %4, %5 offset:%11\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(epoch), "=&v"(l[0]), "=&v"(r[0]), "=&v"(l[1]), "=&v"(r[1])
        : "v"(lds_base), "v"(ep_base), "v"(la[J0]), "v"(la[J0 + 1]), "i"((J0 + 1) * B), "i"(J0 * B + 128), "i"((J0 + 1) * B + 128)
        : "memory");
  } else {
    asm volatile(
        "ds_read_b32 %0, %4 offset:%6\n\t"
        "ds_read_b64 %1, %5\n\tds_read_b64 %2, %3 offset:%7\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(epoch), "=&v"(l[0]), "=&v"(r[0])
        : "v"(lds_base), "v"(ep_base), "v"(la[J0]), "i"(J0 * B), "i"(J0 * B + 128)
        : "memory");
  }
}

// wave 1: inv(L_jj) by forward substitution of the identity on the UNSCALED rows u_i = l_ii w_i:
//   u_i -= (l_ij / l_jj) u_j  (i > j);  w_i = u_i / l_ii once at the end.
// Entries above the diagonal are never touched (updates of row i only reach columns c <= j < i),
// so they stay exactly 0 and nothing is predicated.
__device__ __forceinline__ void leaf_winv(double* Wd, double* share, int lane, int base_epoch) {
  const int i = lane & 15;
  // per-lane LDS byte addresses (the low 32 bits of a generic LDS pointer are the LDS offset)
  const unsigned lds_base = (unsigned)(uintptr_t)(share + i);   // l_i of column 0; +128 B: 1/l_jj copy i
  const unsigned ep_base = (unsigned)(uintptr_t)(share + 32) + 4 * i;   // epoch copy i of column 0
  const unsigned zero_addr = (unsigned)(uintptr_t)(share + LEAF_ZERO_SLOT);
  unsigned la[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) la[j] = (i > j) ? lds_base + j * (LEAF_COL_DOUBLES * 8) : zero_addr;
  double w[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) w[c] = (c == i) ? 1.0 : 0.0;
  // hipcc may sink these VALU initialisations down to the first use, which is a DPP read inside an
  // asm it cannot see into: pin them here, two wait states ahead of every DPP read
  asm volatile("s_nop 1"
               : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]),
                 "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]));
  double gl[4], gr[4];
  static_for<0, 16>([&](auto J) {
    constexpr int j = decltype(J)::value;
    constexpr int j0 = leaf_group_start(j), cnt = leaf_group_size(j);
    if constexpr (j == j0) {
      int epoch;
      fetch_group<j0, cnt>(lds_base, ep_base, la, epoch, gl, gr);
      // Wave 0 publishes in order and never waits on this wave, so the group's last epoch arrives;
      // the spin bound only guards against a hang if that ever stopped being true.
      for (int spin = 0; __any(epoch != base_epoch + j0 + cnt) && spin < (1 << 20); ++spin) {
        __builtin_amdgcn_s_sleep(1);
        fetch_group<j0, cnt>(lds_base, ep_base, la, epoch, gl, gr);
      }
    }
    double m = gl[j - j0] * gr[j - j0];              // l_ij / l_jj (exactly 0 for finished rows)
    // DPP source of the group is w[c], last written by column j-1's group; only the first columns
    // have fewer than two instructions in between
    if constexpr (j <= 2) dpp_fence(w[0]);
    static_for<0, j + 1>([&](auto C) {
      constexpr int c = decltype(C)::value;
      fnmac_bcast<j>(w[c], w[c], m);                 // w[c] -= bcast_j(w[c]) * (l_ij / l_jj)
    });
  });
  const double my_rinv = share[i * LEAF_COL_DOUBLES + 16 + i];   // 1 / l_ii (all columns are published by now)
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      v2d v; v.x = w[c] * my_rinv; v.y = w[c + 1] * my_rinv;
      *reinterpret_cast<v2d*>(Wd + eix(i, c)) = v;
    }
  }
}

// one wave writes LDS tile (ib, jb) of `tiles` to the global matrix `dst` (leading dimension ld);
// for a diagonal tile of L the don't-care entries above the diagonal become zeros
// W_k is published with plain stores and ONE agent-scope release per wave in front of the flag (GPX_W_THROUGH 0):
// write-through stores of every W tile (GPX_W_THROUGH 1) lengthened the kernel from 29 to 36 us.
#ifndef GPX_W_THROUGH
#define GPX_W_THROUGH 0
#endif
// `through`: write-through (agent-scope) stores, for a block that is handed to an already-resident consumer
__device__ __forceinline__ void store_tile(const double* tiles, double* dst, int64_t ld, int ib, int jb, int lane,
                                           bool zero_upper_of_diag, bool through = false) {
  const int r = lane >> 2, c4 = (lane & 3) * 4;
  const double* src = tiles + tix(ib, jb) * 256;
  double* out = dst + (int64_t)(ib * 16 + r) * ld + jb * 16 + c4;
  double v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] = src[eix(r, c4 + q)];
    if (zero_upper_of_diag && ib == jb && c4 + q > r) v[q] = 0.0;
  }
  if (through) {
#pragma unroll
    for (int q = 0; q < 4; ++q) __hip_atomic_store(&out[q], v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = v[q];
}

// Schedule (NT = NB/16 sub-panels; wave 0 carries the critical path):
//   leaf(0) | for jb: phase A (sub-panel solve, all waves) | { wave 0: update tile (jb+1,jb+1),
//   leaf(jb+1) }  ||  { other waves: the rest of the rank-16 update, then row jb of W }
// so the in-block trailing update and the whole inversion hide behind the leaf chain.
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

#ifndef POTRF_WAVES_128
#define POTRF_WAVES_128 8   // waves of the 128-wide kernel.  12 / 16 waves spread the shadow work thinner but spill (168 / 128-register budgets) and crowd the leaf wave's SIMD: 1.77 -> 1.94 / 2.13 ms per N = 4096 step
#endif
template <int NB, int NWAVES>
__device__ __forceinline__ void potrf_block(const double* A, int64_t lda, double* __restrict__ L, int64_t ldl,
                                            double* __restrict__ W, int* info, int col0, const unsigned* wait_flag,
                                            unsigned wait_count, unsigned* done_flag, unsigned long long wait_ticks, double* sm) {
  constexpr int NT = NB / 16, NTRI = NT * (NT + 1) / 2;
  double* Lt = sm;
  double* Wt = sm + NTRI * 256;
  double* sh = sm + 2 * NTRI * 256;   // leaf hand-off area

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  // Resident start (split-panel schedule): the kernel is launched a whole step early, so that it already owns a
  // CU (it needs all of its LDS, and behind a saturating bulk update a freshly launched one waits 20-30 us for a
  // CU to drain), and sleeps here until the `wait_count` workgroups of the preceding A_kk -= S S^T kernel have
  // published their tiles (write-through stores + an agent-scope counter, gemm.hip).  One lane polls, relaxed;
  // ONE agent-scope acquire after the match drops this CU's stale lines; then the barrier releases the loads.
  // The spin is bounded (wait_ticks of s_memrealtime, 0.25 s by default): on a timeout the block reports info = -7 and
  // carries on (the host discards the results and repeats the call with kernel-boundary hand-overs).
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
  for (int t = tid; t < LEAF_SHARE_DOUBLES; t += NWAVES * 64) sh[t] = 0.0;   // epochs start at 0

  // ---- load the lower tiles of A_kk.  Wave 0 fetches only the first diagonal tile (all the first
  // leaf needs), waves 2.. fetch the rest; every global load is issued BEFORE the barrier and the bulk
  // of them lands while the first leaf is already running (plain loads stay in flight across
  // s_barrier).  Wave 1 (the inverse half of the leaf) loads nothing.
  constexpr int LW = NWAVES - 2;                         // loader waves
  constexpr int TPL = (NTRI - 1 + LW - 1) / LW;          // tiles per loader wave
  {
    const int r = lane >> 2, c4 = (lane & 3) * 4;
    double v[TPL][4];
    if (wave == 0) {
      const double* src = A + (int64_t)r * lda + c4;
#pragma unroll
      for (int q = 0; q < 4; ++q) v[0][q] = src[q];
#pragma unroll
      for (int q = 0; q < 4; ++q) Lt[eix(r, c4 + q)] = v[0][q];
    } else if (wave >= 2) {
#pragma unroll
      for (int u = 0; u < TPL; ++u) {
        const int t = 1 + (wave - 2) + u * LW;
        if (t < NTRI) {
          int ib = 0, jb = t;
          while (jb > ib) { jb -= ib + 1; ++ib; }
          const double* src = A + (int64_t)(ib * 16 + r) * lda + jb * 16 + c4;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[u][q] = src[q];
        }
      }
    }
    __syncthreads();   // hand-off area zeroed, tile (0,0) in LDS
    GPX_STAMP(1);
    if (wave == 0) { leaf_potrf(Lt + tix(0, 0) * 256, sh, lane, 0, info, col0); GPX_STAMP(40); }
    else if (wave == 1) { leaf_winv(Wt + tix(0, 0) * 256, sh, lane, 0); GPX_STAMP_T(41, 64); }
    else {
#pragma unroll
      for (int u = 0; u < TPL; ++u) {
        const int t = 1 + (wave - 2) + u * LW;
        if (t < NTRI) {
#pragma unroll
          for (int q = 0; q < 4; ++q) Lt[t * 256 + eix(r, c4 + q)] = v[u][q];
        }
      }
      // zeros above the diagonal of the L block and of W (nothing depends on them)
      for (int t = wave - 2; t < NT * NT; t += LW) {
        const int ib = t / NT, jb = t % NT;
        if (jb > ib) {
          double* ldst = L + (int64_t)(ib * 16 + r) * ldl + jb * 16 + c4;
          double* wdst = W + (int64_t)(ib * 16 + r) * NB + jb * 16 + c4;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ldst[q] = 0.0;
            wdst[q] = 0.0;
          }
        }
      }
    }
  }
  __syncthreads();

#pragma unroll 1
  for (int jb = 0; jb < NT; ++jb) {
    GPX_STAMP(2 + 2 * jb);
    // ---- phase A: sub-panel solve, tiles (ib, jb), ib > jb; wave 0 takes (jb+1, jb)
    {
      const double* Wd = Wt + tix(jb, jb) * 256;
      for (int ib = jb + 1 + wave; ib < NT; ib += NWAVES) tile_trsm(Lt + tix(ib, jb) * 256, Wd, fr, fk);
    }
    __syncthreads();
    GPX_STAMP(3 + 2 * jb);
    // ---- shadow slot: critical path on wave 0, everything else on the other waves
    if (jb + 1 < NT) {
      if (wave == 0) {
        double* Cp = Lt + tix(jb + 1, jb + 1) * 256;
        const double* Ap = Lt + tix(jb + 1, jb) * 256;
        tile_syrk(Cp, Ap, Ap, fr, fk);
        leaf_potrf(Cp, sh, lane, (jb + 1) * 16, info, col0 + (jb + 1) * 16);
      } else if (wave == 1) {
        leaf_winv(Wt + tix(jb + 1, jb + 1) * 256, sh, lane, (jb + 1) * 16);
      } else {
        const int tn = NT - 1 - jb;
        const int ntiles = tn * (tn + 1) / 2;
        // rank-16 update tiles q = 1 .. ntiles-1 (q = 0 is wave 0's), two per wave and iteration
        auto tile_of = [&](int q, double*& Cp, const double*& Ap, const double*& Bp) {
          int ib2 = 0, q2 = q;
          while (q2 > ib2) { q2 -= ib2 + 1; ++ib2; }
          const int ib = jb + 1 + ib2, lb = jb + 1 + q2;
          Cp = Lt + tix(ib, lb) * 256; Ap = Lt + tix(ib, jb) * 256; Bp = Lt + tix(lb, jb) * 256;
        };
        constexpr int SW = NWAVES - 2;   // waves sharing the shadow work
        for (int q = 1 + 2 * (wave - 2); q < ntiles; q += 2 * SW) {
          double *C0, *C1; const double *A0, *B0, *A1, *B1;
          tile_of(q, C0, A0, B0);
          if (q + 1 < ntiles) {
            tile_of(q + 1, C1, A1, B1);
            tile_syrk2(C0, A0, B0, C1, A1, B1, fr, fk);
          } else {
            tile_syrk(C0, A0, B0, fr, fk);
          }
        }
        for (int cb = wave - 2; cb < jb; cb += SW) tile_winv(Lt, Wt, jb, cb, fr, fk);   // row jb of W: off the critical path
        // write out what is final: column jb of L (after phase A) and row jb-1 of W (previous slot)
        for (int q = wave - 2; q < NT; q += NWAVES - 2) {
          if (q < NT - jb) store_tile(Lt, L, ldl, jb + q, jb, lane, true);
          else if (jb >= 1) store_tile(Wt, W, NB, jb - 1, q - (NT - jb), lane, false, GPX_W_THROUGH && done_flag != nullptr);
        }
      }
    } else {
      // last row of W: nothing left to hide behind, all waves share it
      for (int cb = wave; cb < jb; cb += NWAVES) tile_winv(Lt, Wt, jb, cb, fr, fk);
    }
    __syncthreads();
  }

  GPX_STAMP(2 + 2 * NT);
  // ---- write out the remainder: the last L tile and the last two rows of W (everything else went
  // out in the shadow slots)
  {
    for (int q = wave; q < 2 * NT; q += NWAVES) {
      if (q == 0) store_tile(Lt, L, ldl, NT - 1, NT - 1, lane, true);
      else if (q <= NT) store_tile(Wt, W, NB, NT - 1, q - 1, lane, false, GPX_W_THROUGH && done_flag != nullptr);
      else if (NT >= 2) store_tile(Wt, W, NB, NT - 2, q - NT - 1, lane, false, GPX_W_THROUGH && done_flag != nullptr);
    }
  }
  GPX_STAMP(3 + 2 * NT);
  POTRF_TL(2, col0);
  // publish W_k to the already-resident S = A W_k^T kernel: every wave drains its (write-through) stores, the
  // barrier collects the waves, one lane raises the flag
  if (done_flag) {
#if GPX_W_THROUGH
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // plain stores above: write this XCD's dirty lines back
#endif
    __syncthreads();
    if (tid == 0) __hip_atomic_store(done_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// =====================================================================================================
// Dataflow form of the same factorisation (GPX_POTRF_FLOW, the default): no workgroup barrier inside the
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
#ifndef GPX_POTRF_FLOW
#define GPX_POTRF_FLOW 1
#endif
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

// wave w's leaf: L_jj.  As leaf_potrf, but the tile comes from / goes to the stage's diagonal area: rows are read from
// the symmetric column-major tile, column j is published in place the moment it exists (l_ij by lane i, 1/l_jj, and
// the epoch at the end of a reader group).
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
  if (bad && lane == 0 && *info == 0) *info = gcol0 + __ffsll((long long)bad);
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

template <int NB>
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
  // 1/l_jj, zero slots, epochs and flags start at zero (the tiles themselves are written by their owners)
  for (int t = tid; t < NT * (DT_STRIDE - DT_RINV); t += NW * 64)
    DT[(t / (DT_STRIDE - DT_RINV)) * DT_STRIDE + DT_RINV + t % (DT_STRIDE - DT_RINV)] = 0.0;
  for (int t = tid; t < (NLOW + NT + 1) / 2 + 1; t += NW * 64) reinterpret_cast<double*>(fin)[t] = 0.0;

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
        GPX_STAMP_T(320 + w * 8 + s, 64 * w);
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
        store_tile(Wt, W, NB, 0, 0, lane, false);
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
      for (int q = 0; q < 4; ++q) { ldst[q] = 0.0; wdst[q] = 0.0; }
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
    store_tile(Wt, W, NB, jb, w, lane, false);
    if (w == jb - 1) store_tile(Wt, W, NB, jb, jb, lane, false);
  }

  GPX_STAMP_T(2 + 2 * NT, 64 * (NT - 1));
  POTRF_TL(2, col0);
  // publish W_k / L_kk to an already-resident consumer: every wave releases its plain stores, the barrier collects
  // the waves, one lane raises the flag
  if (done_flag) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(done_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  GPX_STAMP_T(3 + 2 * NT, 64 * (NT - 1));
}

struct PotrfArgs {
  const double* A; int64_t lda; double* L; int64_t ldl; double* W; int* info; int col0;
  const unsigned* wait_flag; unsigned wait_count; unsigned* done_flag; unsigned long long wait_ticks;
  int64_t sA, sL, sW, s_info, s_flag;   // per-draw strides of a batched launch (blockIdx.x = draw)
};

template <int NB, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void potrf_diag_kernel(PotrfArgs p) {
  constexpr int NTRI = (NB / 16) * (NB / 16 + 1) / 2;
  const int64_t z = blockIdx.x;
#if GPX_POTRF_FLOW
  static_assert(NWAVES == NB / 16, "dataflow potrf_diag: one wave per block row");
  __shared__ double sm[(NB / 16) * (NB / 16) * 256 + (NB / 16) * DT_STRIDE + (NTRI + NB / 16) / 2 + 2];
  potrf_flow<NB>(p.A + z * p.sA, p.lda, p.L + z * p.sL, p.ldl, p.W + z * p.sW, p.info + z * p.s_info, p.col0,
                 p.wait_flag ? p.wait_flag + z * p.s_flag : nullptr, p.wait_count,
                 p.done_flag ? p.done_flag + z * p.s_flag : nullptr, p.wait_ticks, sm);
#else
  __shared__ double sm[2 * NTRI * 256 + LEAF_SHARE_DOUBLES];
  potrf_block<NB, NWAVES>(p.A + z * p.sA, p.lda, p.L + z * p.sL, p.ldl, p.W + z * p.sW, p.info + z * p.s_info, p.col0,
                          p.wait_flag ? p.wait_flag + z * p.s_flag : nullptr, p.wait_count,
                          p.done_flag ? p.done_flag + z * p.s_flag : nullptr, p.wait_ticks, sm);
#endif
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
      hipExtLaunchKernelGGL((potrf_diag_kernel<64, 4>), grid, dim3(256), 0, s, nullptr, q.stop_event, 0, p);
      break;
    case 128:
      hipExtLaunchKernelGGL((potrf_diag_kernel<128, POTRF_WAVES_128>), grid, dim3(64 * POTRF_WAVES_128), 0, s, nullptr, q.stop_event, 0, p);
      break;
    default:
      gpx_set_error(h, "potrf_diag: nb must be 64 or 128");
      return GPX_ERR_ARG;
  }
  GPX_HIP_CHECK(h, hipGetLastError());
  return GPX_OK;
}
