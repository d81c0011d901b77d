// Probe (diagnostic, not shipped): issue cost / latency of the fp64 instructions the 16x16 leaf is made of,
// one wave alone on a SIMD, timed with s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void probe(double* io, unsigned long long* clk) {
  double a[16];
  for (int i = 0; i < 16; ++i) a[i] = io[threadIdx.x + 64 * i];
  double s = io[threadIdx.x] * 0.5 + 1.0, m = 1.0000001;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < 64; ++it) {
    if (MODE == 0) {  // 16 independent v_fma_f64
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(s), "v"(m));
    } else if (MODE == 1) {  // 16 independent v_fmac_f64_dpp with s_nop 1
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(s), "v"(m));
    } else if (MODE == 2) {  // same without the nop (source not rewritten: no hazard)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(s), "v"(m));
    } else if (MODE == 3) {  // v_mov_b64_dpp + v_fma_f64 pairs
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double t;
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(t) : "v"(s));
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(t), "v"(m));
      }
    } else if (MODE == 4) {  // dependent chain of 16 v_fma_f64 (latency)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(s));
    } else if (MODE == 5) {  // dependent chain: v_rsq_f64 -> v_mul_f64 (x8)
#pragma unroll
      for (int i = 0; i < 8; ++i) { asm volatile("v_rsq_f64 %0, %0" : "+v"(a[0])); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[0]) : "v"(m)); }
    } else if (MODE == 6) {  // dependent chain: v_mov_b64_dpp -> v_fma_f64 (x8)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[0]));
        asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(s));
      }
    } else if (MODE == 7) {  // 16 independent (2x v_mov_b32_dpp via builtin) + fma
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int lo = __double2loint(s), hi = __double2hiint(s);
        lo = __builtin_amdgcn_update_dpp(0, lo, 0x153, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(0, hi, 0x153, 0xf, 0xf, false);
        a[i] = fma(__hiloint2double(hi, lo), m, a[i]);
        asm volatile("" : "+v"(a[i]), "+v"(s));
      }
    } else if (MODE == 9) {  // 16 independent v_fma_f64 with only 16 lanes active (EXEC = 0xffff): are empty passes skipped?
      asm volatile("s_mov_b64 exec, 0xffff" ::: "exec");
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(s), "v"(m));
      asm volatile("s_mov_b64 exec, -1" ::: "exec");
    } else if (MODE == 10) {  // 16 independent v_fmac_f64_dpp with EXEC = 0xffff
      asm volatile("s_mov_b64 exec, 0xffff" ::: "exec");
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(s), "v"(m));
      asm volatile("s_mov_b64 exec, -1" ::: "exec");
    } else if (MODE == 11) {  // dependent chain of 16 v_fma_f64 with EXEC = 0xffff (latency)
      asm volatile("s_mov_b64 exec, 0xffff" ::: "exec");
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(s));
      asm volatile("s_mov_b64 exec, -1" ::: "exec");
    } else if (MODE == 8) {  // dependent chain of 16 v_fmac_f64_dpp on one accumulator
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[0]) : "v"(s), "v"(m));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  double r = 0; for (int i = 0; i < 16; ++i) r += a[i];
  io[threadIdx.x] = r;
  if (threadIdx.x == 0) clk[MODE] = t1 - t0;
}
int main() {
  double* io; unsigned long long* clk; (void)hipMalloc(&io, 64 * 17 * 8); (void)hipMalloc(&clk, 16 * 8);
  (void)hipMemset(io, 0, 64 * 17 * 8);
#define RUN(M) hipLaunchKernelGGL(probe<M>, dim3(1), dim3(64), 0, 0, io, clk); hipLaunchKernelGGL(probe<M>, dim3(1), dim3(64), 0, 0, io, clk);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
  (void)hipDeviceSynchronize();
  unsigned long long h[16]; (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"v_fma_f64 indep (per instr)", "s_nop1+v_fmac_f64_dpp indep", "v_fmac_f64_dpp indep", "v_mov_b64_dpp+v_fma_f64 pair",
                         "v_fma_f64 dependent (latency)", "v_rsq_f64+v_mul_f64 dependent pair", "nop+v_mov_b64_dpp+v_fma_f64 dependent pair",
                         "2x v_mov_b32_dpp + v_fma_f64", "v_fmac_f64_dpp dependent", "v_fma_f64 indep, EXEC=0xffff",
                         "v_fmac_f64_dpp indep, EXEC=0xffff", "v_fma_f64 dependent, EXEC=0xffff"};
  const int per[] = {16, 16, 16, 16, 16, 8, 8, 16, 16, 16, 16, 16};
  for (int m = 0; m < 12; ++m) printf("%-45s %.1f cycles\n", names[m], (double)h[m] / 64.0 / per[m]);
  return 0;
}
