// What a stream-ordering operation between two dependent kernels costs on the stream it sits in (gfx950):
// kernel A (one workgroup, ~20 us) ; <op> ; kernel B -- gap = B's first s_memrealtime - A's last one.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void kern(unsigned long long* t, int slot, int spin_us) {
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    t[2 * slot] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) { }
    t[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
  }
}
__global__ void multi(unsigned long long* t, int slot, int spin_us) {   // 64 workgroups
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) t[2 * slot] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) { }
  if (threadIdx.x == 0 && blockIdx.x == 0) t[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
}
// like potrf_diag's tail: one 512-thread workgroup that ends by storing `kb` KB, then stamps its last instruction
__global__ __launch_bounds__(512) void tail_store(unsigned long long* t, int slot, int spin_us, double* dst, int kb) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) t[2 * slot] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) { }
  for (int i = threadIdx.x; i < kb * 128; i += 512) dst[i] = (double)i;
  if (threadIdx.x == 0) t[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
}
// background load on another stream: `nwg` workgroups streaming read-modify-write over a big buffer for ~`us` us
__global__ __launch_bounds__(256) void background(double* buf, size_t n, int us, int dirty) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) {
    if (dirty) buf[i] = buf[i] + 1.0; else acc += buf[i];
    i += (size_t)gridDim.x * 256; if (i >= n) i -= n;
  }
  if (acc == 12345.678) buf[0] = acc;
}
int main() {
  hipStream_t P, Q; int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&P, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&Q, hipStreamNonBlocking, hi));
  unsigned long long* t; CK(hipMalloc(&t, 64 * 8)); CK(hipMemset(t, 0, 64 * 8));
  hipEvent_t evs[4]; unsigned flags[4] = {hipEventDefault, hipEventDisableTiming, hipEventDisableTiming | hipEventDisableSystemFence, hipEventDisableTiming | hipEventReleaseToDevice};
  for (int i = 0; i < 4; ++i) CK(hipEventCreateWithFlags(&evs[i], flags[i]));
  hipEvent_t qe; CK(hipEventCreateWithFlags(&qe, hipEventDisableTiming | hipEventDisableSystemFence));
  uint64_t* sig = nullptr;
  bool have_sig = hipExtMallocWithFlags(reinterpret_cast<void**>(&sig), 64, hipMallocSignalMemory) == hipSuccess;
  if (have_sig) CK(hipMemset(sig, 0, 64));
  const char* names[] = {"nothing", "record (default event)", "record (DisableTiming)", "record (DisableTiming|DisableSystemFence)", "record (DisableTiming|ReleaseToDevice)",
                         "wait on a completed event of another stream", "record + wait", "hipStreamWriteValue64", "hipStreamWaitValue64 (already satisfied)",
                         "multi-WG A -> one-WG B, nothing", "one-WG A -> multi-WG B, nothing", "multi-WG A, record(nofence), B"};
  for (int op = 0; op < 12; ++op) {
    if ((op == 7 || op == 8) && !have_sig) { printf("%-50s n/a (no signal memory)\n", names[op]); continue; }
    std::vector<double> gaps;
    for (int rep = 0; rep < 12; ++rep) {
      // something completed on Q for the wait cases
      hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, Q, t, 8, 1); CK(hipEventRecord(qe, Q)); CK(hipStreamSynchronize(Q));
      if (op == 9 || op == 11) hipLaunchKernelGGL(multi, dim3(64), dim3(256), 0, P, t, 0, 20);
      else hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, P, t, 0, 20);
      switch (op) {
        case 1: case 2: case 3: case 4: CK(hipEventRecord(evs[op - 1], P)); break;
        case 5: CK(hipStreamWaitEvent(P, qe, 0)); break;
        case 6: CK(hipEventRecord(evs[2], P)); CK(hipStreamWaitEvent(P, qe, 0)); break;
        case 7: CK(hipStreamWriteValue64(P, sig, (uint64_t)(rep + 1), 0)); break;
        case 8: CK(hipStreamWaitValue64(P, sig, 0, hipStreamWaitValueGte, 0xffffffffffffffffull)); break;
        case 11: CK(hipEventRecord(evs[2], P)); break;
        default: break;
      }
      if (op == 10) hipLaunchKernelGGL(multi, dim3(64), dim3(256), 0, P, t, 1, 5);
      else hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, P, t, 1, 5);
      CK(hipStreamSynchronize(P));
      unsigned long long h[4]; CK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
      if (rep >= 2) gaps.push_back((double)(h[2] - h[1]) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("%-50s gap median %.2f us (min %.2f, max %.2f)\n", names[op], gaps[gaps.size() / 2], gaps.front(), gaps.back());
  }
  double* dst; CK(hipMalloc(&dst, 1 << 20));
  {   // stopEvent attached to the kernel itself (hipExtLaunchKernelGGL) instead of a separate hipEventRecord
    hipEvent_t se; CK(hipEventCreateWithFlags(&se, hipEventDisableTiming | hipEventDisableSystemFence));
    for (int variant = 0; variant < 2; ++variant) {
      std::vector<double> gaps, qlag;
      for (int rep = 0; rep < 12; ++rep) {
        if (variant == 0) {
          hipLaunchKernelGGL(tail_store, dim3(1), dim3(512), 0, P, t, 0, 20, dst, 34);
          CK(hipEventRecord(se, P));
        } else {
          hipExtLaunchKernelGGL(tail_store, dim3(1), dim3(512), 0, P, nullptr, se, 0, t, 0, 20, dst, 34);
        }
        CK(hipStreamWaitEvent(Q, se, 0));
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, Q, t, 2, 1);          // consumer on the other stream
        hipLaunchKernelGGL(multi, dim3(16), dim3(256), 0, P, t, 1, 5);       // next kernel on the same stream
        CK(hipStreamSynchronize(P)); CK(hipStreamSynchronize(Q));
        unsigned long long h[6]; CK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
        if (rep >= 2) { gaps.push_back((double)(h[2] - h[1]) / 100.0); qlag.push_back(((double)h[4] - (double)h[1]) / 100.0); }
      }
      std::sort(gaps.begin(), gaps.end()); std::sort(qlag.begin(), qlag.end());
      printf("%s: same-stream gap median %.2f us; other-stream consumer starts %.2f us after the producer's end (min %.2f)\n",
             variant ? "stopEvent on the kernel (hipExtLaunchKernelGGL)" : "kernel + hipEventRecord                        ",
             gaps[gaps.size() / 2], qlag[qlag.size() / 2], qlag.front());
    }
  }
  {
    hipStream_t T; CK(hipStreamCreateWithPriority(&T, hipStreamNonBlocking, lo));
    const size_t nbig = (size_t)1 << 26;   // 512 MB
    double* big; CK(hipMalloc(&big, nbig * 8)); CK(hipMemset(big, 0, nbig * 8));
    for (int mode = 0; mode < 4; ++mode) {   // 0 none, 1 read-only background, 2 read-modify-write background, 3 rmw, few WGs
      std::vector<double> gaps;
      for (int rep = 0; rep < 12; ++rep) {
        if (mode == 1) hipLaunchKernelGGL(background, dim3(1024), dim3(256), 0, T, big, nbig, 200, 0);
        if (mode == 2) hipLaunchKernelGGL(background, dim3(1024), dim3(256), 0, T, big, nbig, 200, 1);
        if (mode == 3) hipLaunchKernelGGL(background, dim3(128), dim3(256), 0, T, big, nbig, 200, 1);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, P, t, 2, 40);   // let the background get going
        hipLaunchKernelGGL(tail_store, dim3(1), dim3(512), 0, P, t, 0, 20, dst, 34);
        CK(hipEventRecord(evs[2], P));
        hipLaunchKernelGGL(multi, dim3(16), dim3(256), 0, P, t, 1, 5);
        CK(hipStreamSynchronize(P)); CK(hipStreamSynchronize(T));
        unsigned long long h[4]; CK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
        if (rep >= 2) gaps.push_back((double)(h[2] - h[1]) / 100.0);
      }
      std::sort(gaps.begin(), gaps.end());
      const char* bn[] = {"idle GPU", "1024-WG streaming READ kernel on another stream", "1024-WG streaming READ-MODIFY-WRITE kernel on another stream", "128-WG streaming READ-MODIFY-WRITE kernel on another stream"};
      printf("34 KB tail + event record -> 16-WG kernel, %s: gap median %.2f us (min %.2f, max %.2f)\n", bn[mode], gaps[gaps.size() / 2], gaps.front(), gaps.back());
    }
  }
  for (int kb : {0, 2, 34, 256}) {
    for (int with_record = 0; with_record < 2; ++with_record) {
      std::vector<double> gaps;
      for (int rep = 0; rep < 12; ++rep) {
        hipLaunchKernelGGL(tail_store, dim3(1), dim3(512), 0, P, t, 0, 20, dst, kb);
        if (with_record) CK(hipEventRecord(evs[2], P));
        hipLaunchKernelGGL(multi, dim3(16), dim3(256), 0, P, t, 1, 5);
        CK(hipStreamSynchronize(P));
        unsigned long long h[4]; CK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
        if (rep >= 2) gaps.push_back((double)(h[2] - h[1]) / 100.0);
      }
      std::sort(gaps.begin(), gaps.end());
      printf("one 512-thread WG ending with %3d KB of stores%s -> 16-WG kernel: gap median %.2f us (min %.2f, max %.2f)\n", kb,
             with_record ? ", event record," : "", gaps[gaps.size() / 2], gaps.front(), gaps.back());
    }
  }
  return 0;
}
