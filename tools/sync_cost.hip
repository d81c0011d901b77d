// What a stream-ordering operation between two dependent kernels costs on the stream it sits in (gfx950):
// kernel A (one workgroup, ~20 us) ; <op> ; kernel B -- gap = B's first s_memrealtime - A's last one.
#include <hip/hip_runtime.h>
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
  return 0;
}
