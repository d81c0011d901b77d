// Host-side cost of the enqueue calls the schedule is made of (diagnostic, not shipped): empty kernels, four streams.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Args { double* p[12]; long long q[12]; };
__global__ void empty_kernel(Args a) { if (a.q[0] == 12345 && a.p[0]) a.p[0][0] = 1.0; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int lo, hi; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  hipStream_t s[4];
  for (int i = 0; i < 4; ++i) (void)hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, i == 3 ? lo : hi);
  std::vector<hipEvent_t> ev(4096);
  for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence);
  Args a{}; const int N = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    double t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[i & 3], a);
    double t1 = now(); (void)hipDeviceSynchronize();
    for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[i & 3], nullptr, nullptr, 0, a);
    double t2 = now(); (void)hipDeviceSynchronize();
    for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s[i & 3], nullptr, ev[i], 0, a);
    double t3 = now(); (void)hipDeviceSynchronize();
    for (int i = 0; i < N; ++i) { (void)hipEventRecord(ev[i], s[i & 3]); }
    double t4 = now(); (void)hipDeviceSynchronize();
    for (int i = 0; i < N; ++i) { (void)hipStreamWaitEvent(s[(i + 1) & 3], ev[i], 0); }
    double t5 = now(); (void)hipDeviceSynchronize();
    // the pattern of one block step: 6 launches on 4 streams, 4 cross-stream waits on attached events
    double t6 = now();
    for (int k = 0; k < N / 8; ++k) {
      hipEvent_t* e = &ev[(k * 8) & 4095];
      hipExtLaunchKernelGGL(empty_kernel, dim3(1), dim3(512), 0, s[0], nullptr, e[0], 0, a);
      (void)hipStreamWaitEvent(s[1], e[0], 0);
      hipExtLaunchKernelGGL(empty_kernel, dim3(16), dim3(256), 0, s[1], nullptr, nullptr, 0, a);
      (void)hipStreamWaitEvent(s[2], e[0], 0);
      hipExtLaunchKernelGGL(empty_kernel, dim3(100), dim3(256), 0, s[2], nullptr, e[1], 0, a);
      hipExtLaunchKernelGGL(empty_kernel, dim3(100), dim3(256), 0, s[2], nullptr, e[2], 0, a);
      hipExtLaunchKernelGGL(empty_kernel, dim3(100), dim3(256), 0, s[2], nullptr, nullptr, 0, a);
      (void)hipStreamWaitEvent(s[3], e[1], 0);
      hipExtLaunchKernelGGL(empty_kernel, dim3(1000), dim3(256), 0, s[3], nullptr, e[3], 0, a);
      (void)hipStreamWaitEvent(s[1], e[2], 0);
    }
    double t7 = now(); (void)hipDeviceSynchronize();
    double t8 = now();
    if (rep == 1)
      printf("per call: hipLaunchKernelGGL %.2f us | hipExtLaunchKernelGGL %.2f | with stop event %.2f | hipEventRecord %.2f | hipStreamWaitEvent %.2f\n"
             "one block step's pattern (6 launches, 4 waits): %.1f us enqueue; drained %.1f us after the last enqueue\n",
             (t1 - t0) / N, (t2 - t1) / N, (t3 - t2) / N, (t4 - t3) / N, (t5 - t4) / N, (t7 - t6) / (N / 8), t8 - t7);
  }
  return 0;
}
