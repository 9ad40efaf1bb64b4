// Diagnostic: external event-record nodes inside a stream capture (ROCm 7.2 behaviour probe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e)); } } while (0)
__global__ void spin(int* p, int v, long long cycles) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  *p = v;
}
__global__ void copy1(const int* s, int* d) { *d = *s; }
int run(int variant) {
  printf("variant %d\n", variant);
  hipStream_t o, m, side; hipEvent_t fork, join, mark;
  CK(hipStreamCreateWithFlags(&o, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&mark, hipEventDisableTiming));
  int *a, *b, *shadow; CK(hipMalloc(&a, 4)); CK(hipMalloc(&b, 4)); CK(hipMalloc(&shadow, 4));
  CK(hipMemset(a, 0, 4)); CK(hipMemset(b, 0, 4)); CK(hipMemset(shadow, 0, 4)); CK(hipDeviceSynchronize());
  hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
  CK(hipStreamBeginCapture(o, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(spin, 1, 1, 0, o, a, 7, 100000LL);  // ~1 ms at 100 MHz wall clock
  CK(hipEventRecord(fork, o));
  CK(hipStreamWaitEvent(m, fork, 0));
  CK(hipEventRecordWithFlags(mark, m, hipEventRecordExternal));
  if (variant == 1) hipLaunchKernelGGL(copy1, 1, 1, 0, m, a, b);  // a real node after the external record
  hipLaunchKernelGGL(spin, 1, 1, 0, o, b, 9, 100000LL);  // origin goes on
  CK(hipEventRecord(join, m));
  CK(hipStreamWaitEvent(o, join, 0));
  if (variant == 2) hipLaunchKernelGGL(copy1, 1, 1, 0, o, a, b);  // a real node after the join
  hipError_t e = hipStreamEndCapture(o, &g);
  printf("  end capture: %s\n", hipGetErrorString(e));
  if (e != hipSuccess) { (void)hipGetLastError(); return 1; }
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemsetAsync(a, 0, 4, o)); CK(hipStreamSynchronize(o));
    CK(hipGraphLaunch(ge, o));
    hipError_t q = hipEventQuery(mark);
    printf("  rep %d: query right after launch: %s\n", rep, hipGetErrorString(q)); (void)hipGetLastError();
    CK(hipStreamWaitEvent(side, mark, 0));
    hipLaunchKernelGGL(copy1, 1, 1, 0, side, a, shadow);
    CK(hipStreamSynchronize(side));
    int hs = -1; CK(hipMemcpy(&hs, shadow, 4, hipMemcpyDeviceToHost));
    printf("  rep %d: side stream saw a = %d (7 = waited for the first kernel, 0 = did not wait)\n", rep, hs);
    CK(hipStreamSynchronize(o));
  }
  return 0;
}
// the library's pattern: side-stream kernels ("fast-forwards") forked off the origin, marks on a third stream that
// waits for the origin's position and for those side kernels, everything joined at the end
int run_lib(int n_marks, bool join_marks_first) {
  printf("library pattern, %d marks, join marks %s\n", n_marks, join_marks_first ? "first" : "last");
  hipStream_t o, m, ff[4]; hipEvent_t fork, join, ev_step[4], ev_ff[4], mark[8];
  CK(hipStreamCreateWithFlags(&o, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking));
  for (int k = 0; k < 4; ++k) { CK(hipStreamCreateWithFlags(&ff[k], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ev_step[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev_ff[k], hipEventDisableTiming)); }
  for (int k = 0; k < 8; ++k) CK(hipEventCreateWithFlags(&mark[k], hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  int* a; CK(hipMalloc(&a, 64)); CK(hipMemset(a, 0, 64)); CK(hipDeviceSynchronize());
  hipGraph_t g = nullptr;
  CK(hipStreamBeginCapture(o, hipStreamCaptureModeThreadLocal));
  int busy[4] = {0, 0, 0, 0};
  for (int c = 0; c < n_marks; ++c) {
    hipLaunchKernelGGL(spin, 1, 1, 0, o, a, c, 1000LL);
    CK(hipEventRecord(ev_step[c % 4], o)); CK(hipStreamWaitEvent(ff[c % 4], ev_step[c % 4], 0));
    hipLaunchKernelGGL(spin, 1, 1, 0, ff[c % 4], a + 1 + c % 4, c, 20000LL);
    CK(hipEventRecord(ev_ff[c % 4], ff[c % 4])); busy[c % 4] = 1;
    hipLaunchKernelGGL(spin, 1, 1, 0, o, a, c, 1000LL);
    CK(hipEventRecord(fork, o)); CK(hipStreamWaitEvent(m, fork, 0));
    for (int k = 0; k < 4; ++k) if (busy[k]) CK(hipStreamWaitEvent(m, ev_ff[k], 0));
    CK(hipEventRecordWithFlags(mark[c], m, hipEventRecordExternal));
  }
  if (join_marks_first) { CK(hipEventRecord(join, m)); CK(hipStreamWaitEvent(o, join, 0)); }
  for (int k = 0; k < 4; ++k) if (busy[k]) CK(hipStreamWaitEvent(o, ev_ff[k], 0));
  if (!join_marks_first) { CK(hipEventRecord(join, m)); CK(hipStreamWaitEvent(o, join, 0)); }
  hipError_t e = hipStreamEndCapture(o, &g);
  printf("  end capture: %s\n", hipGetErrorString(e));
  (void)hipGetLastError();
  return 0;
}
int main() { for (int v = 0; v < 3; ++v) run(v); run_lib(1, false); run_lib(2, false); run_lib(4, false); run_lib(4, true); return 0; }
