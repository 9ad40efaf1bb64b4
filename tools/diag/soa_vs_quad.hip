// Microbenchmark behind the state layout (run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/svq soa_vs_quad.hip && /tmp/svq):
// what a step-kernel-shaped launch can move -- every load issued first, a little arithmetic, then the stores -- when an env's
// state words are (A) one dword per lane in W separate SoA rows, (B) float4 per lane in W/4 rows of 16-byte quads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int R, int W>
__global__ void __launch_bounds__(128) rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = src[(size_t)r * n + i];
  float acc = 0.0f;
#pragma unroll
  for (int r = 0; r < R; ++r) acc = fmaf(v[r], 1.0001f, acc);
#pragma unroll
  for (int r = 0; r < W; ++r) dst[(size_t)r * n + i] = v[r] + acc;
}

template <int RQ, int WQ>
__global__ void __launch_bounds__(128) quads_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 v[RQ];
#pragma unroll
  for (int r = 0; r < RQ; ++r) v[r] = src[(size_t)r * n + i];
  float acc = 0.0f;
#pragma unroll
  for (int r = 0; r < RQ; ++r) acc = fmaf(v[r].x + v[r].y + v[r].z + v[r].w, 1.0001f, acc);
#pragma unroll
  for (int r = 0; r < WQ; ++r) dst[(size_t)r * n + i] = make_float4(v[r].x + acc, v[r].y, v[r].z, v[r].w);
}

template <typename F> float time_it(F launch, int reps) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int k = 0; k < 3; ++k) launch(k);
  CHECK(hipEventRecord(a));
  for (int k = 0; k < reps; ++k) launch(k);
  CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  for (int n : {1 << 20, 1 << 22}) {
    float *p, *q;
    CHECK(hipMalloc(&p, (size_t)32 * n * 4)); CHECK(hipMalloc(&q, (size_t)32 * n * 4));
    CHECK(hipMemset(p, 0, (size_t)32 * n * 4)); CHECK(hipMemset(q, 0, (size_t)32 * n * 4));
    dim3 g((n + 127) / 128), b(128);
    // in place, like the step kernels (state read and written at the same addresses)
    float t1 = time_it([&](int) { hipLaunchKernelGGL((rows_kernel<30, 23>), g, b, 0, 0, p, p, n); }, 50);
    float t2 = time_it([&](int) { hipLaunchKernelGGL((quads_kernel<8, 6>), g, b, 0, 0, (const float4*)p, (float4*)p, n); }, 50);
    float t3 = time_it([&](int k) { if (k & 1) hipLaunchKernelGGL((rows_kernel<30, 23>), g, b, 0, 0, p, q, n); else hipLaunchKernelGGL((rows_kernel<30, 23>), g, b, 0, 0, q, p, n); }, 50);
    float t4 = time_it([&](int k) { if (k & 1) hipLaunchKernelGGL((quads_kernel<8, 6>), g, b, 0, 0, (const float4*)p, (float4*)q, n); else hipLaunchKernelGGL((quads_kernel<8, 6>), g, b, 0, 0, (const float4*)q, (float4*)p, n); }, 50);
    printf("n = %d envs\n", n);
    printf("  dword rows, 30 read + 23 written, in place : %7.1f us  %.2f TB/s\n", t1 * 1e3, (double)(30 + 23) * 4 * n / (t1 * 1e-3) / 1e12);
    printf("  float4 quads, 8 read + 6 written, in place : %7.1f us  %.2f TB/s\n", t2 * 1e3, (double)(32 + 24) * 4 * n / (t2 * 1e-3) / 1e12);
    printf("  dword rows, ping-pong between two buffers  : %7.1f us  %.2f TB/s\n", t3 * 1e3, (double)(30 + 23) * 4 * n / (t3 * 1e-3) / 1e12);
    printf("  float4 quads, ping-pong                    : %7.1f us  %.2f TB/s\n", t4 * 1e3, (double)(32 + 24) * 4 * n / (t4 * 1e-3) / 1e12);
    CHECK(hipFree(p)); CHECK(hipFree(q));
  }
  return 0;
}
