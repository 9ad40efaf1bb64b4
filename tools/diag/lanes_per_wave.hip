// Microbenchmark (run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/lpw tools/diag/lanes_per_wave.hip && /tmp/lpw):
// a chain of 1040 DEPENDENT step-kernel-shaped launches in one replayed hipGraph -- every launch reads R state rows of N envs, does a
// little arithmetic and writes W rows back in place of the next launch's input -- with the N envs spread over more or fewer waves:
// L active lanes per 64-lane wave (N / L one-wave workgroups). Question: at 4096 envs (64 full waves on a 256-CU chip), is the
// per-launch time bound by what ONE wave can pull through its CU, so that 256 quarter-filled waves on 256 CUs would be faster?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int R, int W>
__global__ void __launch_bounds__(64) step_like(const float* __restrict__ src, float* __restrict__ dst, int n, int lanes) {
  const int lane = threadIdx.x;
  const int i = blockIdx.x * lanes + lane;
  if (lane >= lanes || i >= n) return;
  float v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = src[(size_t)r * n + i];
  float acc = 0.0f;
#pragma unroll
  for (int r = 0; r < R; ++r) acc = fmaf(v[r], 1.0001f, acc);
#pragma unroll
  for (int r = 0; r < W; ++r) dst[(size_t)r * n + i] = v[r] + acc * 1e-9f;
}

// the same launch with its arguments the way tb_step_kernel gets them: one 616-byte struct by value, the pointers in its middle, and
// (STAGE) a 2.5 KB table copied to LDS behind a barrier before anything is computed
struct Big { float pad0[70]; const float* src; float* dst; int n, lanes; float pad1[76]; const float4* table; };
// KW: how many of the struct's padding words the launch reads (scalar loads from the kernel-argument segment, all of them
// needed before the first store): does a lone wave wait for its ARGUMENTS? (tb_step_kernel fetches 50-60 words of TbParams)
template <int R, int W, bool STAGE, int KW = 2>
__global__ void __launch_bounds__(64) step_like_big(Big A) {
  __shared__ float4 s_tab[160];
  const int lane = threadIdx.x;
  const int i = blockIdx.x * A.lanes + lane;
  float v[R];
  const bool live = lane < A.lanes && i < A.n;
  if (live) {
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = A.src[(size_t)r * A.n + i];
  }
  float extra = A.pad0[3] + A.pad1[70];
  if (KW > 2) {
#pragma unroll
    for (int k = 0; k < KW / 2; ++k) extra += A.pad0[k % 70] * 1e-30f + A.pad1[k % 76] * 1e-30f;
  }
  if (STAGE) {
    for (int k = threadIdx.x; k < 160; k += 64) s_tab[k] = A.table[k];
    __syncthreads();
    extra += s_tab[(lane * 7) % 160].x;
  }
  if (!live) return;
  float acc = extra;
#pragma unroll
  for (int r = 0; r < R; ++r) acc = fmaf(v[r], 1.0001f, acc);
#pragma unroll
  for (int r = 0; r < W; ++r) A.dst[(size_t)r * A.n + i] = v[r] + acc * 1e-9f;
}

template <bool STAGE, int KW = 2>
void run_big(hipStream_t s, float* a, float* b, const float4* table, int N, int T, const char* label) {
  constexpr int R = 36, W = 30;
  hipGraph_t g; hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int t = 0; t < T; ++t) {
    Big A = {};
    A.src = t % 2 ? b : a; A.dst = t % 2 ? a : b; A.n = N; A.lanes = 64; A.table = table;
    hipLaunchKernelGGL((step_like_big<R, W, STAGE, KW>), dim3(N / 64), dim3(64), 0, s, A);
  }
  CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int k = 0; k < 50; ++k) CHECK(hipGraphLaunch(ge, s));
  CHECK(hipStreamSynchronize(s));
  double best = 1e9;
  for (int k = 0; k < 15; ++k) {
    auto t0 = std::chrono::steady_clock::now();
    CHECK(hipGraphLaunch(ge, s)); CHECK(hipStreamSynchronize(s));
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (dt < best) best = dt;
  }
  printf("%s: %.2f us per launch\n", label, best / T * 1e6);
  CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
}

int main() {
  const int N = 4096, R = 36, W = 30, T = 1040;
  float *a, *b;
  CHECK(hipMalloc(&a, sizeof(float) * R * N)); CHECK(hipMalloc(&b, sizeof(float) * R * N));
  CHECK(hipMemset(a, 0, sizeof(float) * R * N)); CHECK(hipMemset(b, 0, sizeof(float) * R * N));
  hipStream_t s; CHECK(hipStreamCreate(&s));
  for (int lanes : {64, 32, 16, 8, 64}) {
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int t = 0; t < T; ++t)
      hipLaunchKernelGGL((step_like<R, W>), dim3((N + lanes - 1) / lanes), dim3(64), 0, s, t % 2 ? b : a, t % 2 ? a : b, N, lanes);
    CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int k = 0; k < 50; ++k) CHECK(hipGraphLaunch(ge, s));
    CHECK(hipStreamSynchronize(s));
    double best = 1e9;
    for (int k = 0; k < 15; ++k) {
      auto t0 = std::chrono::steady_clock::now();
      CHECK(hipGraphLaunch(ge, s)); CHECK(hipStreamSynchronize(s));
      double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (dt < best) best = dt;
    }
    printf("%2d lanes per wave = %4d waves: %.2f us per launch (%d rows read, %d written, %d envs)\n", lanes, (N + lanes - 1) / lanes, best / T * 1e6, R, W, N);
    CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
  }
  float4* table; CHECK(hipMalloc(&table, sizeof(float4) * 160)); CHECK(hipMemset(table, 0, sizeof(float4) * 160));
  run_big<false>(s, a, b, table, N, T, "616-byte argument struct, 64 full waves");
  run_big<true>(s, a, b, table, N, T, "616-byte argument struct + 2.5 KB table staged into LDS behind a barrier");
  run_big<false>(s, a, b, table, N, T, "616-byte argument struct, 64 full waves");
  run_big<false, 32>(s, a, b, table, N, T, "... 32 argument words read");
  run_big<false, 64>(s, a, b, table, N, T, "... 64 argument words read");
  run_big<false, 140>(s, a, b, table, N, T, "... 140 argument words read");
  run_big<false, 2>(s, a, b, table, N, T, "... 2 argument words read");
  return 0;
}
