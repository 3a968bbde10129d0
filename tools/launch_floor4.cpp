// Separate instruction-fetch cost from a low clock: the same dependent FMA chain as a loop (hot I-cache).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int REP, int SALT>
__global__ void k_loop(float* p, const float* q, int iters, long long* clk) {
  float v = q[threadIdx.x];
  long long t0 = __builtin_amdgcn_s_memtime(); long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < REP; ++i) v = fmaf(v, 1.0001f + 0.001f * (float)((i * 7 + SALT) % 13), 0.5f + (float)((i + SALT) % 5));
  }
  long long t1 = __builtin_amdgcn_s_memtime(); long long r1 = __builtin_amdgcn_s_memrealtime();
  p[threadIdx.x] = v;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int REP>
static int run(hipStream_t s, float* a, float* b, long long* clk, hipEvent_t e0, hipEvent_t e1, int iters) {
  const int N = 200;
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < N; ++i) {
    float* dst = (i & 1) ? a : b; const float* src = (i & 1) ? b : a;
    if (i & 1) hipLaunchKernelGGL((k_loop<REP, 0>), dim3(1), dim3(256), 0, s, dst, src, iters, clk);
    else hipLaunchKernelGGL((k_loop<REP, 1>), dim3(1), dim3(256), 0, s, dst, src, iters, clk);
  }
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
  printf("REP %5d x iters %3d (%6d dependent FMAs, ~%2d KB code): %.2f us/kernel; in-kernel %lld shader cycles, %lld x10ns => %.0f MHz\n", REP, iters, REP * iters,
         REP * 8 / 1024, ms * 1e3 / (N * 5), h[0], h[1], h[1] ? (double)h[0] / (h[1] * 0.01) : 0.0);
  return 0;
}
int main() {
  float *a, *b; long long* clk;
  CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMalloc(&clk, 64)); CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  run<256>(s, a, b, clk, e0, e1, 1); run<256>(s, a, b, clk, e0, e1, 4); run<256>(s, a, b, clk, e0, e1, 16);
  run<1024>(s, a, b, clk, e0, e1, 1); run<64>(s, a, b, clk, e0, e1, 16); run<64>(s, a, b, clk, e0, e1, 64);
  return 0;
}
