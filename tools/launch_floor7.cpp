// Cold instruction fetch: time per launch of a kernel whose body is REP straight-line dependent FMAs (8 bytes each),
// executed `loops` times (the second pass runs from a warm instruction cache).  400 launches in one captured graph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int REP, int SALT = 0>
__global__ void k_code(float* p, const float* q, int loops) {
  float v = q[threadIdx.x], a = q[threadIdx.x + 64], b = q[threadIdx.x + 128];
  for (int l = 0; l < loops; ++l) {
#pragma unroll
    for (int i = 0; i < REP; ++i) v = fmaf(v, a, b);
    asm volatile("" : "+v"(v));
  }
  p[threadIdx.x] = v;
}
template <int REP>
static int run(hipStream_t s, float* a, float* b, hipEvent_t e0, hipEvent_t e1, int blocks, int loops) {
  const int N = 400;
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k_code<REP>), dim3(blocks), dim3(64), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, loops);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%5d FMAs (%5.1f KB) x %d pass(es), %4d blocks: %6.2f us/kernel\n", REP, REP * 8 / 1024.0, loops, blocks, ms * 1e3 / (N * 5));
  return 0;
}
// eight distinct kernels of the same size, round-robin: every launch starts with its code cold in the instruction caches
template <int REP>
static int run_mixed(hipStream_t s, float* a, float* b, hipEvent_t e0, hipEvent_t e1, int blocks, int loops) {
  const int N = 400;
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < N; ++i) {
    float* d = (i & 1) ? a : b; const float* q = (i & 1) ? b : a;
    switch (i % 8) {
      case 0: hipLaunchKernelGGL((k_code<REP, 1>), dim3(blocks), dim3(64), 0, s, d, q, loops); break;
      case 1: hipLaunchKernelGGL((k_code<REP, 2>), dim3(blocks), dim3(64), 0, s, d, q, loops); break;
      case 2: hipLaunchKernelGGL((k_code<REP, 3>), dim3(blocks), dim3(64), 0, s, d, q, loops); break;
      case 3: hipLaunchKernelGGL((k_code<REP, 4>), dim3(blocks), dim3(64), 0, s, d, q, loops); break;
      case 4: hipLaunchKernelGGL((k_code<REP, 5>), dim3(blocks), dim3(64), 0, s, d, q, loops); break;
      case 5: hipLaunchKernelGGL((k_code<REP, 6>), dim3(blocks), dim3(64), 0, s, d, q, loops); break;
      case 6: hipLaunchKernelGGL((k_code<REP, 7>), dim3(blocks), dim3(64), 0, s, d, q, loops); break;
      default: hipLaunchKernelGGL((k_code<REP, 8>), dim3(blocks), dim3(64), 0, s, d, q, loops); break;
    }
  }
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("MIXED %5d FMAs (%5.1f KB) x %d pass(es), %4d blocks: %6.2f us/kernel\n", REP, REP * 8 / 1024.0, loops, blocks, ms * 1e3 / (N * 5));
  return 0;
}
int main() {
  float *a, *b;
  CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {1, 1024}) for (int loops : {1, 2}) {
    run<128>(s, a, b, e0, e1, blocks, loops); run<256>(s, a, b, e0, e1, blocks, loops); run<512>(s, a, b, e0, e1, blocks, loops);
    run<768>(s, a, b, e0, e1, blocks, loops); run<1024>(s, a, b, e0, e1, blocks, loops); run<1536>(s, a, b, e0, e1, blocks, loops);
    run<2048>(s, a, b, e0, e1, blocks, loops); run<4096>(s, a, b, e0, e1, blocks, loops);
  }
  for (int blocks : {1, 1024}) {
    run_mixed<128>(s, a, b, e0, e1, blocks, 1); run_mixed<512>(s, a, b, e0, e1, blocks, 1); run_mixed<1024>(s, a, b, e0, e1, blocks, 1);
    run_mixed<2048>(s, a, b, e0, e1, blocks, 1); run_mixed<4096>(s, a, b, e0, e1, blocks, 1);
  }
  return 0;
}
