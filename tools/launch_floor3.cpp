// Does code size (cold instruction cache) explain the ~5-8 us floor of our small kernels?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
// straight-line code of ~N*8 bytes per REP: each step depends on the previous (executed once, no loop)
template <int REP, int SALT>
__global__ void k_code(float* p, const float* q) {
  float v = q[threadIdx.x];
#pragma unroll
  for (int i = 0; i < REP; ++i) v = fmaf(v, 1.0001f + 0.001f * (float)((i * 7 + SALT) % 13), 0.5f + (float)((i + SALT) % 5));
  p[threadIdx.x] = v;
}
template <int REP>
static int run(hipStream_t s, float* a, float* b, hipEvent_t e0, hipEvent_t e1, int blocks) {
  const int N = 400;
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < N; ++i) {
    float* dst = (i & 1) ? a : b; const float* src = (i & 1) ? b : a;
    switch (i % 4) {   // four different kernels of the same size take turns
      case 0: hipLaunchKernelGGL((k_code<REP, 0>), dim3(blocks), dim3(256), 0, s, dst, src); break;
      case 1: hipLaunchKernelGGL((k_code<REP, 1>), dim3(blocks), dim3(256), 0, s, dst, src); break;
      case 2: hipLaunchKernelGGL((k_code<REP, 2>), dim3(blocks), dim3(256), 0, s, dst, src); break;
      default: hipLaunchKernelGGL((k_code<REP, 3>), dim3(blocks), dim3(256), 0, s, dst, src); break;
    }
  }
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("straight-line %5d FMAs (~%3d KB code), %3d blocks: %.2f us/kernel\n", REP, REP * 8 / 1024, blocks, ms * 1e3 / (N * 5));
  return 0;
}
int main() {
  float *a, *b;
  CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {1, 64}) {
    run<16>(s, a, b, e0, e1, blocks); run<256>(s, a, b, e0, e1, blocks); run<1024>(s, a, b, e0, e1, blocks); run<4096>(s, a, b, e0, e1, blocks);
  }
  return 0;
}
