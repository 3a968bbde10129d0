// Does a forked side branch in a captured HIP graph overlap with the main chain on this stack?
// main chain: N dependent kernels of ~T us each; side chain: M kernels, kernel j depends on main kernel j*(N/M) (like a
// wgrad that needs the dY the main chain just produced); joined before the last main kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_work(float* p, const float* q, int iters) {
  int i = blockIdx.x * 256 + threadIdx.x;
  float v = q[i];
  for (int k = 0; k < iters; ++k) v = fmaf(v, 1.0001f, 0.5f);
  p[i] = v;
}
static int run(int N, int M, int blocks, int iters, int mode, int sblocks) {
  // mode 0: everything on one stream (N + M kernels); 1: side branch with per-kernel cross edges; 2: side branch forked once
  float *a, *b, *c, *d;
  size_t bytes = (size_t)2048 * 256 * 4;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes)); CK(hipMemset(d, 0, bytes));
  hipStream_t s, t; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<hipEvent_t> ev(M + 2);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipGraph_t g; hipGraphExec_t ex;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  int every = N / M, j = 0;
  if (mode == 2) { CK(hipEventRecord(ev[M], s)); CK(hipStreamWaitEvent(t, ev[M], 0)); }
  for (int i = 0; i < N; ++i) {
    hipLaunchKernelGGL(k_work, dim3(blocks), dim3(256), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, iters);
    if ((i + 1) % every == 0 && j < M) {
      if (mode == 0) hipLaunchKernelGGL(k_work, dim3(sblocks), dim3(256), 0, s, (j & 1) ? c : d, (j & 1) ? d : c, iters);
      else {
        if (mode == 1) { CK(hipEventRecord(ev[j], s)); CK(hipStreamWaitEvent(t, ev[j], 0)); }
        hipLaunchKernelGGL(k_work, dim3(sblocks), dim3(256), 0, t, (j & 1) ? c : d, (j & 1) ? d : c, iters);
      }
      ++j;
    }
  }
  if (mode != 0) { CK(hipEventRecord(ev[M + 1], t)); CK(hipStreamWaitEvent(s, ev[M + 1], 0)); }
  hipLaunchKernelGGL(k_work, dim3(blocks), dim3(256), 0, s, a, c, iters);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ex, s));
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("mode %d  N=%d M=%d blocks=%d/%d iters=%d: %.1f us per graph  (%.2f us per kernel)\n", mode, N, M, blocks, sblocks, iters, ms * 100, ms * 100 / (N + M + 1));
  hipFree(a); hipFree(b); hipFree(c); hipFree(d);
  return 0;
}
int main() {
  for (int iters : {200, 2000}) for (int blocks : {256, 1024}) {
    for (int mode : {0, 1, 2}) run(200, 50, blocks, iters, mode, blocks);
    run(200, 0 + 1, blocks, iters, 0, blocks);
  }
  return 0;
}
