// Measures the per-kernel floor of dependent launches on one stream: eager vs HIP-graph replay.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void tiny256(float* p) { p[blockIdx.x * 256 + threadIdx.x] += 1.f; }
int main() {
  float* d; hipMalloc(&d, 256 * 256 * 4); hipMemset(d, 0, 256 * 256 * 4);
  hipStream_t s; hipStreamCreate(&s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 1000;
  for (int variant = 0; variant < 2; ++variant) {
    for (int i = 0; i < 10; ++i) { if (variant) hipLaunchKernelGGL(tiny256, dim3(256), dim3(256), 0, s, d); else hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d); }
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < N; ++i) { if (variant) hipLaunchKernelGGL(tiny256, dim3(256), dim3(256), 0, s, d); else hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d); }
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d eager: %.2f us/kernel\n", variant, ms * 1e3 / N);
    hipGraph_t g; hipGraphExec_t ex;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) { if (variant) hipLaunchKernelGGL(tiny256, dim3(256), dim3(256), 0, s, d); else hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d); }
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    hipGraphLaunch(ex, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < 5; ++r) hipGraphLaunch(ex, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("variant %d graph: %.2f us/kernel\n", variant, ms * 1e3 / (N * 5));
  }
  return 0;
}
