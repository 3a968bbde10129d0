// Is a CU mask (hipExtStreamCreateWithCUMask) honoured by eager launches / by a graph launched into the masked stream?
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.cpp -o tools/build/cumask_probe && tools/build/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) a = fmaf(a, b, 1e-7f);
  if (a == 123.f) out[0] = a;
}
static float timed(hipStream_t s, float* buf, int wgs, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, s, buf, 20000);
  hipStreamSynchronize(s);
  hipEventRecord(a, s);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, s, buf, 20000);
  hipEventRecord(b, s); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
int main() {
  float* buf; CK(hipMalloc(&buf, 1024));
  hipStream_t plain; CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  printf("plain stream, 2048 WGs: %.1f us\n", timed(plain, buf, 2048, 10));
  for (int keep : {128, 64, 32}) {
    // 256 CUs = 8 XCDs x 32; mask bit i = CU i (the runtime's numbering interleaves XCDs)
    std::vector<uint32_t> mask(8, 0);
    for (int i = 0; i < keep; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t m; CK(hipExtStreamCreateWithCUMask(&m, 8, mask.data()));
    printf("mask %3d CUs (low bits), eager: %.1f us\n", keep, timed(m, buf, 2048, 10));
    // graph captured on the masked stream, launched into it
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(m, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, m, buf, 20000);
    CK(hipStreamEndCapture(m, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    CK(hipGraphLaunch(ge, m)); hipStreamSynchronize(m);
    hipEventRecord(a, m); CK(hipGraphLaunch(ge, m)); hipEventRecord(b, m); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mask %3d CUs, graph launched into the masked stream: %.1f us per kernel\n", keep, ms / 10 * 1e3f);
    // strided mask: every (256/keep)-th CU
    std::vector<uint32_t> m2(8, 0);
    for (int i = 0; i < 256; i += 256 / keep) m2[i / 32] |= 1u << (i % 32);
    hipStream_t s2; CK(hipExtStreamCreateWithCUMask(&s2, 8, m2.data()));
    printf("mask %3d CUs (strided), eager: %.1f us\n", keep, timed(s2, buf, 2048, 10));
  }
  return 0;
}
