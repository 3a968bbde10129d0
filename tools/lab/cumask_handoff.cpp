// Round 3 probe: does a CU mask on the SIDE stream (hipExtStreamCreateWithCUMask, k compute units per XCD: tools/cumask_map.cpp)
// keep chip-filling side kernels from delaying a latency-bound main chain, and does a masked stream still hand off through device
// flags at the ~2 us of a plain / low-priority stream?  Main: 240 dependent 64-workgroup kernels (~5 us) with S signal kernels;
// side: S groups of wait + chip-filling kernels (2048 workgroups); graphs as the trainers use them (one per stream, launched
// back to back).   hipcc --offload-arch=gfx950 -O2 tools/cumask_handoff.cpp -o tools/build/cumask_handoff
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_work(float* p, const float* q, int iters) {
  int i = (blockIdx.x * 256 + threadIdx.x) & (4096 * 256 - 1);
  float v = q[i];
  for (int k = 0; k < iters; ++k) v = fmaf(v, 1.0001f, 0.5f);
  p[i] = v;
}
__global__ void k_signal(unsigned* flag, const unsigned* step) { __hip_atomic_store(flag, *step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_wait(const unsigned* flag, const unsigned* step, unsigned* timeouts) {
  unsigned want = *step;
  for (long it = 0; it < 400000; ++it) {
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= want) return;
    __builtin_amdgcn_s_sleep(8);
  }
  atomicAdd(timeouts, 1u);
}
__global__ void k_bump(unsigned* step) { *step += 1; }

static int run(const char* what, hipStream_t s, hipStream_t t, int M, int sblocks, int siters) {
  const int N = 240, S = 16;
  float *a, *b, *c, *d; unsigned* dev;
  size_t bytes = (size_t)4096 * 256 * 4;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&dev, 4096));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes)); CK(hipMemset(d, 0, bytes)); CK(hipMemset(dev, 0, 4096));
  unsigned one = 1; CK(hipMemcpy(dev, &one, 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dev + 2, &one, 4, hipMemcpyHostToDevice));
  unsigned* flags = dev + 8;
  hipGraph_t g; hipGraphExec_t gm, gs;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) {
    hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, 200);
    if ((i + 1) % (N / S) == 0) hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s, flags + (i + 1) / (N / S) - 1, dev);
  }
  hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, s, flags + S, dev, dev + 1);
  hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, s, dev);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&gm, g, nullptr, nullptr, 0));
  CK(hipStreamBeginCapture(t, hipStreamCaptureModeThreadLocal));
  for (int seg = 0; seg < S; ++seg) {
    hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, t, flags + seg, dev + 2, dev + 1);
    for (int j = 0; j < M; ++j) hipLaunchKernelGGL(k_work, dim3(sblocks), dim3(256), 0, t, (j & 1) ? c : d, (j & 1) ? d : c, siters);
  }
  hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, t, flags + S, dev + 2);
  hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, t, dev + 2);
  CK(hipStreamEndCapture(t, &g)); CK(hipGraphInstantiate(&gs, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r) { CK(hipGraphLaunch(gs, t)); CK(hipGraphLaunch(gm, s)); CK(hipDeviceSynchronize()); }
  CK(hipEventRecord(e0, s));
  const int R = 20;
  for (int r = 0; r < R; ++r) { CK(hipGraphLaunch(gs, t)); CK(hipGraphLaunch(gm, s)); }
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned h[2]; CK(hipMemcpy(h, dev, 8, hipMemcpyDeviceToHost));
  printf("%-44s side %d x %4d-workgroup kernels per cut: %7.1f us per step (time-outs %u)\n", what, M, sblocks, ms * 1000 / R, h[1]);
  fflush(stdout);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(d); (void)hipFree(dev);
  return 0;
}
static hipStream_t masked(int cus) {
  uint32_t w[8];
  for (int i = 0; i < 8; ++i) { w[i] = 0; for (int b = 0; b < 32; ++b) if (i * 32 + b < 8 * cus) w[i] |= 1u << b; }
  hipStream_t t = nullptr;
  if (hipExtStreamCreateWithCUMask(&t, 8, w) != hipSuccess) printf("mask create failed\n");
  return t;
}
int main(int argc, char** argv) {
  int siters = argc > 1 ? atoi(argv[1]) : 400;
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t plain, low; CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking)); CK(hipStreamCreateWithPriority(&low, hipStreamNonBlocking, lo));
  run("plain side stream, no side work", s, plain, 0, 2048, siters);
  run("plain side stream", s, plain, 2, 2048, siters);
  run("low-priority side stream", s, low, 2, 2048, siters);
  for (int cus : {28, 24, 16}) {
    for (int k = 0; k < 5; ++k) {   // several streams of one mask: each is a hardware queue of its own, placed in creation order
      hipStream_t m = masked(cus);
      char nm[64]; snprintf(nm, 64, "masked side stream #%d, %d CUs per XCD", k, cus);
      run(nm, s, m, 2, 2048, siters);
      if (k == 0) { snprintf(nm, 64, "masked #%d, %d CUs per XCD, no side work", k, cus); run(nm, s, m, 0, 2048, siters); }
    }
  }
  return 0;
}
