// Cost of a device-wide barrier between dependent stages inside ONE persistent kernel vs. the launch boundary between dependent
// kernels in a HIP graph (tools/launch_floor.cpp: 1.65 us).  Each stage: every workgroup reads what ANOTHER workgroup (on another
// XCD: workgroup ids are dealt round-robin to the 8 XCDs) wrote in the previous stage, so the barrier must make writes visible
// across XCD L2s (agent-scope release / acquire).   hipcc --offload-arch=gfx950 -O3 tools/grid_barrier.cpp -o tools/bin/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned nwg, unsigned& epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                   // release: this workgroup's stores visible device-wide
    const unsigned target = (epoch + 1) * nwg;
    atomicAdd(counter, 1u);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    __threadfence();                                   // acquire
  }
  epoch += 1;
  __syncthreads();
}

__global__ void persistent(float* buf, unsigned* counter, int stages, int n) {
  unsigned epoch = 0;
  const unsigned nwg = gridDim.x;
  for (int s = 0; s < stages; ++s) {
    const int src = (blockIdx.x + 1) % nwg;            // another workgroup's slice (next id = another XCD)
    const float* in = buf + (size_t)(s & 1) * nwg * n + (size_t)src * n;
    float* out = buf + (size_t)((s + 1) & 1) * nwg * n + (size_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = in[i] + 1.0f;
    grid_barrier(counter, nwg, epoch);
  }
}

__global__ void stage_kernel(float* buf, int s, int n) {
  const unsigned nwg = gridDim.x;
  const int src = (blockIdx.x + 1) % nwg;
  const float* in = buf + (size_t)(s & 1) * nwg * n + (size_t)src * n;
  float* out = buf + (size_t)((s + 1) & 1) * nwg * n + (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = in[i] + 1.0f;
}

int main(int argc, char** argv) {
  const int stages = 200;
  for (int nwg : {32, 64, 128, 256}) {
    for (int n : {256, 4096}) {
      float* buf; unsigned* counter;
      hipMalloc(&buf, (size_t)2 * nwg * n * 4); hipMemset(buf, 0, (size_t)2 * nwg * n * 4);
      hipMalloc(&counter, 4);
      hipStream_t st; hipStreamCreate(&st);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms;
      // persistent
      hipMemsetAsync(counter, 0, 4, st);
      hipLaunchKernelGGL(persistent, dim3(nwg), dim3(256), 0, st, buf, counter, stages, n);
      hipStreamSynchronize(st);
      hipMemsetAsync(counter, 0, 4, st);
      hipEventRecord(e0, st);
      hipLaunchKernelGGL(persistent, dim3(nwg), dim3(256), 0, st, buf, counter, stages, n);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      const float per_barrier = ms * 1e3f / stages;
      // check
      float h; hipMemcpy(&h, buf + (size_t)(stages & 1) * nwg * n, 4, hipMemcpyDeviceToHost);
      // graph of dependent launches
      hipGraph_t g; hipGraphExec_t ex;
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
      for (int s = 0; s < stages; ++s) hipLaunchKernelGGL(stage_kernel, dim3(nwg), dim3(256), 0, st, buf, s, n);
      hipStreamEndCapture(st, &g); hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
      hipGraphLaunch(ex, st); hipStreamSynchronize(st);
      hipEventRecord(e0, st);
      for (int r = 0; r < 3; ++r) hipGraphLaunch(ex, st);
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("nwg %3d n %4d: persistent %.2f us/stage (value %.0f, expect %d)   graph launches %.2f us/stage\n", nwg, n, per_barrier, h, 2 * stages, ms * 1e3f / (3 * stages));
      hipFree(buf); hipFree(counter);
    }
  }
  return 0;
}
