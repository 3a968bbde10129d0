// Round 3 probe: is a device-wide barrier INSIDE a kernel cheaper than a dependent launch boundary (~1.65 us) on MI355X?
// Decides whether the two-launch GroupNorm-epilogue backward (reduce2 -> apply_gn2) of the 16^3 / 8^3 levels is worth folding into
// one cooperative launch.  Chain of 200 steps in a HIP graph; a step is  pass A (read 16 B / thread, partial sums to memory) ->
// exchange -> pass B (read the partials, write 16 B / thread).
//   hipcc --offload-arch=gfx950 -O2 tools/grid_barrier_probe.cpp -o tools/build/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void passA(const float4* x, float* part) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float4 v = x[i];
  float s = v.x + v.y + v.z + v.w;
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) part[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;
}
__global__ void passB(const float4* x, const float* part, float4* y, int nb) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float s = 0;
  for (int k = threadIdx.x & 63; k < nb * 4; k += 64) s += part[k];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  float4 v = x[i];
  y[i] = make_float4(v.x + s, v.y + s, v.z + s, v.w + s);
}
// one launch: A, device-wide barrier (monotonic ticket counter, all workgroups resident), B from registers
template <int SCOPE>
__global__ void fused(const float4* x, float* part, float4* y, unsigned* counter, int nb) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float4 v = x[i];
  float s = v.x + v.y + v.z + v.w;
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(&part[blockIdx.x * 4 + (threadIdx.x >> 6)], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = (ticket / nb + 1) * nb;
    while ((int)(__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  float t = 0;
  for (int k = threadIdx.x & 63; k < nb * 4; k += 64) t += __hip_atomic_load(&part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int o = 32; o; o >>= 1) t += __shfl_xor(t, o);
  y[i] = make_float4(v.x + t, v.y + t, v.z + t, v.w + t);
}
int main() {
  const int STEPS = 200;
  for (int nb : {4, 16, 64, 256}) {
    float4 *x, *y; float* part; unsigned* cnt;
    CK(hipMalloc(&x, nb * 256 * 16)); CK(hipMalloc(&y, nb * 256 * 16)); CK(hipMalloc(&part, 4096 * 4)); CK(hipMalloc(&cnt, 4 * STEPS));
    CK(hipMemset(x, 0, nb * 256 * 16)); CK(hipMemset(cnt, 0, 4 * STEPS));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < STEPS; ++k) {
        if (mode == 0) {
          hipLaunchKernelGGL(passA, dim3(nb), dim3(256), 0, s, (k & 1) ? y : x, part);
          hipLaunchKernelGGL(passB, dim3(nb), dim3(256), 0, s, (k & 1) ? y : x, part, (k & 1) ? x : y, nb);
        } else {
          hipLaunchKernelGGL(fused<0>, dim3(nb), dim3(256), 0, s, (k & 1) ? y : x, part, (k & 1) ? x : y, cnt + k, nb);
        }
      }
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      CK(hipEventRecord(a, s));
      for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("%3d workgroups, %s: %.2f us per step\n", nb, mode ? "ONE launch with a device-wide barrier" : "two dependent launches          ", ms * 1000 / (10 * STEPS));
    }
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(part); (void)hipFree(cnt);
  }
  return 0;
}
