// What makes a tiny dependent kernel slow on this box?  Graph-replay timing of kernel variants.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
struct Big { float* p; const float* q; long a[16]; };
__global__ void k_trivial(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
__global__ void k_load1(float* p, const float* q) { p[blockIdx.x * 256 + threadIdx.x] = q[blockIdx.x * 256 + threadIdx.x] + 1.f; }
__global__ void k_load2(float* p, const float* q, const int* idx) { int i = idx[threadIdx.x]; p[blockIdx.x * 256 + threadIdx.x] = q[blockIdx.x * 256 + i] + 1.f; }
__global__ void k_lds(float* p, const float* q) { __shared__ float s[256]; s[threadIdx.x] = q[blockIdx.x * 256 + threadIdx.x]; __syncthreads(); p[blockIdx.x * 256 + threadIdx.x] = s[255 - threadIdx.x]; }
__global__ void k_big(Big b) { b.p[blockIdx.x * 256 + threadIdx.x] = b.q[blockIdx.x * 256 + threadIdx.x] + (float)b.a[3]; }
__global__ __launch_bounds__(1024) void k_1024(float* p, const float* q) { p[blockIdx.x * 1024 + threadIdx.x] = q[blockIdx.x * 1024 + threadIdx.x] + 1.f; }
__global__ void k_dbl(double* p, const double* q) { double v = q[threadIdx.x]; for (int i = 0; i < 16; ++i) v = v / 1.0000001 + sqrt(v + i); p[threadIdx.x] = v; }
int main() {
  float *a, *b; int* idx; double *da, *db;
  CK(hipMalloc(&a, 1 << 24)); CK(hipMalloc(&b, 1 << 24)); CK(hipMalloc(&idx, 4096)); CK(hipMalloc(&da, 1 << 16)); CK(hipMalloc(&db, 1 << 16));
  CK(hipMemset(a, 0, 1 << 24)); CK(hipMemset(b, 0, 1 << 24)); CK(hipMemset(idx, 0, 4096)); CK(hipMemset(da, 0, 1 << 16)); CK(hipMemset(db, 0, 1 << 16));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 400;
  for (int variant = 0; variant < 11; ++variant) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) {
      float* dst = (i & 1) ? a : b; const float* src = (i & 1) ? b : a;   // ping-pong: each kernel reads its predecessor's output
      switch (variant) {
        case 0: hipLaunchKernelGGL(k_trivial, dim3(1), dim3(64), 0, s, dst); break;
        case 1: hipLaunchKernelGGL(k_load1, dim3(4), dim3(256), 0, s, dst, src); break;
        case 2: hipLaunchKernelGGL(k_load1, dim3(256), dim3(256), 0, s, dst, src); break;
        case 3: hipLaunchKernelGGL(k_load1, dim3(4096), dim3(256), 0, s, dst, src); break;
        case 4: hipLaunchKernelGGL(k_load2, dim3(4), dim3(256), 0, s, dst, src, idx); break;
        case 5: hipLaunchKernelGGL(k_lds, dim3(4), dim3(256), 0, s, dst, src); break;
        case 6: { Big bg; bg.p = dst; bg.q = src; for (int k = 0; k < 16; ++k) bg.a[k] = k; hipLaunchKernelGGL(k_big, dim3(4), dim3(256), 0, s, bg); } break;
        case 7: hipLaunchKernelGGL(k_1024, dim3(1), dim3(1024), 0, s, dst, src); break;
        case 8: hipLaunchKernelGGL(k_dbl, dim3(1), dim3(256), 0, s, (i & 1) ? da : db, (i & 1) ? db : da); break;
        case 9: hipLaunchKernelGGL(k_lds, dim3(4), dim3(256), 32768, s, dst, src); break;
        case 10: hipLaunchKernelGGL(k_load1, dim3(4, 4, 2), dim3(256), 0, s, dst, src); break;
      }
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ex, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const char* names[] = {"trivial 1x64", "load1 4x256", "load1 256x256", "load1 4096x256 (4 MB)", "dependent loads 4x256", "lds+sync 4x256", "big kernarg", "1 block x1024", "fp64 div/sqrt chain", "lds 32KB dyn", "3-D grid 32 blocks"};
    printf("%-28s %.2f us/kernel\n", names[variant], ms * 1e3 / (N * 5));
  }
  return 0;
}
