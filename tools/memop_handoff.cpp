// What does a hand-off cost on the WAITING and on the SIGNALLING chain: one-lane flag kernels (libn3d's n3d_sync_signal / n3d_sync_wait)
// against the runtime's stream memory operations (hipStreamWriteValue32 / hipStreamWaitValue32: command-processor packets, no dispatch).
//   hipcc -O2 --offload-arch=gfx950 tools/memop_handoff.cpp -o tools/memop_handoff && tools/memop_handoff
// Chain A: [work, SIGNAL(a_i)] [work, WAIT(b_i)] ... ; chain B: [WAIT(a_i), work, SIGNAL(b_i)] ...; timed on A with events, eager and as graphs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void work(float* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = p[i];
  for (int k = 0; k < n; ++k) v = v * 1.0001f + 0.5f;
  p[i] = v;
}
__global__ void sig_k(unsigned* f, unsigned v) { __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void wait_clear_k(unsigned* f) {
  for (long it = 0; it < 20000; ++it) {
    if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
    __builtin_amdgcn_s_sleep(8);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __hip_atomic_store(f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void wait_k(const unsigned* f, unsigned v) {
  for (long it = 0; it < 100000000; ++it) {
    if ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - v) >= 0) break;
    __builtin_amdgcn_s_sleep(8);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

enum Mode { NONE, KERNEL, MEMOP, MIXED };   // MIXED: signal = memory-operation write, wait = one-lane kernel that polls and clears
static unsigned* flagp[64];  // one 8-byte signal-memory allocation each (hipMallocSignalMemory takes exactly 8 bytes)
#define FLAG(i) (flagp[i])
static float *bufA, *bufB;
static const int NH = 32;    // hand-off pairs per pass

static void signal(Mode m, hipStream_t s, unsigned* f, unsigned v) {
  if (m == KERNEL) hipLaunchKernelGGL(sig_k, dim3(1), dim3(1), 0, s, f, v);
  else if (m == MEMOP) CK(hipStreamWriteValue32(s, f, v, 0));
}
static void wait(Mode m, hipStream_t s, unsigned* f, unsigned v) {
  if (m == KERNEL) hipLaunchKernelGGL(wait_k, dim3(1), dim3(1), 0, s, f, v);
  else if (m == MEMOP) CK(hipStreamWaitValue32(s, f, v, hipStreamWaitValueGte, 0xffffffffu));
}
static void issueA(Mode m, hipStream_t a, unsigned v, int wn) {
  for (int i = 0; i < NH; ++i) {
    hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, a, bufA, wn);
    signal(m, a, FLAG(2 * i), v);
    hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, a, bufA, wn);
    wait(m, a, FLAG(2 * i + 1), v);
  }
}
static void issueB(Mode m, hipStream_t b, unsigned v, int wn) {
  for (int i = 0; i < NH; ++i) {
    wait(m, b, FLAG(2 * i), v);
    hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, b, bufB, wn / 4);
    signal(m, b, FLAG(2 * i + 1), v);
  }
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  const bool plain = argc > 2;      // any third argument: ordinary hipMalloc words instead of signal memory
  for (int i = 0; i < 2 * NH; ++i) {
    if (plain) CK(hipMalloc((void**)&flagp[i], 8)); else CK(hipExtMallocWithFlags((void**)&flagp[i], 8, hipMallocSignalMemory));
    CK(hipMemset(flagp[i], 0, 8));
  }
  printf("flags in %s\n", plain ? "hipMalloc memory" : "signal memory");
  CK(hipMalloc(&bufA, 64 * 256 * 4)); CK(hipMalloc(&bufB, 64 * 256 * 4));
  CK(hipMemset(bufA, 0, 64 * 256 * 4)); CK(hipMemset(bufB, 0, 64 * 256 * 4));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int wn = argc > 1 ? atoi(argv[1]) : 200;
  unsigned step = 0;
  const char* names[4] = {"no hand-offs (chain A alone: 2 x 32 work kernels)", "flag kernels", "stream memory operations", "memory-operation signal + polling kernel that clears"};
  // ---- eager
  for (int m = 0; m < 3; ++m) {
    float best = 1e9f;
    for (int rep = 0; rep < 8; ++rep) {
      ++step;
      CK(hipEventRecord(e0, a));
      if (m != NONE) issueB((Mode)m, b, step, wn);
      issueA((Mode)m, a, step, wn);
      CK(hipEventRecord(e1, a));
      CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 1 && ms < best) best = ms;
    }
    printf("eager  %-52s %8.1f us per pass  (%.2f us per work kernel or hand-off slot)\n", names[m], best * 1e3f, best * 1e3f / (2 * NH));
  }
  // ---- graphs, built node by node (stream memory operations are NOT recorded by stream capture on this stack: a captured chain came
  // out with its kernel nodes only): inside a graph the value is a constant -- 1 = set -- and the consumer clears the flag after its wait
  hipCtx_t ctx; CK(hipCtxGetCurrent(&ctx));
  static float* argA; static float* argB; static int argn, argn4; static unsigned one = 1u, zero = 0u;
  argA = bufA; argB = bufB; argn = wn; argn4 = wn / 4;
  struct Builder {
    hipGraph_t g; hipGraphNode_t last = nullptr; hipCtx_t ctx; std::vector<void*> keep;
    void dep(hipGraphNode_t n) { last = n; }
    void kernel(void* fn, dim3 grid, dim3 blk, void** args) {
      hipKernelNodeParams kp = {}; kp.func = fn; kp.gridDim = grid; kp.blockDim = blk; kp.kernelParams = args;
      hipGraphNode_t n; CK(hipGraphAddKernelNode(&n, g, last ? &last : nullptr, last ? 1 : 0, &kp)); dep(n);
    }
    void memops(std::vector<hipStreamBatchMemOpParams> ops) {
      hipBatchMemOpNodeParams np = {}; np.ctx = ctx; np.count = (unsigned)ops.size(); np.paramArray = ops.data(); np.flags = 0;
      hipGraphNode_t n; CK(hipGraphAddBatchMemOpNode(&n, g, last ? &last : nullptr, last ? 1 : 0, &np)); dep(n);
    }
  };
  auto op_wait = [](unsigned* f) { hipStreamBatchMemOpParams o = {}; o.operation = hipStreamMemOpWaitValue32; o.waitValue.address = (hipDeviceptr_t)f; o.waitValue.value = 1u; o.waitValue.flags = hipStreamWaitValueGte; return o; };
  auto op_write = [](unsigned* f, unsigned v) { hipStreamBatchMemOpParams o = {}; o.operation = hipStreamMemOpWriteValue32; o.writeValue.address = (hipDeviceptr_t)f; o.writeValue.value = v; o.writeValue.flags = 0; return o; };
  for (int m = 0; m < 4; ++m) {
    Builder A, B; A.ctx = B.ctx = ctx; CK(hipGraphCreate(&A.g, 0)); CK(hipGraphCreate(&B.g, 0));
    static unsigned* fp[64]; for (int i = 0; i < 2 * NH; ++i) fp[i] = flagp[i];
    for (int i = 0; i < NH; ++i) {
      void* wa[2] = {&argA, &argn}; void* wb[2] = {&argB, &argn4};
      void* s_a1[2] = {&fp[2 * i], &one}; void* w_b[2] = {&fp[2 * i + 1], &one}; void* s_b0[2] = {&fp[2 * i + 1], &zero};
      void* w_a[2] = {&fp[2 * i], &one}; void* s_a0[2] = {&fp[2 * i], &zero}; void* s_b1[2] = {&fp[2 * i + 1], &one};
      A.kernel((void*)work, dim3(64), dim3(256), wa);
      if (m == KERNEL) A.kernel((void*)sig_k, dim3(1), dim3(1), s_a1);
      if (m == MEMOP || m == MIXED) A.memops({op_write(fp[2 * i], 1u)});
      A.kernel((void*)work, dim3(64), dim3(256), wa);
      if (m == KERNEL) { A.kernel((void*)wait_k, dim3(1), dim3(1), w_b); A.kernel((void*)sig_k, dim3(1), dim3(1), s_b0); }
      if (m == MEMOP) A.memops({op_wait(fp[2 * i + 1]), op_write(fp[2 * i + 1], 0u)});
      void* wc_b[1] = {&fp[2 * i + 1]}; void* wc_a[1] = {&fp[2 * i]};
      if (m == MIXED) A.kernel((void*)wait_clear_k, dim3(1), dim3(1), wc_b);
      if (m == KERNEL) { B.kernel((void*)wait_k, dim3(1), dim3(1), w_a); B.kernel((void*)sig_k, dim3(1), dim3(1), s_a0); }
      if (m == MEMOP) B.memops({op_wait(fp[2 * i]), op_write(fp[2 * i], 0u)});
      if (m == MIXED) B.kernel((void*)wait_clear_k, dim3(1), dim3(1), wc_a);
      if (m != NONE) B.kernel((void*)work, dim3(64), dim3(256), wb);
      if (m == KERNEL) B.kernel((void*)sig_k, dim3(1), dim3(1), s_b1);
      if (m == MEMOP || m == MIXED) B.memops({op_write(fp[2 * i + 1], 1u)});
    }
    hipGraphExec_t xa = nullptr, xb = nullptr;
    CK(hipGraphInstantiate(&xa, A.g, nullptr, nullptr, 0));
    if (m != NONE) CK(hipGraphInstantiate(&xb, B.g, nullptr, nullptr, 0));
    for (int i = 0; i < 2 * NH; ++i) CK(hipMemset(flagp[i], 0, 8));
    CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 10; ++rep) {
      if (xb) CK(hipGraphLaunch(xb, b));
      CK(hipEventRecord(e0, a));
      CK(hipGraphLaunch(xa, a));
      CK(hipEventRecord(e1, a));
      CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 1 && ms < best) best = ms;
    }
    size_t nn = 0; CK(hipGraphGetNodes(A.g, nullptr, &nn));
    printf("graph  %-52s %8.1f us per pass  (%.2f us per slot; %zu nodes in chain A's graph)\n", names[m], best * 1e3f, best * 1e3f / (2 * NH), nn);
    unsigned left = 0; for (int i = 0; i < 2 * NH; ++i) { unsigned v; CK(hipMemcpy(&v, flagp[i], 4, hipMemcpyDeviceToHost)); left += v; }
    printf("       flags left set after the last pass: %u (0 = every hand-off was consumed)\n", left);
  }
  return 0;
}
