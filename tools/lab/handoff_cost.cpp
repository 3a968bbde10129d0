// Round 3 probe: cost of a cross-stream hand-off between separately launched HIP graphs, by mechanism.
//   main chain: N dependent kernels (~5 us, 64 workgroups), side work: M kernels (~10 us, 256 workgroups), one "tail" kernel.
// serial modes (ONE cut: main graph -> side graph -> tail; nothing overlaps, so time - mode0 = hand-off overhead):
//   0  everything in one graph on one stream
//   4  plain events between launches (hipEventRecord / hipStreamWaitEvent)
//   5  external event nodes inside the graphs (hipEventRecordWithFlags(External) at the end of main, hipStreamWaitEvent(External)
//      at the head of side, ...), graphs launched back to back on two streams
//   6  device flags: main's last kernel publishes step to a flag, side's first kernel spins on it (one wave), etc.
// overlapped modes (S cuts; side group i may start when main segment i is done):
//   14 plain events, main cut into S graphs     16 device flags, ONE main graph + ONE side graph per step
//   7 / 17  stream memory operations (hipStreamWriteValue32 / hipStreamWaitValue32) instead of the one-lane kernels of 6 / 16:
//           NOT a hand-off inside a captured graph on this stack -- the serial form runs FASTER than the main chain alone, i.e. the
//           waits do not hold (the operations are not captured as dependencies); kept as the record of that
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
// flag protocol: flags[i] holds the step number whose stage i is complete; `step` lives in device memory (graphs are static)
__global__ void k_signal(unsigned* flag, const unsigned* step) {
  __hip_atomic_store(flag, *step, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_wait(const unsigned* flag, const unsigned* step, unsigned* timeouts) {
  unsigned want = *step;
  for (long it = 0; it < 4000000; ++it) {   // bounded: ~ a second at worst, never a hang
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= want) return;
    __builtin_amdgcn_s_sleep(8);
  }
  atomicAdd(timeouts, 1u);
}
__global__ void k_bump(unsigned* step) { *step += 1; }

static int run(int mode, int N, int M, int S, int mblocks, int miters, int sblocks, int siters) {
  float *a, *b, *c, *d; unsigned* dev;   // dev[0] = step, dev[1] = timeouts, dev[8 + i] = flags
  size_t bytes = (size_t)4096 * 256 * 4;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&dev, 4096));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes)); CK(hipMemset(d, 0, bytes)); CK(hipMemset(dev, 0, 4096));
  unsigned one = 1; CK(hipMemcpy(dev, &one, 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dev + 2, &one, 4, hipMemcpyHostToDevice));
  hipStream_t s, t; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<hipEvent_t> ev(S + 2);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto mk = [&](int i, hipStream_t st) { hipLaunchKernelGGL(k_work, dim3(mblocks), dim3(256), 0, st, (i & 1) ? a : b, (i & 1) ? b : a, miters); };
  auto sk = [&](int j, hipStream_t st) { hipLaunchKernelGGL(k_work, dim3(sblocks), dim3(256), 0, st, (j & 1) ? c : d, (j & 1) ? d : c, siters); };
  unsigned* flags = dev + 8;
  std::vector<hipGraphExec_t> gm, gs; hipGraphExec_t gtail = nullptr;
  auto endcap = [&](hipStream_t st, hipGraphExec_t* ex) -> int { hipGraph_t g; CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(ex, g, nullptr, nullptr, 0)); return 0; };
  int every = N / M;
  bool serial = mode < 10;
  int base = mode % 10;
  int nseg = serial ? 1 : S;
  if (base == 0) {
    hipGraphExec_t ex; CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int j = 0;
    for (int i = 0; i < N; ++i) { mk(i, s); if ((i + 1) % every == 0 && j < M) sk(j++, s); }
    mk(N, s);
    if (endcap(s, &ex)) return 1;
    gm.push_back(ex);
  } else if (base == 4 || base == 5) {
    // main graphs (cut into nseg), side graphs, tail graph
    int i = 0, j = 0;
    for (int seg = 0; seg < nseg; ++seg) {
      hipGraphExec_t ex; CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      int end = (seg == nseg - 1) ? N : (seg + 1) * (N / nseg);
      for (; i < end; ++i) mk(i, s);
      if (base == 5) CK(hipEventRecordWithFlags(ev[seg], s, hipEventRecordExternal));
      if (endcap(s, &ex)) return 1;
      gm.push_back(ex);
      CK(hipStreamBeginCapture(t, hipStreamCaptureModeThreadLocal));
      if (base == 5) CK(hipStreamWaitEvent(t, ev[seg], hipEventWaitExternal));
      while (j < M && (j + 1) * every <= end) sk(j++, t);
      if (base == 5 && seg == nseg - 1) CK(hipEventRecordWithFlags(ev[S], t, hipEventRecordExternal));
      if (endcap(t, &ex)) return 1;
      gs.push_back(ex);
    }
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    if (base == 5) CK(hipStreamWaitEvent(s, ev[S], hipEventWaitExternal));
    mk(N, s);
    if (endcap(s, &gtail)) return 1;
  } else if (base == 7) {
    // as 6, but the hand-offs are stream memory operations (no kernels): hipStreamWriteValue32(flag, 1) on the producer,
    // hipStreamWaitValue32(flag == 1) + hipStreamWriteValue32(flag, 0) on the consumer (constants: a static graph can hold them)
    hipGraphExec_t ex; CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
      mk(i, s);
      for (int seg = 0; seg < nseg; ++seg) {
        int end = (seg == nseg - 1) ? N : (seg + 1) * (N / nseg);
        if (i + 1 == end) CK(hipStreamWriteValue32(s, flags + seg, 1, 0));
      }
    }
    CK(hipStreamWaitValue32(s, flags + S, 1, hipStreamWaitValueEq, 0xFFFFFFFF));
    CK(hipStreamWriteValue32(s, flags + S, 0, 0));
    mk(N, s);
    if (endcap(s, &ex)) return 1;
    gm.push_back(ex);
    CK(hipStreamBeginCapture(t, hipStreamCaptureModeThreadLocal));
    int j = 0;
    for (int seg = 0; seg < nseg; ++seg) {
      int end = (seg == nseg - 1) ? N : (seg + 1) * (N / nseg);
      CK(hipStreamWaitValue32(t, flags + seg, 1, hipStreamWaitValueEq, 0xFFFFFFFF));
      CK(hipStreamWriteValue32(t, flags + seg, 0, 0));
      while (j < M && (j + 1) * every <= end) sk(j++, t);
    }
    CK(hipStreamWriteValue32(t, flags + S, 1, 0));
    if (endcap(t, &ex)) return 1;
    gs.push_back(ex);
  } else if (base == 6) {
    // ONE main graph with signal kernels at the cuts, ONE side graph with wait kernels, tail behind a wait kernel on main
    hipGraphExec_t ex; CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
      mk(i, s);
      for (int seg = 0; seg < nseg; ++seg) {
        int end = (seg == nseg - 1) ? N : (seg + 1) * (N / nseg);
        if (i + 1 == end) hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, s, flags + seg, dev);
      }
    }
    hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, s, flags + S, dev, dev + 1);
    mk(N, s);
    hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, s, dev);
    if (endcap(s, &ex)) return 1;
    gm.push_back(ex);
    CK(hipStreamBeginCapture(t, hipStreamCaptureModeThreadLocal));
    int j = 0;
    for (int seg = 0; seg < nseg; ++seg) {
      int end = (seg == nseg - 1) ? N : (seg + 1) * (N / nseg);
      hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, t, flags + seg, dev + 2, dev + 1);
      while (j < M && (j + 1) * every <= end) sk(j++, t);
    }
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, t, flags + S, dev + 2);
    hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, t, dev + 2);
    if (endcap(t, &ex)) return 1;
    gs.push_back(ex);
  }
  auto step = [&]() -> int {
    if (base == 0) { CK(hipGraphLaunch(gm[0], s)); }
    else if (base == 4) {
      for (int seg = 0; seg < nseg; ++seg) {
        CK(hipGraphLaunch(gm[seg], s)); CK(hipEventRecord(ev[seg], s)); CK(hipStreamWaitEvent(t, ev[seg], 0)); CK(hipGraphLaunch(gs[seg], t));
      }
      CK(hipEventRecord(ev[S], t)); CK(hipStreamWaitEvent(s, ev[S], 0)); CK(hipGraphLaunch(gtail, s));
    } else if (base == 5) {
      for (int seg = 0; seg < nseg; ++seg) { CK(hipGraphLaunch(gm[seg], s)); CK(hipGraphLaunch(gs[seg], t)); }
      CK(hipGraphLaunch(gtail, s));
    } else {
      // each stream counts its own steps (dev[0] main, dev[2] side), so neither graph reads a counter the other one bumps
      CK(hipGraphLaunch(gs[0], t)); CK(hipGraphLaunch(gm[0], s));
    }
    return 0;
  };
  for (int r = 0; r < 3; ++r) { if (step()) return 1; CK(hipDeviceSynchronize()); }
  CK(hipEventRecord(e0, s));
  const int R = 20;
  for (int r = 0; r < R; ++r) { if (step()) return 1; }
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned h[2]; CK(hipMemcpy(h, dev, 8, hipMemcpyDeviceToHost));
  printf("mode %2d N=%d M=%d S=%d: %.1f us per step%s (timeouts %u)\n", mode, N, M, S, ms * 1000 / R, "", h[1]);
  fflush(stdout);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(d); (void)hipFree(dev);
  return 0;
}
int main(int argc, char** argv) {
  int miters = argc > 1 ? atoi(argv[1]) : 200, siters = argc > 2 ? atoi(argv[2]) : 500;
  run(0, 240, 1, 1, 64, miters, 256, siters);      // main alone (+1 side kernel)
  for (int mode : {0, 4, 5, 6, 7}) if (run(mode, 240, 32, 1, 64, miters, 256, siters)) printf("mode %d failed\n", mode);
  for (int S : {4, 8, 32}) for (int mode : {14, 15, 16, 17}) if (run(mode, 240, 32, S, 64, miters, 256, siters)) printf("mode %d failed\n", mode);
  return 0;
}
