// Round 3 probe: what does it cost to run the weight-gradient kernels of a step on a SIDE stream next to the dependent main chain,
// when the two are separately launched HIP graphs tied by events (no intra-graph fork)?
//   main chain: N dependent kernels (~tm us each, few workgroups: latency-bound, like the deep-level launches)
//   side work : M kernels (~ts us each); side kernel j may start once main kernel (j+1)*N/M - 1 is done
// modes: 0  everything in ONE graph on one stream (today's schedule)
//        1  main chain cut into S graphs; after graph i: event -> side stream waits -> side graph i  (plain events between launches)
//        2  main chain ONE graph with S external event-record nodes; side graphs wait on them (hipEventRecordExternal)
//        3  main chain cut into S graphs, side work still inline (cost of the cuts alone)
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
static int run(int mode, int N, int M, int S, int mblocks, int miters, int sblocks, int siters) {
  float *a, *b, *c, *d;
  size_t bytes = (size_t)4096 * 256 * 4;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes)); CK(hipMemset(d, 0, bytes));
  hipStream_t s, t; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&t, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<hipEvent_t> ev(S + 1);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto mk = [&](int i) { hipLaunchKernelGGL(k_work, dim3(mblocks), dim3(256), 0, s, (i & 1) ? a : b, (i & 1) ? b : a, miters); };
  auto sk = [&](int j, hipStream_t st) { hipLaunchKernelGGL(k_work, dim3(sblocks), dim3(256), 0, st, (j & 1) ? c : d, (j & 1) ? d : c, siters); };
  std::vector<hipGraphExec_t> gm, gs;
  int nseg = (mode == 0 || mode == 2) ? 1 : S;
  int every = N / M;
  // main graphs
  {
    int i = 0, j = 0;
    for (int seg = 0; seg < nseg; ++seg) {
      hipGraph_t g; hipGraphExec_t ex;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      int end = (seg == nseg - 1) ? N : (seg + 1) * (N / nseg);
      for (; i < end; ++i) {
        mk(i);
        if ((i + 1) % every == 0 && j < M) {
          if (mode == 0 || mode == 3) sk(j, s);
          ++j;
        }
        if (mode == 2 && (i + 1) % (N / S) == 0) CK(hipEventRecordWithFlags(ev[(i + 1) / (N / S) - 1], s, hipEventRecordExternal));
      }
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
      gm.push_back(ex);
    }
  }
  // side graphs (modes 1, 2): side kernels whose main predecessor lies in segment seg
  if (mode == 1 || mode == 2) {
    int j = 0;
    for (int seg = 0; seg < S; ++seg) {
      int end = (seg == S - 1) ? N : (seg + 1) * (N / S);
      hipGraph_t g; hipGraphExec_t ex;
      CK(hipStreamBeginCapture(t, hipStreamCaptureModeThreadLocal));
      int cnt = 0;
      while (j < M && (j + 1) * every <= end) { sk(j, t); ++j; ++cnt; }
      if (cnt == 0) sk(j, t);   // keep the graph non-empty (never happens with M >= S)
      CK(hipStreamEndCapture(t, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
      gs.push_back(ex);
    }
  }
  auto step = [&]() -> int {
    if (mode == 0) { CK(hipGraphLaunch(gm[0], s)); }
    else if (mode == 3) { for (auto& g : gm) CK(hipGraphLaunch(g, s)); }
    else if (mode == 1) {
      for (int seg = 0; seg < S; ++seg) {
        CK(hipGraphLaunch(gm[seg], s));
        CK(hipEventRecord(ev[seg], s));
        CK(hipStreamWaitEvent(t, ev[seg], 0));
        CK(hipGraphLaunch(gs[seg], t));
      }
      CK(hipEventRecord(ev[S], t)); CK(hipStreamWaitEvent(s, ev[S], 0));
    } else {
      CK(hipGraphLaunch(gm[0], s));
      for (int seg = 0; seg < S; ++seg) { CK(hipStreamWaitEvent(t, ev[seg], 0)); CK(hipGraphLaunch(gs[seg], t)); }
      CK(hipEventRecord(ev[S], t)); CK(hipStreamWaitEvent(s, ev[S], 0));
    }
    mk(N);   // the "Adam" launch: needs both
    return 0;
  };
  for (int r = 0; r < 3; ++r) if (step()) return 1;
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  const int R = 20;
  for (int r = 0; r < R; ++r) if (step()) return 1;
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("mode %d N=%d M=%d S=%d main %d blk x %d it, side %d blk x %d it: %.1f us per step\n", mode, N, M, S, mblocks, miters, sblocks, siters, ms * 1000 / R);
  fflush(stdout);
  hipFree(a); hipFree(b); hipFree(c); hipFree(d);
  return 0;
}
int main() {
  // main kernels: 64 workgroups, ~5 us; side kernels: 1024 workgroups, ~10 us
  for (int S : {4, 8, 16}) {
    for (int mode : {0, 3, 1, 2}) if (run(mode, 240, 32, S, 64, 1500, 1024, 3000)) printf("mode %d failed\n", mode);
  }
  // side alone / main alone references
  run(0, 240, 1, 1, 64, 1500, 1024, 3000);
  return 0;
}
