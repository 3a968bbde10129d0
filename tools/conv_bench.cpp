// Micro-benchmark of one conv shape through the C ABI (used for kernel ablations; not part of the product).
//   hipcc -O2 tools/conv_bench.cpp -o gpurun_out/conv_bench -ldl ; ./conv_bench <libn3d.so> C D H W dil B iters
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../include/n3d.h"

typedef int (*conv_fwd_t)(const n3d_conv_geom*, const float*, int64_t, const float*, const float*, float*, int64_t, int, const float*, double*, void*, size_t, void*);
typedef size_t (*ws_t)(const n3d_conv_geom*);
typedef int (*rows_t)(const n3d_conv_geom*, int, int);
typedef const char* (*err_t)(void);

int main(int argc, char** argv) {
  if (argc < 9) { printf("usage: %s lib C D H W dil B iters [flags]\n", argv[0]); return 1; }
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 1; }
  conv_fwd_t conv = (conv_fwd_t)dlsym(h, "n3d_conv_fwd");
  ws_t wsb = (ws_t)dlsym(h, "n3d_conv_workspace_bytes");
  rows_t rowsf = (rows_t)dlsym(h, "n3d_conv_stats_rows");
  err_t lerr = (err_t)dlsym(h, "n3d_last_error");
  int C = atoi(argv[2]), D = atoi(argv[3]), H = atoi(argv[4]), W = atoi(argv[5]), dil = atoi(argv[6]), B = atoi(argv[7]), iters = atoi(argv[8]);
  int flags = argc > 9 ? atoi(argv[9]) : 0;
  int stride = argc > 10 ? atoi(argv[10]) : 1;
  int Do = (D + 2 * dil - dil * 2 - 1) / stride + 1, Ho = (H + 2 * dil - dil * 2 - 1) / stride + 1, Wo = (W + 2 * dil - dil * 2 - 1) / stride + 1;
  n3d_conv_geom g = {B, D, H, W, C, Do, Ho, Wo, C, 3, stride, dil, dil, 0};
  size_t n = (size_t)B * D * H * W * C;
  float *x, *y, *w, *bias; void* ws; double* stats;
  hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&w, C * C * 27 * 4); hipMalloc(&bias, C * 4);
  std::vector<float> hx(n); for (size_t i = 0; i < n; ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<float> hw(C * C * 27, 0.05f); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  hipMemset(bias, 0, C * 4);
  size_t wsn = wsb(&g); hipMalloc(&ws, wsn);
  int rows = rowsf(&g, 0, flags); hipMalloc(&stats, (size_t)B * (rows > 0 ? rows : 1) * C * 2 * 8);
  hipStream_t s; hipStreamCreate(&s);
  for (int i = 0; i < 3; ++i) { int r = conv(&g, x, C, w, bias, y, C, flags, nullptr, rows > 0 ? stats : nullptr, ws, wsn, s); if (r) { printf("err %d %s\n", r, lerr()); return 1; } }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, s);
  for (int i = 0; i < iters; ++i) conv(&g, x, C, w, bias, y, C, flags, nullptr, rows > 0 ? stats : nullptr, ws, wsn, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double us = ms * 1e3 / iters, fl = 2.0 * B * D * H * W * C * C * 27;
  printf("C=%d %dx%dx%d dil=%d B=%d: eager %.2f us/call (pack+conv)  %.2f TFLOP/s  rows=%d\n", C, D, H, W, dil, B, us, fl / us / 1e6, rows);
  // the same calls replayed from a HIP graph: no host launch cost, only GPU time + kernel boundaries
  hipGraph_t graph; hipGraphExec_t exec;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < iters; ++i) conv(&g, x, C, w, bias, y, C, flags, nullptr, rows > 0 ? stats : nullptr, ws, wsn, s);
  hipStreamEndCapture(s, &graph);
  hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  for (int r = 0; r < 2; ++r) hipGraphLaunch(exec, s);
  hipEventRecord(e0, s);
  for (int r = 0; r < 5; ++r) hipGraphLaunch(exec, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  us = ms * 1e3 / (iters * 5);
  printf("C=%d %dx%dx%d dil=%d B=%d: graph %.2f us/call (pack+conv)  %.2f TFLOP/s\n", C, D, H, W, dil, B, us, fl / us / 1e6);
  return 0;
}
