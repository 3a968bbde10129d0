// Micro-benchmark of one conv shape through the C ABI (used for kernel ablations; not part of the product).
//   hipcc -O2 tools/conv_bench.cpp -o gpurun_out/conv_bench -ldl ; ./conv_bench <libn3d.so> C D H W dil B iters
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
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
  typedef int (*stamps_t)(unsigned long long*, int);
  stamps_t stf = (stamps_t)dlsym(h, "n3d_debug_vox_stamps");
  if (stf) {  // VOX_STAMP build: phase stamps of the last launch
    hipStreamSynchronize(s);
    const int nwg = B * (W / 16) * (H / 4) * (D / (argc > 11 ? atoi(argv[11]) : 4));
    const int n = nwg < 8192 ? nwg : 8192;
    std::vector<unsigned long long> st((size_t)n * 8);
    stf(st.data(), n * 8);
    unsigned long long r0 = ~0ull, r1 = 0;
    for (int i = 0; i < n; ++i) { if (st[i * 8 + 6] < r0) r0 = st[i * 8 + 6]; if (st[i * 8 + 6] > r1) r1 = st[i * 8 + 6]; }
    printf("stamps: %d workgroups, start spread %.2f us (100 MHz clock)\n", n, (r1 - r0) / 100.0);
    const char* nm[5] = {"loads landed", "LDS written", "MFMA done", "stats done", "stores done"};
    for (int k = 0; k < 5; ++k) {
      std::vector<double> d(n);
      for (int i = 0; i < n; ++i) d[i] = (double)(st[i * 8 + k + 1] - st[i * 8 + k]);
      std::sort(d.begin(), d.end());
      printf("  phase %d (%s): p10 %.0f  p50 %.0f  p90 %.0f  max %.0f ticks\n", k, nm[k], d[n / 10], d[n / 2], d[n * 9 / 10], d[n - 1]);
    }
    std::vector<double> d(n);
    for (int i = 0; i < n; ++i) d[i] = (double)(st[i * 8 + 5] - st[i * 8 + 0]);
    std::sort(d.begin(), d.end());
    printf("  total: p10 %.0f p50 %.0f p90 %.0f max %.0f ticks\n", d[n / 10], d[n / 2], d[n * 9 / 10], d[n - 1]);
    {  // placement: waves per SIMD / per CU (HW_ID: simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID [3:0])
      std::vector<int> per_simd(8 * 8 * 2 * 16 * 4, 0), per_cu(8 * 8 * 2 * 16, 0);
      for (int i = 0; i < n; ++i) {
        const unsigned hw = (unsigned)st[i * 8 + 7], xcc = (unsigned)(st[i * 8 + 7] >> 32) & 15;
        const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const int cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu;
        per_cu[cuid]++; per_simd[cuid * 4 + simd]++;
      }
      int hs[16] = {0}, hc[32] = {0}, ncu = 0;
      for (size_t k = 0; k < per_cu.size(); ++k) if (per_cu[k]) { ++ncu; hc[per_cu[k] < 31 ? per_cu[k] : 31]++; for (int q = 0; q < 4; ++q) hs[per_simd[k * 4 + q] < 15 ? per_simd[k * 4 + q] : 15]++; }
      printf("  CUs used %d; waves per CU histogram:", ncu); for (int k = 0; k < 32; ++k) if (hc[k]) printf(" %d:%d", k, hc[k]);
      printf("\n  waves per SIMD histogram:"); for (int k = 0; k < 16; ++k) if (hs[k]) printf(" %d:%d", k, hs[k]); printf("\n");
    }
    stamps_t stf2 = (stamps_t)dlsym(h, "n3d_debug_vox_stamps2");
    if (stf2) {
      std::vector<unsigned long long> s2((size_t)n * 8);
      stf2(s2.data(), n * 8);
      printf("  MFMA issue per input plane (ticks from LDS-ready, median):");
      for (int k = 0; k < 6; ++k) {
        std::vector<double> d(n);
        for (int i = 0; i < n; ++i) d[i] = (double)(s2[i * 8 + k] - (k ? s2[i * 8 + k - 1] : st[i * 8 + 2]));
        std::sort(d.begin(), d.end());
        printf(" %.0f", d[n / 2]);
      }
      printf("\n");
    }
    std::vector<double> st0(n);
    for (int i = 0; i < n; ++i) st0[i] = (st[i * 8 + 6] - r0) / 100.0;
    std::sort(st0.begin(), st0.end());
    printf("  start offsets us: p10 %.2f p50 %.2f p90 %.2f max %.2f\n", st0[n / 10], st0[n / 2], st0[n * 9 / 10], st0[n - 1]);
  }
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
