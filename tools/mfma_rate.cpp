// Issue-rate probe for v_mfma_f32_4x4x1_16b_f32 (and 16x16x4 f32): ticks per MFMA with 1..3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CHAINS, int KIND>
__global__ __launch_bounds__(64) void probe(float* out, unsigned long long* ticks, int iters) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  const unsigned long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
        else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int CHAINS, int KIND>
void run(int wps, const char* nm) {
  const int nb = 256 * 4 * wps, iters = 200;
  float* out; unsigned long long* tk; hipMalloc(&out, nb * 64 * 4); hipMalloc(&tk, nb * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<CHAINS, KIND><<<nb, 64>>>(out, tk, iters);
  hipEventRecord(e0); probe<CHAINS, KIND><<<nb, 64>>>(out, tk, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[4096]; hipMemcpy(h, tk, (nb < 4096 ? nb : 4096) * 8, hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < (nb < 4096 ? nb : 4096); ++i) m += h[i]; m /= (nb < 4096 ? nb : 4096);
  const double nm_ = (double)iters * 8 * CHAINS;
  const double flop = (KIND == 0 ? 512.0 : 2048.0) * nm_ * nb;
  printf("%s chains=%d waves/SIMD=%d: %.2f ticks per MFMA per wave (x%d waves = %.2f per SIMD-MFMA), wall %.1f us, %.1f TFLOP/s, tick rate %.2f GHz\n", nm, CHAINS, wps,
         m / nm_, wps, m / nm_ / wps, ms * 1e3, flop / (ms * 1e-3) / 1e12, m / (ms * 1e-3) / 1e9);
  hipFree(out); hipFree(tk);
}
int main() {
  run<1, 0>(1, "4x4x1"); run<2, 0>(1, "4x4x1"); run<4, 0>(1, "4x4x1"); run<4, 0>(2, "4x4x1"); run<4, 0>(3, "4x4x1"); run<1, 0>(2, "4x4x1");
  run<4, 1>(1, "16x16x4"); run<4, 1>(2, "16x16x4");
  return 0;
}
