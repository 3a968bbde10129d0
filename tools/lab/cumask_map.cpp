// Which physical compute units does bit i of a hipExtStreamCreateWithCUMask mask select on MI355X?  Each workgroup records
// (XCC id, SE id, CU id) from the hardware registers; the host prints the set seen per mask.
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_map.cpp -o tools/build/cumask_map && tools/build/cumask_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void where(unsigned* out, int iters) {
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) a = fmaf(a, 1.0001f, 1e-7f);
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hw | (a == 123.f);
  }
}
static int show(const char* what, const std::vector<uint32_t>& mask) {
  hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  const int wgs = 4096;
  unsigned* d; CK(hipMalloc(&d, wgs * 8));
  std::vector<unsigned> h(wgs * 2);
  hipLaunchKernelGGL(where, dim3(wgs), dim3(256), 0, s, d, 3000);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost));
  std::map<int, std::set<int>> per;   // xcc -> {se*16 + cu}
  for (int i = 0; i < wgs; ++i) {
    const unsigned xcc = h[2 * i] & 15, hw = h[2 * i + 1];
    per[xcc].insert(((hw >> 13) & 7) * 16 + ((hw >> 8) & 15));
  }
  printf("%s:", what);
  int total = 0;
  for (auto& kv : per) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
  printf("  (%d CUs)\n", total);
  hipFree(d); hipStreamDestroy(s);
  return 0;
}
int main() {
  std::vector<uint32_t> all(8, 0xffffffffu);
  show("all 256 bits", all);
  for (int w = 0; w < 8; ++w) { std::vector<uint32_t> m(8, 0); m[w] = 0xffffffffu; char nm[64]; snprintf(nm, 64, "word %d only", w); show(nm, m); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0xffu; show("bits 0-7", m); }
  { std::vector<uint32_t> m(8, 0); m[0] = 0xff00u; show("bits 8-15", m); }
  { std::vector<uint32_t> m(8, 0x00ffffffu); show("low 24 bits of every word", m); }
  { std::vector<uint32_t> m(8, 0); for (int i = 0; i < 192; ++i) m[i / 32] |= 1u << (i % 32); show("bits 0-191", m); }
  { std::vector<uint32_t> m(8, 0x55555555u); show("every second bit", m); }
  { std::vector<uint32_t> m(8, 0xffffffffu); m[7] = 0; show("bits 0-223", m); }
  { std::vector<uint32_t> m(8, 0xffffffffu); for (int i = 0; i < 8; ++i) m[0] &= ~(1u << i); show("all but bits 0-7", m); }
  { std::vector<uint32_t> m(8, 0xffffffffu); for (int i = 0; i < 64; ++i) m[i / 32] &= ~(1u << (i % 32)); show("all but bits 0-63", m); }
  return 0;
}
