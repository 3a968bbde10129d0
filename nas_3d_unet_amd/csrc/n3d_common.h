// Shared host/device helpers for libn3d (gfx950 only).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "../../include/n3d.h"

namespace n3d {

void set_error(const char* fmt, ...);

#define N3D_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      n3d::set_error(__VA_ARGS__);      \
      return N3D_ERR_INVALID;           \
    }                                   \
  } while (0)

#define N3D_UNSUPPORTED(...)            \
  do {                                  \
    n3d::set_error(__VA_ARGS__);        \
    return N3D_ERR_UNSUPPORTED;         \
  } while (0)

#define N3D_LAUNCH_CHECK()                                                     \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      n3d::set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__,         \
                     hipGetErrorString(e__));                                  \
      return N3D_ERR_HIP;                                                      \
    }                                                                          \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Division by a launch-time constant without the ~40-200 instruction integer-division sequences hipcc emits
// for runtime divisors (index decoding used to dominate the small kernels).  Exact for 0 <= n < 2^31.
struct FastDiv {
  uint32_t d, m, s;
#ifdef __HIPCC__
  __host__ __device__
#endif
  FastDiv() : d(1), m(0x80000000u), s(0) {}
  explicit FastDiv(uint32_t dd) {
    d = dd ? dd : 1;
    s = 0;
    while ((1ull << s) < d) ++s;
    m = (uint32_t)(((1ull << (31 + s)) + d - 1) / d);
  }
#ifdef __HIPCC__
  __device__ __forceinline__ uint32_t div(uint32_t n) const { return (uint32_t)(((uint64_t)n * m) >> (31 + s)); }
  __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const { q = div(n); r = n - q * d; }
#endif
};
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------
// Element-wise mapping over a pitched NDHWC tensor with C % 4 == 0:
// a 256-thread block covers `vpb` voxels x `cpb` channel quads per iteration; thread t handles
// channel quad t % cpb of voxel t / cpb, so consecutive lanes touch consecutive 16-byte pieces.
// A block walks `iters` iterations = one "row" of `vpc` voxels; per-sample reductions emit one
// partial row per block (deterministic two-stage reduction, no atomics).
// ---------------------------------------------------------------------------------------------
struct EwMap {
  int cpb;      // channel quads per voxel
  int vpb;      // voxels per block iteration
  int iters;    // iterations per block
  int64_t vpc;  // voxels per block (chunk)
  int rows;     // blocks (partial rows) per sample
  FastDiv fcpb; // thread -> (voxel lane, channel quad) without a runtime division
};

static inline EwMap ew_map(int64_t N, int C) {
  EwMap m;
  m.cpb = C / 4;
  if (m.cpb < 1) m.cpb = 1;
  m.vpb = 256 / m.cpb;
  if (m.vpb < 1) m.vpb = 1;
  int64_t nit = cdiv(N, m.vpb);          // block-iterations per sample
  int64_t it = cdiv(nit, 1024);          // cap rows per sample at 1024
  constexpr int min_it = 4;
  // channel counts like the stems' 12: the class sums of a reduction go through ds_bpermute there, so fewer, longer rows pay
  // (epilogue backward of stem0 at 64^3: 43 -> 33 us over its three launches)
  constexpr int min_it_np2 = 16;
  int mi = (m.cpb & (m.cpb - 1)) ? min_it_np2 : min_it;
  // small tensors (the deep levels): a block's iterations are dependent memory round trips (load, use, store, next load), so a
  // launch lasts iterations x latency whatever the chip has idle -- one iteration for <= T1 block-iterations per sample, two up
  // to T2 (an N-term epilogue on 2 x 64 ch x 4^3: 7.9 -> ~4 us; search step 12.66 -> 12.45 ms with two everywhere below the
  // 64^3 level, while the 64^3 / 128^3 levels want the four: 4.53 -> 4.57 ms at 128^3 with two)
  constexpr int t1 = 8, t2 = 256;
  if (nit <= t1) mi = 1;
  else if (nit <= t2 && mi > 2) mi = 2;
  if (it < mi) it = nit < mi ? (nit < 1 ? 1 : nit) : mi;
  m.iters = (int)it;
  m.vpc = (int64_t)m.vpb * m.iters;
  m.rows = (int)cdiv(N, m.vpc);
  m.fcpb = FastDiv((uint32_t)m.cpb);
  return m;
}

#ifdef __HIPCC__
// Kernels of the dependent chain raise their waves' issue priority (s_setprio 0..3, 0 = the reset value): the weight-gradient
// kernels, which only ever run beside the chain on the side streams, stay at 0, so where both have waves on a SIMD the chain's
// instructions go first.  No effect when a kernel has the chip to itself.
#define N3D_PRIO 3
#define N3D_CHAIN_PRIO() __builtin_amdgcn_s_setprio(N3D_PRIO)
// ---- activation storage types.  fp32 is the reference's arithmetic; bf16 is the STORAGE format of the HBM-bound
// levels in the bf16 configuration (BASELINE configs[4]): every kernel converts on load / store and computes in fp32.
// A "quad" is four consecutive channels of one voxel: 16 bytes in fp32, 8 bytes in bf16.
typedef uint16_t bf16_t;
typedef __bf16 n3d_bf16x2 __attribute__((ext_vector_type(2)));
typedef float n3d_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                     __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u));
}
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {   // round-to-nearest-even (v_cvt_pk_bf16_f32)
  const n3d_f32x2 f = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, n3d_bf16x2));
}
__device__ __forceinline__ void st4(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, const float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __builtin_bit_cast(float, (uint32_t)*p << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)(pack_bf16x2(v, 0.f) & 0xffffu); }
#endif

#ifdef __HIPCC__
// ---- cross-lane sums.  A lone wave issues ~1 instruction per 4-5 cycles, so on the small tensors of the deep
// U-net levels kernel time IS the dynamic instruction count: reductions use DPP row operations and permlane swaps
// (1-2 VALU instructions per step and 32-bit half) instead of ds_bpermute sequences (~8 instructions + an LDS trip).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// lane i + lane i^16, and lane i + lane i^32, in every lane: gfx950's v_permlane16_swap / v_permlane32_swap exchange
// row pairs / wave halves between two VGPRs inside the VALU (a ds_swizzle or ds_bpermute pays an LDS round trip)
__device__ __forceinline__ float xsum16_f(float v) {
  const int x = __builtin_bit_cast(int, v);
  const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ float xsum32_f(float v) {
  const int x = __builtin_bit_cast(int, v);
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ double xsum16_d(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __builtin_bit_cast(double, ((long long)(int)rh[0] << 32) | (unsigned int)rl[0]) +
         __builtin_bit_cast(double, ((long long)(int)rh[1] << 32) | (unsigned int)rl[1]);
}
__device__ __forceinline__ double xsum32_d(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
  const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __builtin_bit_cast(double, ((long long)(int)rh[0] << 32) | (unsigned int)rl[0]) +
         __builtin_bit_cast(double, ((long long)(int)rh[1] << 32) | (unsigned int)rl[1]);
}
// every lane ends up with the sum over the lanes of its class (lane % cpb), cpb a power of two <= 64
__device__ __forceinline__ float wave_classsum_f(float v, int cpb) {
  if (cpb <= 1) v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  if (cpb <= 2) v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  if (cpb <= 4) v += dpp_f<0x124>(v);   // row_ror:4
  if (cpb <= 8) v += dpp_f<0x128>(v);   // row_ror:8
  if (cpb <= 16) v = xsum16_f(v);
  if (cpb <= 32) v = xsum32_f(v);
  return v;
}
// class sums of FOUR values at once, cpb a power of two <= 16.  v_permlane32_swap of (v0, v1) lays the two wave halves of v0 side by
// side in the low half of the register pair and those of v1 in the high half, so one add folds lane ^ 32 of both; the same for
// (v2, v3); a v_permlane16_swap of the two results folds lane ^ 16 of all four and leaves row r of the wave (lanes 16 r ..) with
// v[{0, 2, 1, 3}[r]]; DPP rotations finish inside the rows.  3 swaps + 3 adds + <= 4 DPP adds instead of 4 x (2 swaps + ...):
// the swaps are what these reductions wait for.  Lane l of row r returns the sum of v[{0,2,1,3}[r]] over the lanes of class l % cpb.
// (cpb is a template argument here and in class_dispatch16 below: with a run-time cpb every `if (cpb <= ..)` step of every value
// became a scalar branch -- 24 values x 6 branches in the epilogue-backward reduction, most of that kernel after its loads)
template <int CPB>
__device__ __forceinline__ float wave_classsum4_f(float v0, float v1, float v2, float v3) {
  const auto s01 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v0), __builtin_bit_cast(int, v1), false, false);
  const auto s23 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v2), __builtin_bit_cast(int, v3), false, false);
  const float ab = __builtin_bit_cast(float, (int)s01[0]) + __builtin_bit_cast(float, (int)s01[1]);
  const float cd = __builtin_bit_cast(float, (int)s23[0]) + __builtin_bit_cast(float, (int)s23[1]);
  const auto sx = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(int, ab), __builtin_bit_cast(int, cd), false, false);
  float x = __builtin_bit_cast(float, (int)sx[0]) + __builtin_bit_cast(float, (int)sx[1]);
  if (CPB <= 8) x += dpp_f<0x128>(x);   // row_ror:8
  if (CPB <= 4) x += dpp_f<0x124>(x);   // row_ror:4
  if (CPB <= 2) x += dpp_f<0x4E>(x);    // quad_perm [2,3,0,1]
  if (CPB <= 1) x += dpp_f<0xB1>(x);    // quad_perm [1,0,3,2]
  return x;
}
// f(std::integral_constant<int, cpb>) for cpb in {1, 2, 4, 8, 16}: one uniform branch, straight-line code behind it
template <typename F>
__device__ __forceinline__ void class_dispatch16(int cpb, F&& f) {
  switch (cpb) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 8: f(std::integral_constant<int, 8>{}); break;
    default: f(std::integral_constant<int, 16>{}); break;
  }
}
// the same for four doubles, whole-wave sums: row r of the wave returns the total of v[{0, 2, 1, 3}[r]] (6 swaps instead of 16)
__device__ __forceinline__ double wave_sum4_d(double v0, double v1, double v2, double v3) {
  auto lo = [](double v) { return (int)(__builtin_bit_cast(long long, v) & 0xffffffffll); };
  auto hi = [](double v) { return (int)(__builtin_bit_cast(long long, v) >> 32); };
  auto mk = [](int l, int h) { return __builtin_bit_cast(double, ((long long)h << 32) | (unsigned int)l); };
  const auto l01 = __builtin_amdgcn_permlane32_swap(lo(v0), lo(v1), false, false), h01 = __builtin_amdgcn_permlane32_swap(hi(v0), hi(v1), false, false);
  const auto l23 = __builtin_amdgcn_permlane32_swap(lo(v2), lo(v3), false, false), h23 = __builtin_amdgcn_permlane32_swap(hi(v2), hi(v3), false, false);
  const double ab = mk((int)l01[0], (int)h01[0]) + mk((int)l01[1], (int)h01[1]);
  const double cd = mk((int)l23[0], (int)h23[0]) + mk((int)l23[1], (int)h23[1]);
  const auto lx = __builtin_amdgcn_permlane16_swap(lo(ab), lo(cd), false, false), hx = __builtin_amdgcn_permlane16_swap(hi(ab), hi(cd), false, false);
  double x = mk((int)lx[0], (int)hx[0]) + mk((int)lx[1], (int)hx[1]);
  x += dpp_d<0x128>(x);   // row_ror:8
  x += dpp_d<0x124>(x);   // row_ror:4
  x += dpp_d<0x4E>(x);    // quad_perm [2,3,0,1]
  x += dpp_d<0xB1>(x);    // quad_perm [1,0,3,2]
  return x;
}
// the value index (0..3) that wave_classsum4_f leaves in this lane's row
__device__ __forceinline__ int classsum4_sel(int lane) { return ((lane >> 4) & 1) * 2 + (lane >> 5); }
__device__ __forceinline__ double wave_classsum_d(double v, int cpb) {
  if (cpb <= 1) v += dpp_d<0xB1>(v);
  if (cpb <= 2) v += dpp_d<0x4E>(v);
  if (cpb <= 4) v += dpp_d<0x124>(v);
  if (cpb <= 8) v += dpp_d<0x128>(v);
  if (cpb <= 16) v = xsum16_d(v);
  if (cpb <= 32) v = xsum32_d(v);
  return v;
}
template <int CPB>
__device__ __forceinline__ double wave_classsum_d_t(double v) {
  if (CPB <= 1) v += dpp_d<0xB1>(v);
  if (CPB <= 2) v += dpp_d<0x4E>(v);
  if (CPB <= 4) v += dpp_d<0x124>(v);
  if (CPB <= 8) v += dpp_d<0x128>(v);
  if (CPB <= 16) v = xsum16_d(v);
  if (CPB <= 32) v = xsum32_d(v);
  return v;
}
// NV values at once behind ONE uniform switch on the width: with a run-time width every step of every value is a scalar branch
// (six per value), and the values' chains cannot interleave across the branches
template <int NV>
__device__ __forceinline__ void wave_classsum_dn(double (&v)[NV], int cpb) {
#define N3D_CS_CASE(W) case W: _Pragma("unroll") for (int q = 0; q < NV; ++q) v[q] = wave_classsum_d_t<W>(v[q]); break;
  switch (cpb) {
    N3D_CS_CASE(4) N3D_CS_CASE(8) N3D_CS_CASE(16) N3D_CS_CASE(32)
    case 64: break;
    default:
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = wave_classsum_d(v[q], cpb);
  }
#undef N3D_CS_CASE
}
__device__ __forceinline__ bool is_pow2(int x) { return (x & (x - 1)) == 0; }
// every lane ends up with the sum over its aligned group of cg consecutive lanes (cg = 1, 2, 4, 8 or 16): DPP only
__device__ __forceinline__ double wave_groupsum_d(double v, int cg) {
  if (cg >= 2) v += dpp_d<0xB1>(v);    // quad_perm [1,0,3,2]
  if (cg >= 4) v += dpp_d<0x4E>(v);    // quad_perm [2,3,0,1]
  if (cg >= 8) v += dpp_d<0x141>(v);   // row_half_mirror: lane i <-> 7 - i of each 8
  if (cg >= 16) v += dpp_d<0x140>(v);  // row_mirror: lane i <-> 15 - i of each 16
  return v;
}

template <int CG>
__device__ __forceinline__ double wave_groupsum_d_t(double v) {
  if (CG >= 2) v += dpp_d<0xB1>(v);
  if (CG >= 4) v += dpp_d<0x4E>(v);
  if (CG >= 8) v += dpp_d<0x141>(v);
  if (CG >= 16) v += dpp_d<0x140>(v);
  return v;
}
template <int NV>
__device__ __forceinline__ void wave_groupsum_dn(double (&v)[NV], int cg) {
#define N3D_GS_CASE(W) case W: _Pragma("unroll") for (int q = 0; q < NV; ++q) v[q] = wave_groupsum_d_t<W>(v[q]); break;
  switch (cg) {
    N3D_GS_CASE(4) N3D_GS_CASE(8) N3D_GS_CASE(16)
    default:
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q] = wave_groupsum_d(v[q], cg);
  }
#undef N3D_GS_CASE
}

// sum over the lanes of a wave that share (lane % cpb); result valid in lanes < cpb (any cpb <= 64)
__device__ __forceinline__ double wave_sum_strided(double v, int cpb) {
  if (is_pow2(cpb)) return wave_classsum_d(v, cpb);
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const int off = s * cpb;
    if (off < 64) {
      double o = __shfl_down(v, off, 64);
      if (lane + off < 64) v += o;
    }
  }
  return v;
}
__device__ __forceinline__ float wave_sum_strided_f(float v, int cpb) {
  if (is_pow2(cpb)) return wave_classsum_f(v, cpb);
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const int off = s * cpb;
    if (off < 64) {
      float o = __shfl_down(v, off, 64);
      if (lane + off < 64) v += o;
    }
  }
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) { return wave_classsum_f(v, 1); }
__device__ __forceinline__ double wave_sum_d(double v) { return wave_classsum_d(v, 1); }
#endif

// data-gradient extras of one conv in a paired MFMA launch (conv_mfma.hip, mfma_conv_pair_try): ReLU mask source of the
// conv input (and its pitch), per-(b,c) gate applied to the result
struct PairExtras { const float* relu_src; int64_t rld; const float* out_gate; };

}  // namespace n3d
