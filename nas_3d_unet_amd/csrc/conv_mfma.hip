// MFMA implicit-GEMM convolution kernels (fp32 in / fp32 accumulate, exact fp32: v_mfma_f32_16x16x4_f32).
//
// gemm16 family -- channel counts that are multiples of 16 (the deep U-net levels, C = 16/32/64, and the
// wide 1x1x1 cell-preprocess convs): GEMM view  D[voxel][cd] = sum_{tap,cs} S[map(voxel,tap)][cs] * W[tap][cs][cd]
//   M = voxels (16 per MFMA tile, lane&15), N = cd (16 per tile), K = taps * Cs (4 per MFMA).
//   A operand: lane (m = lane&15, kk = lane>>4) loads ONE float4 = S[voxel m][16*c16 + 4*kk + j], j = 0..3,
//              i.e. 16 voxels x 64 contiguous bytes; element j feeds MFMA step j (k-slot kk <-> channel 16*c16+4*kk+j).
//   B operand: packed weights Wp[tap][c16][kk][cd][j] -> one float4 per lane, 1 KiB contiguous per wave.
// These levels have only 16..8192 voxels, so the kernels are latency-bound: the K loop (taps x Cs/16) is split
// over the 4 waves of a workgroup (KSPLIT = 4) and reduced through LDS when there are few tiles.
//
// The same kernel is the data-gradient / transposed-conv kernel through the gather map
//   src coordinate = (dst*sn + off + tap*dt) / den   (valid iff divisible and in range).
#include <stdlib.h>
#include "n3d_common.h"
#include <type_traits>
// cache policy of the weight-gradient kernels' LDS-DMA loads (cpol bits: 1 = sc0, 2 = nt, 16 = sc1).  These kernels run on the side
// stream next to the backward chain and stream 8-25 MB tensors through the L2s that hold the chain's working set.
#define N3D_WGRAD_AUX 0

namespace n3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// N3D_MM_BF16 (the bf16 configuration, BASELINE configs[4]): the C >= 16 levels keep fp32 STORAGE, but their matrix products round both
// operands to bfloat16 in registers and run on v_mfma_f32_16x16x16_bf16 (fp32 accumulate): lane (m, kk) of the fp32 kernels holds the
// four K values 4 kk + j of its row as a float4 -- exactly the bf16x4 operand of the 16-deep instruction -- so four
// v_mfma_f32_16x16x4_f32 (128 cycles) become two v_cvt_pk_bf16_f32 per operand and ONE 16-cycle MFMA.  The fp32 configuration never
// sets the flag: its arithmetic stays exact fp32.
typedef short mm_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ mm_bf16x4 mm_cvt4(float a, float b, float c, float d) {
  const uint2 p = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
  return __builtin_bit_cast(mm_bf16x4, p);
}
__device__ __forceinline__ mm_bf16x4 mm_cvt4(const float4 v) { return mm_cvt4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ f32x4 mm_bf16(const mm_bf16x4 a, const mm_bf16x4 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

struct MfArgs {
  const float* src; int64_t sld; int Ds, Hs, Ws, Cs;
  float* dst; int64_t dld; int Dd, Hd, Wd, Cd;
  const float* wp;
  const float* bias;
  int k, sn, off, dt, den, flags, B;
  const float* in_gate;
  const float* relu_src; int64_t rld;
  const float* out_gate;
  double* stats;
  int rows_per_sample;
  FastDiv fNd, fWd, fHd, fC16;
};

// Wp[tap][c16][kk][cd][j] <- w native (Co, Ci, taps); transpose=0: cs=ci, cd=co; transpose=1: cs=co, cd=ci
__global__ void pack16_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int taps, int transpose) {
  const int Cs = transpose ? Co : Ci, Cd = transpose ? Ci : Co;
  const int total = taps * Cs * Cd;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int j = i & 3;
  const int cd = (i >> 2) % Cd;
  const int rest = (i >> 2) / Cd;  // (tap*(Cs/16) + c16)*4 + kk
  const int kk = rest & 3, c16 = (rest >> 2) % (Cs / 16), tap = (rest >> 2) / (Cs / 16);
  const int cs = c16 * 16 + kk * 4 + j;
  const int co = transpose ? cs : cd, ci = transpose ? cd : cs;
  wp[i] = w[((int64_t)co * Ci + ci) * taps + tap];
}

template <int MT, int NT, int KSPLIT, bool BF = false>
__device__ __forceinline__ void gemm16_body_t(const MfArgs& a, const int bx, const int by, float* lds) {
  N3D_CHAIN_PRIO();
  // the wave index as a SCALAR: the K-slice a wave owns (tap, channel block, their offsets) is then computed on the scalar unit
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, kk = lane >> 4;
  const int64_t Nd = (int64_t)a.Dd * a.Hd * a.Wd;
  const int64_t Mtot = (int64_t)a.B * Nd;
  constexpr int ROWS_PER_WAVE = 16 * MT;
  constexpr int ROWS_PER_BLOCK = ROWS_PER_WAVE * (KSPLIT == 1 ? 4 : 1);
  const int64_t row0 = (int64_t)bx * ROWS_PER_BLOCK + (KSPLIT == 1 ? wave * ROWS_PER_WAVE : 0);
  const int n0 = by * 16 * NT;

  // A-side rows of this lane (voxel m of each M tile)
  int rb[MT], rd[MT], rh[MT], rw[MT];
  bool rvalid[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int64_t i = row0 + t * 16 + m;
    rvalid[t] = i < Mtot;
    const uint32_t ii = rvalid[t] ? (uint32_t)i : 0u;
    uint32_t ub, uv, q1, uw, ud, uh;
    a.fNd.divmod(ii, ub, uv);
    a.fWd.divmod(uv, q1, uw);
    a.fHd.divmod(q1, ud, uh);
    rb[t] = (int)ub; rw[t] = (int)uw; rh[t] = (int)uh; rd[t] = (int)ud;
  }
  f32x4 acc[MT][NT], acc2[MT][NT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) { acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

  // epilogue operands of the writer wave (bias, ReLU mask source, output gate, previous output) are requested
  // here, ahead of the K loop, so the epilogue never waits on memory
  const bool writer = (KSPLIT == 1) || (wave == 0);
  const bool accum = a.flags & N3D_ACCUMULATE;
  float e_bias[NT], e_relu[MT][4][NT], e_gate[MT][4][NT], e_prev[MT][4][NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) e_bias[n] = 0.f;
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int n = 0; n < NT; ++n) { e_relu[t][r][n] = 1.f; e_gate[t][r][n] = 1.f; e_prev[t][r][n] = 0.f; }
  if (writer) {
#pragma unroll
    for (int n = 0; n < NT; ++n) if (a.bias) e_bias[n] = a.bias[n0 + n * 16 + m];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t i = row0 + t * 16 + kk * 4 + r;
        if (i < Mtot) {
          const int b = (int)a.fNd.div((uint32_t)i);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const int c = n0 + n * 16 + m;
            if (a.relu_src) e_relu[t][r][n] = a.relu_src[i * a.rld + c];
            if (a.out_gate) e_gate[t][r][n] = a.out_gate[(int64_t)b * a.Cd + c];
            if (accum) e_prev[t][r][n] = a.dst[i * a.dld + c];
          }
        }
      }
  }

  const int k = a.k, taps = k * k * k;
  const int c16n = a.Cs >> 4;
  const int ngroups = taps * c16n;
  const bool relu_in = a.flags & N3D_RELU_IN;
  // K loop, software-pipelined by hand: the operands of group g+step are requested before the MFMAs of group g
  // issue, so each wave keeps one group of global loads in flight behind its matrix work.
  //
  // Operand addressing (round 3).  Sixteen waves share four SIMDs here, so what a wave spends before its loads are out is VALU
  // issue slots: the per-group, per-lane address arithmetic (tap offsets, 64-bit pointer, range tests, zero select) was ~40 VALU
  // instructions and the load phase 5 of this kernel's 8 us (round-3 phase stamps: profiles/r03_gemm16_anatomy.log).  The voxel index is linear in (lane part) +
  // (tap part) -- also for the stride-2 data gradient, where the source voxel is (row + tap) / 2 and exists only when row and tap
  // agree in parity: then (row + tap) / 2 = (row >> 1) + ((tap + (tap & 1)) >> 1) -- so the lane part is a byte offset computed ONCE
  // (voffA), the tap part moves the base of a BUFFER resource on the scalar unit, and a lane whose tap falls outside the volume
  // (or whose row does not exist) presents an offset beyond the resource's range: the buffer load returns zeros for it, no select.
  // Per group and lane that leaves three adds, three compares and one select.
  const bool den2 = a.den == 2;
  const float relu_floor = relu_in ? 0.f : -INFINITY;
  int pd[MT], ph[MT], pw[MT], lpar[MT];
  uint32_t voffA[MT];
  constexpr uint32_t OOB = 0x80000000u;           // >= num_records of the resources below, with or without the scalar offset
  constexpr int RSRC_FLAGS = 0x00020000;          // gfx9 raw buffer: 32-bit data format, no swizzle
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    pd[t] = den2 ? rd[t] >> 1 : rd[t] * a.sn;
    ph[t] = den2 ? rh[t] >> 1 : rh[t] * a.sn;
    pw[t] = den2 ? rw[t] >> 1 : rw[t] * a.sn;
    lpar[t] = den2 ? ((rd[t] & 1) | ((rh[t] & 1) << 1) | ((rw[t] & 1) << 2)) : 0;
    const int vox = ((rb[t] * a.Ds + pd[t]) * a.Hs + ph[t]) * a.Ws + pw[t];
    voffA[t] = rvalid[t] ? (uint32_t)((vox * (int)a.sld + kk * 4) * 4) : OOB;
  }
  const uint32_t voffB = (uint32_t)((kk * a.Cd + n0 + m) * 16);
  // The tap part, ONE table per workgroup: thread g works out group g (tap = g / (Cs/16), its three offsets, the stride-2 halves and
  // parities, the element offset of the tap in the source, the byte offset of the group in the packed weights) -- ~100 instructions
  // once instead of on the scalar unit of every wave for every group, where 16 waves x 8 groups x ~100 scalar instructions on the
  // compute unit's ONE scalar ALU were the longest part of the kernel.  Entry: { source byte offset + BIAS, weight byte offset,
  // sd | sh << 8 | sw << 16 | parity << 24 }.  The source resource starts BIAS bytes in front of the tensor so that the scalar offset
  // of a tap that reaches backwards stays non-negative; only lanes whose voxel exists add it.
  __shared__ int gtab[256 * 4];
  const int bias_el = ((4 * a.Hs + 4) * a.Ws + 4) * (int)a.sld;      // |tap offset| <= 4 voxels per axis (pad, dilation <= 2)
  for (int g = threadIdx.x; g < ngroups; g += blockDim.x) {
    const int tap = (int)a.fC16.div((uint32_t)g), c16 = g - tap * c16n;
    // k is 1 or 3: constant divisors
    const int kd = (k == 3) ? tap / 9 : 0, kh = (k == 3) ? (tap % 9) / 3 : 0, kw = (k == 3) ? tap % 3 : 0;
    const int od = a.off + kd * a.dt, oh = a.off + kh * a.dt, ow = a.off + kw * a.dt;
    // tap part of the voxel coordinate: the offset itself, or its upper half for the stride-2 data gradient
    const int sd = den2 ? (od + (od & 1)) >> 1 : od, sh = den2 ? (oh + (oh & 1)) >> 1 : oh, sw = den2 ? (ow + (ow & 1)) >> 1 : ow;
    const int spar = den2 ? ((od & 1) | ((oh & 1) << 1) | ((ow & 1) << 2)) : 0;
    const int soff = ((sd * a.Hs + sh) * a.Ws + sw) * (int)a.sld + c16 * 16;
    int4 e;
    e.x = (soff + bias_el) * 4;
    e.y = g * 4 * a.Cd * 16;
    e.z = (sd & 0xff) | ((sh & 0xff) << 8) | ((sw & 0xff) << 16) | (spar << 24);
    e.w = 0;
    *reinterpret_cast<int4*>(&gtab[g * 4]) = e;
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src) - bias_el, 0, 0x7fffffff, RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rwt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, 0x7fffffff, RSRC_FLAGS);
  typedef int v4i_t __attribute__((ext_vector_type(4)));
  // GATE (a std::bool_constant): the per-(sample, channel) input gate of the SE primitives; the run-time test is made once, around
  // the load phase (a branch per group would make every group its own basic block)
  auto load_group = [&](auto gate_c, int g, float4 (&av)[MT], float4 (&bv)[NT], float4 (&gv)[MT]) {
    constexpr bool GATE = decltype(gate_c)::value;
    const int4 e = *reinterpret_cast<const int4*>(&gtab[g * 4]);      // one address for the whole wave
    const int soffA = __builtin_amdgcn_readfirstlane(e.x), soffB = __builtin_amdgcn_readfirstlane(e.y);
    const int pk = e.z;
    const int sd = (int)(signed char)(pk & 0xff), sh = (int)(signed char)((pk >> 8) & 0xff), sw = (int)(signed char)((pk >> 16) & 0xff);
    const int spar = pk >> 24;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int nd = pd[t] + sd, nh = ph[t] + sh, nw = pw[t] + sw;
      // unsigned compares fold the two-sided range tests (bitwise &: a short-circuit && becomes control flow)
      const bool ok = ((unsigned)nd < (unsigned)a.Ds) & ((unsigned)nh < (unsigned)a.Hs) & ((unsigned)nw < (unsigned)a.Ws) & (lpar[t] == spar);
      const v4i_t raw = __builtin_amdgcn_raw_buffer_load_b128(ra, (int)(ok ? voffA[t] : OOB), soffA, 0);
      float4 v = __builtin_bit_cast(float4, raw);
      // (the input ReLU and the gate are applied in mfma_group: anything that touches the loaded value HERE makes the compiler
      // wait for it between the groups' requests -- s_waitcnt vmcnt(2) in the middle of the load phase -- and the round trips of
      // a wave's groups then follow each other instead of overlapping)
      av[t] = v;
      if constexpr (GATE) {
        const int c16 = g % c16n;
        gv[t] = *reinterpret_cast<const float4*>(a.in_gate + (int64_t)rb[t] * a.Cs + c16 * 16 + kk * 4);
      }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      bv[n] = __builtin_bit_cast(float4, (v4i_t)__builtin_amdgcn_raw_buffer_load_b128(rwt, (int)(voffB + n * 256), soffB, 0));
    }
  };
  auto mfma_group = [&](auto gate_c, const float4 (&av0)[MT], const float4 (&bv)[NT], const float4 (&gv)[MT]) {
    constexpr bool GATE = decltype(gate_c)::value;
    float4 av[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      // (unconditional fmax: a branch on the ReLU flag per group would make every group its own basic block)
      float4 v = av0[t];
      v.x = fmaxf(v.x, relu_floor); v.y = fmaxf(v.y, relu_floor); v.z = fmaxf(v.z, relu_floor); v.w = fmaxf(v.w, relu_floor);
      if constexpr (GATE) { v.x *= gv[t].x; v.y *= gv[t].y; v.z *= gv[t].z; v.w *= gv[t].w; }
      av[t] = v;
    }
    if constexpr (BF) {
      // N3D_MM_BF16: one 16-deep bf16 MFMA per tile and group (operands rounded in registers)
      mm_bf16x4 ab[MT], bb[NT];
#pragma unroll
      for (int t = 0; t < MT; ++t) ab[t] = mm_cvt4(av[t]);
#pragma unroll
      for (int n = 0; n < NT; ++n) bb[n] = mm_cvt4(bv[n]);
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = mm_bf16(ab[t], bb[n], acc[t][n]);
      return;
    }
    // two accumulator chains per tile (x,z / y,w): a dependent 16x16x4 MFMA has 40 cycles latency vs 32 issue
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].x, bv[n].x, acc[t][n], 0, 0, 0);
        acc2[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].y, bv[n].y, acc2[t][n], 0, 0, 0);
      }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].z, bv[n].z, acc[t][n], 0, 0, 0);
        acc2[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].w, bv[n].w, acc2[t][n], 0, 0, 0);
      }
  };
  if ((KSPLIT == 16 || (KSPLIT == 4 && MT == 1 && NT == 1)) && ngroups <= KSPLIT * 8) {
    // tiny GEMM: every operand this wave will ever need is requested up front (<= 8 groups, 16 float4 per lane),
    // so the whole K loop costs one memory round trip.  (KSPLIT == 4: the 16-channel convs of the 16^3 level, 27 groups over four
    // waves -- the two-buffer walk below took seven round trips there)
    // The number of groups per wave is dispatched to a compile-time count (2 / 4 / 8): the loop used to run eight times with the
    // groups beyond the wave's share clamped to the last one -- loads and address arithmetic issued for nothing, 1.7 useful
    // iterations of 8 for a 16-channel conv, 3.4 for 32 channels -- and issue slots are what this kernel is short of.
    auto tiny = [&](auto ni_c) {
      constexpr int NI = decltype(ni_c)::value;
      float4 avs[NI][MT], bvs[NI][NT];
      if (a.in_gate) {
        float4 gvs[NI][MT];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int g = wave + i * KSPLIT;
          load_group(std::true_type{}, g < ngroups ? g : ngroups - 1, avs[i], bvs[i], gvs[i]);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
          if (wave + i * KSPLIT < ngroups) mfma_group(std::true_type{}, avs[i], bvs[i], gvs[i]);
      } else {
        float4 gnone[MT];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int g = wave + i * KSPLIT;
          load_group(std::false_type{}, g < ngroups ? g : ngroups - 1, avs[i], bvs[i], gnone);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
          if (wave + i * KSPLIT < ngroups) mfma_group(std::false_type{}, avs[i], bvs[i], gnone);
      }
    };
    if (ngroups <= KSPLIT * 2) tiny(std::integral_constant<int, 2>{});
    else if (ngroups <= KSPLIT * 4) tiny(std::integral_constant<int, 4>{});
    else tiny(std::integral_constant<int, 8>{});
  } else {
    const int g0 = (KSPLIT > 1 ? wave : 0), step = (KSPLIT > 1 ? KSPLIT : 1);
    float4 avA[MT], bvA[NT], avB[MT], bvB[NT], gvA[MT], gvB[MT];
    auto walk = [&](auto gate_c) {
      if (g0 < ngroups) load_group(gate_c, g0, avA, bvA, gvA);
      for (int g = g0; g < ngroups; g += 2 * step) {
        const bool hasB = g + step < ngroups;
        if (hasB) load_group(gate_c, g + step, avB, bvB, gvB);
        mfma_group(gate_c, avA, bvA, gvA);
        if (g + 2 * step < ngroups) load_group(gate_c, g + 2 * step, avA, bvA, gvA);
        if (hasB) mfma_group(gate_c, avB, bvB, gvB);
      }
    };
    if (a.in_gate) walk(std::true_type{}); else walk(std::false_type{});
  }
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] += acc2[t][n];

  if (KSPLIT > 1) {
    // reduce the K-slices through LDS into wave 0
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);
    if (wave > 0) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) l4[(((wave - 1) * MT + t) * NT + n) * 64 + lane] = acc[t][n];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int w = 0; w < KSPLIT - 1; ++w)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[t][n] += l4[((w * MT + t) * NT + n) * 64 + lane];
    }
  }

  // ---- epilogue: D layout -> lane holds column n = lane&15, rows 4*(lane>>4) + r
  float csum[NT], csq[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) { csum[n] = 0.f; csq[n] = 0.f; }
  if (writer) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t i = row0 + t * 16 + kk * 4 + r;
        if (i >= Mtot) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int c = n0 + n * 16 + m;
          float v = acc[t][n][r] + e_bias[n];
          if (!(e_relu[t][r][n] > 0.f)) v = 0.f;
          v = v * e_gate[t][r][n] + e_prev[t][r][n];
          a.dst[i * a.dld + c] = v;
          csum[n] += v; csq[n] = fmaf(v, v, csq[n]);
        }
      }
    }
  }
  if (a.stats && KSPLIT > 1) {
    // only wave 0 holds data: its lanes kk == 0 write the block's partial row directly (no LDS, no barrier)
    if (wave == 0) {
      const uint32_t first = (uint32_t)bx * ROWS_PER_BLOCK;
      uint32_t ub2, ur2;
      a.fNd.divmod(first, ub2, ur2);
      const int row = (int)(ur2 / ROWS_PER_BLOCK);
      // 8 voxels per sample (the 2^3 level): rows 0-7 (lanes kk 0/1) are one sample, rows 8-15 (kk 2/3) the next
      const bool two = Nd * 2 == ROWS_PER_BLOCK;
      // the sum and the sum of squares of a column are folded over the four row groups TOGETHER (fp32, as the other conv kernels'
      // statistics epilogues): one v_permlane swap pair instead of eight
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (two) {
          // fold lane ^ 16 only: rows 0 / 1 = sum / sum of squares of the first sample, rows 2 / 3 of the second
          const auto sx = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(int, csum[n]), __builtin_bit_cast(int, csq[n]), false, false);
          const float x = __builtin_bit_cast(float, (int)sx[0]) + __builtin_bit_cast(float, (int)sx[1]);
          const int64_t smp = (int64_t)ub2 + (kk >> 1);
          if (smp < a.B) a.stats[(smp * a.Cd + n0 + n * 16 + m) * 2 + (kk & 1)] = (double)x;
        } else {
          const auto s32 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, csum[n]), __builtin_bit_cast(int, csq[n]), false, false);
          const float ab = __builtin_bit_cast(float, (int)s32[0]) + __builtin_bit_cast(float, (int)s32[1]);   // rows 0,1: sum; rows 2,3: squares
          const float x = xsum16_f(ab);
          if (!(kk & 1))
            a.stats[(((int64_t)ub2 * a.rows_per_sample + row) * a.Cd + n0 + n * 16 + m) * 2 + (kk >> 1)] = (double)x;
        }
      }
    }
  } else if (a.stats) {
    // all rows of a block belong to one sample (host guarantees Nd % ROWS_PER_BLOCK == 0)
    double* red = reinterpret_cast<double*>(lds);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      double s = csum[n], q = csq[n];
      s = xsum32_d(xsum16_d(s));
      q = xsum32_d(xsum16_d(q));
      if (writer && kk == 0) { red[((wave * NT + n) * 16 + m) * 2] = s; red[((wave * NT + n) * 16 + m) * 2 + 1] = q; }
    }
    __syncthreads();
    const int nw = (KSPLIT == 1) ? 4 : 1;
    if (threadIdx.x < NT * 16 * 2) {
      const int q2 = threadIdx.x & 1, col = (threadIdx.x >> 1) & 15, n = threadIdx.x >> 5;
      double s = 0;
      for (int w = 0; w < nw; ++w) s += red[((w * NT + n) * 16 + col) * 2 + q2];
      const uint32_t first = (uint32_t)bx * ROWS_PER_BLOCK;
      uint32_t ub2, ur2;
      a.fNd.divmod(first, ub2, ur2);
      const int b = (int)ub2;
      const int row = (int)(ur2 / ROWS_PER_BLOCK);
      a.stats[(((int64_t)b * a.rows_per_sample + row) * a.Cd + n0 + n * 16 + col) * 2 + q2] = s;
    }
  }
}

template <int MT, int NT, int KSPLIT>
__device__ __forceinline__ void gemm16_body(const MfArgs& a, const int bx, const int by, float* lds) {
  if (a.flags & N3D_MM_BF16) gemm16_body_t<MT, NT, KSPLIT, true>(a, bx, by, lds);
  else gemm16_body_t<MT, NT, KSPLIT, false>(a, bx, by, lds);
}

template <int MT, int NT, int KSPLIT>
__global__ __launch_bounds__(KSPLIT == 16 ? 1024 : 256, KSPLIT == 16 ? 4 : 2) void conv_gemm16_kernel(MfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  gemm16_body<MT, NT, KSPLIT>(a, blockIdx.x, blockIdx.y, lds);
}

// ------------------------------------------------------------------------------------------------
// weight gradient for 16-multiple channel counts:  G[ci][co] (per tap) = sum_v X[map(v,tap)][ci] * dY[v][co]
//   MFMA A[m = ci][k = voxel], B[k = voxel][n = co]; the voxel range is split over workgroups (grid.y) and
//   over the 4 waves of a workgroup; partial slabs are summed by wgrad_final_batch_kernel in fixed order.
// ------------------------------------------------------------------------------------------------
struct Wg16Args {
  const float* x; int64_t xld; int Di, Hi, Wi, Ci;
  const float* dy; int64_t dyld; int Do, Ho, Wo, Co;
  int B, k, stride, dil, pad, flags;
  const float* in_gate;
  float* partial;  // [nchunks][ntiles][256]
  float* pbias;    // [nchunks][tco][16]
  int tci, tco;
  int64_t chunk;   // voxels (flattened b,o) per workgroup, multiple of 16
  FastDiv fNo, fWo, fHo, fTco, fTci;
};

// body for one (tile, chunk) unit executed by 256 threads (tid = 0..255); `active` = false units only take part in
// the workgroup barrier (the dual backward kernel packs four units into one 1024-thread workgroup)
__device__ __forceinline__ void wgrad16_body(const Wg16Args& a, const int tile, const int cy, const int ntiles, const int tid,
                                             const bool active, f32x4* l4 /*[3*64]*/, float (*lb)[16] /*[4][16]*/) {
  const int lane = tid & 63, wave = tid >> 6;
  const int m = lane & 15, kk = lane >> 4;
  uint32_t t1, ucot, utap, ucit;
  a.fTco.divmod((uint32_t)tile, t1, ucot);
  a.fTci.divmod(t1, utap, ucit);
  const int cot = (int)ucot, cit = (int)ucit, tap = (int)utap;
  const int kd = (a.k == 3) ? tap / 9 : 0, kh = (a.k == 3) ? (tap % 9) / 3 : 0, kw = (a.k == 3) ? tap % 3 : 0;
  const int64_t No = (int64_t)a.Do * a.Ho * a.Wo, Ni = (int64_t)a.Di * a.Hi * a.Wi;
  const int64_t total = (int64_t)a.B * No;
  const int64_t c0 = (int64_t)cy * a.chunk;
  int64_t c1 = c0 + a.chunk;
  if (c1 > total) c1 = total;
  if (!active) c1 = c0;  // no work
  const bool relu_in = a.flags & N3D_RELU_IN;
  const bool do_bias = (tap == 0 && cit == 0);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  // this lane's voxel walks i = c0 + wave*4 + kk, step 16
  int64_t i = c0 + wave * 4 + kk;
  uint32_t ub, uo, q1, uw, ud, uh;
  a.fNo.divmod((uint32_t)i, ub, uo);
  a.fWo.divmod(uo, q1, uw);
  a.fHo.divmod(q1, ud, uh);
  int b = (int)ub, ow = (int)uw, oh = (int)uh, od = (int)ud;
  // four voxel steps per iteration: all eight loads are requested before the first MFMA consumes them
  for (; i - kk - wave * 4 < c1; i += 64) {
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t iu = i + 16 * u;
      const bool in = iu < c1;
      const int64_t ic = in ? iu : c0;
      bv[u] = a.dy[ic * a.dyld + cot * 16 + m];
      const int id = od * a.stride - a.pad + kd * a.dil, ih = oh * a.stride - a.pad + kh * a.dil, iw = ow * a.stride - a.pad + kw * a.dil;
      const bool ok = in && id >= 0 && id < a.Di && ih >= 0 && ih < a.Hi && iw >= 0 && iw < a.Wi;
      const int cd_ = min(max(id, 0), a.Di - 1), ch_ = min(max(ih, 0), a.Hi - 1), cw_ = min(max(iw, 0), a.Wi - 1);
      const int bc = min(b, a.B - 1);
      float xv = a.x[((int64_t)bc * Ni + ((int64_t)cd_ * a.Hi + ch_) * a.Wi + cw_) * a.xld + cit * 16 + m];
      if (relu_in) xv = fmaxf(xv, 0.f);
      if (a.in_gate) xv *= a.in_gate[(int64_t)bc * a.Ci + cit * 16 + m];
      av[u] = ok ? xv : 0.f;
      if (!in) bv[u] = 0.f;
      // advance this lane's voxel by 16
      ow += 16;
      while (ow >= a.Wo) {
        ow -= a.Wo;
        if (++oh >= a.Ho) { oh = 0; if (++od >= a.Do) { od = 0; ++b; } }
      }
    }
    if (a.flags & N3D_MM_BF16) {
      // (the K index of the 16-deep instruction is 4 kk + u: any one-to-one map of K onto voxels serves a sum over voxels)
      acc = mm_bf16(mm_cvt4(av[0], av[1], av[2], av[3]), mm_cvt4(bv[0], bv[1], bv[2], bv[3]), acc);
#pragma unroll
      for (int u = 0; u < 4; ++u) bsum += bv[u];
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
        bsum += bv[u];
      }
    }
  }
  if (wave > 0) l4[(wave - 1) * 64 + lane] = acc;
  if (do_bias) {
    bsum = xsum32_f(xsum16_f(bsum));
    if (kk == 0) lb[wave][m] = bsum;
  }
  __syncthreads();
  if (wave == 0 && active) {
    acc += l4[lane]; acc += l4[64 + lane]; acc += l4[128 + lane];
    float* p = a.partial + ((int64_t)cy * ntiles + tile) * 256;
    // D: rows (ci) 4*kk + r, column (co) m  ->  q = ci*16 + co
#pragma unroll
    for (int r = 0; r < 4; ++r) p[(kk * 4 + r) * 16 + m] = acc[r];
    if (do_bias && lane < 16) a.pbias[((int64_t)cy * a.tco + cot) * 16 + lane] = lb[0][lane] + lb[1][lane] + lb[2][lane] + lb[3][lane];
  }
}

__global__ __launch_bounds__(256, 2) void conv_wgrad16_kernel(Wg16Args a) {
  __shared__ f32x4 l4[3 * 64];
  __shared__ float lb[4][16];
  wgrad16_body(a, blockIdx.x, blockIdx.y, gridDim.x, threadIdx.x, true, l4, lb);
}

// Backward of one deep-level conv in ONE launch: workgroups [0, nA) run the data-gradient GEMM (K split over the 16
// waves), workgroups [nA, ..) each run four (tap, channel tile, chunk) units of the weight gradient.  Both halves
// only read the same d(raw) tensor, so nothing orders them; on the 2^3 .. 8^3 levels either half alone leaves most
// of the chip idle and costs a full launch + memory round trip.
struct DualArgs { MfArgs d; Wg16Args w; int nA, gxA, ntilesB, nB; };

template <int KS>
__global__ __launch_bounds__(KS == 16 ? 1024 : 256, KS == 16 ? 4 : 2) void conv_bwd16_dual_kernel(DualArgs q) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int UNITS = KS == 16 ? 4 : 1;  // weight-gradient units (256 threads each) per workgroup
  const int L = blockIdx.x;
  if (L < q.nA) {
    gemm16_body<1, 1, KS>(q.d, L % q.gxA, L / q.gxA, lds);
  } else {
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);
    float (*lb)[16] = reinterpret_cast<float (*)[16]>(lds + UNITS * 3 * 64 * 4);
    const int sub = threadIdx.x >> 8;
    const int unit = (L - q.nA) * UNITS + sub;
    const bool active = unit < q.nB;
    const int u = active ? unit : 0;
    wgrad16_body(q.w, u % q.ntilesB, u / q.ntilesB, q.ntilesB, threadIdx.x & 255, active, l4 + sub * 3 * 64, lb + sub * 4);
  }
}


// Two independent small convolutions in ONE launch (the two ops of a searched-cell node, or a cell's two preprocess
// convs): forward pair = two K-split GEMMs, backward "quad" = two (data gradient + weight gradient) duals.  On the
// 2^3 .. 16^3 levels each problem alone occupies a fraction of the chip and costs a full launch + memory round trip.
struct PairArgs { MfArgs a0, a1; int n0, gx0, gx1; };

template <int KS>
__global__ __launch_bounds__(KS == 16 ? 1024 : 256, KS == 16 ? 4 : 2) void conv_gemm16_pair_kernel(PairArgs q) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = blockIdx.x;
  if (L < q.n0) gemm16_body<1, 1, KS>(q.a0, L % q.gx0, L / q.gx0, lds);
  else gemm16_body<1, 1, KS>(q.a1, (L - q.n0) % q.gx1, (L - q.n0) / q.gx1, lds);
}

// up to four independent small convs in one launch (the plain-conv primitives of a supernet node, cell.py:76-81): the
// descriptor of the workgroup's conv is copied out of the kernel arguments by a switch (static indexing), one body
struct MultiArgs { MfArgs a[4]; int start[5]; int gx[4]; };

template <int KS>
__global__ __launch_bounds__(KS == 16 ? 1024 : 256, KS == 16 ? 4 : 2) void conv_gemm16_multi_kernel(MultiArgs q) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int L = blockIdx.x;
  const int k = (L >= q.start[1]) + (L >= q.start[2]) + (L >= q.start[3]);
  MfArgs a;
  int s0, gx;
  switch (k) {
    case 0: a = q.a[0]; s0 = q.start[0]; gx = q.gx[0]; break;
    case 1: a = q.a[1]; s0 = q.start[1]; gx = q.gx[1]; break;
    case 2: a = q.a[2]; s0 = q.start[2]; gx = q.gx[2]; break;
    default: a = q.a[3]; s0 = q.start[3]; gx = q.gx[3]; break;
  }
  gemm16_body<1, 1, KS>(a, (L - s0) % gx, (L - s0) / gx, lds);
}

struct QuadArgs { DualArgs q0, q1; int n0; };

template <int KS>
__global__ __launch_bounds__(KS == 16 ? 1024 : 256, KS == 16 ? 4 : 2) void conv_bwd16_quad_kernel(QuadArgs z) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int UNITS = KS == 16 ? 4 : 1;
  const bool second = (int)blockIdx.x >= z.n0;
  const DualArgs& q = second ? z.q1 : z.q0;
  const int L = second ? blockIdx.x - z.n0 : blockIdx.x;
  if (L < q.nA) {
    gemm16_body<1, 1, KS>(q.d, L % q.gxA, L / q.gxA, lds);
  } else {
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);
    float (*lb)[16] = reinterpret_cast<float (*)[16]>(lds + UNITS * 3 * 64 * 4);
    const int sub = threadIdx.x >> 8;
    const int unit = (L - q.nA) * UNITS + sub;
    const bool active = unit < q.nB;
    const int u = active ? unit : 0;
    wgrad16_body(q.w, u % q.ntilesB, u / q.ntilesB, q.ntilesB, threadIdx.x & 255, active, l4 + sub * 3 * 64, lb + sub * 4);
  }
}

// ------------------------------------------------------------------------------------------------
// vox64 family -- the FLOP-heavy shallow levels: 3x3x3 stride-1 (dilation 1 or 2) convolution with C = 4 or 8
// channels on 16^3 .. 128^3 volumes (up-cells 3/4 and their data gradients: > 60 % of the net's FLOPs).
//   N = C is far too small for the 16x16 / 32x32 MFMA shapes, so the kernel uses the 16-block form
//   v_mfma_f32_4x4x1_16b_f32: one instruction = 16 independent (4 voxels x 4 channels) outer products,
//   i.e. 64 voxels x 4 output channels x K=1 at the full fp32 matrix rate (512 FLOP / 8 cycles / SIMD).
//   A operand: lane l = voxel l of a 64-voxel group (4 H rows x 16 W), read from the LDS halo tile with one
//              ds_read_b128 = 4 input channels of one tap;
//   B operand: lane l holds W[tap][cd = 4*half + (l&3)][cs]: identical in all 16 blocks (weight broadcast),
//              read from an LDS copy of the packed weights (4 distinct addresses per wave -> broadcast);
//   D: lane (block b = l>>2, j = l&3) holds channel 4*half + j of voxels 4b .. 4b+3 (4 accumulator registers).
// ONE WAVE = ONE WORKGROUP: a wave stages its own (TD+2d) x (4+2d) x (16+2d) halo tile (10-28 KB of LDS), so
// there is no barrier anywhere; 8-12 independent waves per CU hide each other's fill latency, and small tiles
// give thousands of workgroups even on the 32^3 level.  Loop order: (kh,kw) outer with the three kd weights in
// registers, input plane dz inner, so every A read feeds up to 3 output planes (12 MFMAs on 3 independent
// accumulator chains) -- 54 + 27 LDS reads per 432 MFMAs at C = 4.
// The data gradient is the same kernel on spatially flipped, channel-transposed weights (pack kernel).
// ------------------------------------------------------------------------------------------------
__device__ float4 n3d_zero_page[1];  // zero-initialised; source of the conv zero padding for LDS-DMA fills
static const void* zero_page_ptr() {
  static thread_local const void* p = nullptr;
  static thread_local int dev = -1;
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) return nullptr;
  if (!p || d != dev) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(n3d_zero_page)) != hipSuccess) return nullptr;  // callers fall back to the kernels without LDS-DMA
    p = q; dev = d;
  }
  return p;
}

struct VxArgs {
  const float* src; int64_t sld;
  float* dst; int64_t dld;
  const float* wq;      // packed [27][C (cd)][C (cs)]
  const float* bias;
  int D, H, W, flags;
  double* stats; int rows_per_sample;
  int tiles;            // tiles per sample
  const void* zero_page; // 16 zero bytes in device memory
  // workgroup index -> (sample, tile row, tile column) without run-time integer divisions (three of them, ~0.2 us of scalar work in front of
  // the first fill instruction of every wave): by tiles per sample, tile columns (W / 16) and tile rows (H / (4 NW))
  FastDiv fT, fTw, fTh;
};

// Wq[tap][cd][cs]; forward: cd=co, cs=ci, tap'=tap; data gradient: cd=ci, cs=co, tap'=26-tap
__global__ void pack_vox_kernel(const float* __restrict__ w, float* __restrict__ wq, int C, int data_grad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 27 * C * C) return;
  const int cs = i % C, cd = (i / C) % C, tap = i / (C * C);
  // data_grad: 0 forward; 1 stride-1 data gradient (channels transposed, taps flipped); 2 gather form of the stride-2
  // data gradient / transposed forward (channels transposed, taps as they are: the gather map carries the direction)
  const int co = data_grad ? cs : cd, ci = data_grad ? cd : cs, t2 = data_grad == 1 ? 26 - tap : tap;
  wq[i] = w[((int64_t)co * C + ci) * 27 + t2];
}

// NW = waves per workgroup (1 or 2).  With NW = 2 the workgroup owns an 8-row tile: each wave computes its own 4 rows
// from ONE shared halo tile (6 x 10 x 18 positions instead of 2 x 6 x 6 x 18), which cuts the LDS-DMA instructions
// per output voxel by 29 % -- the fill is bound by the texture-address path (16 cycles per 1 KiB instruction per CU).
// The two waves split the fill chunk-wise and meet at ONE workgroup barrier before the MFMA phase.
#define VOX_LB 2     // minimum waves per SIMD the vox64 kernels are compiled for (190 VGPRs: held to 170 or 128 they are 8-55 % slower)
// The kernel body is a device function of (arguments, workgroup index, workgroup count, LDS base): conv_vox64_kernel runs it for
// one conv per launch, conv_vox_multi_kernel for several independent convs in one launch (each with its own range of workgroups)
template <int C, int TD, int DIL, int NW>
__device__ __forceinline__ void vox64_body(const VxArgs& a, const int wg_raw, const int nwg, float4* const vlds) {
  constexpr int Q = C / 4, GH = 4 * NW, GW = 16;
  constexpr int LD = TD + 2 * DIL, LH = GH + 2 * DIL, LW = GW + 2 * DIL;
  constexpr int PLANE = LH * LW, NPOS = (PLANE + 63) / 64, PSTRIDE = NPOS * 64, QSTRIDE = LD * PSTRIDE;
  constexpr int NW4 = 27 * C * Q, NWI = (NW4 + 63) / 64;  // float4 count of the packed weights
  // vlds: tile [Q][LD][PSTRIDE >= LH*LW], then weights [27][C][Q]
  float4* tile = vlds;
  float4* wl = vlds + Q * QSTRIDE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware placement: workgroup ids are dealt round-robin to the 8 XCDs (private L2 each); remap so that every
  // XCD works on one contiguous run of tiles (a D-slab of one sample) and the halo re-reads of neighbouring tiles
  // hit that XCD's L2 instead of being fetched from HBM once per XCD.  (Inside a multi-conv launch wg_raw counts from the conv's first
  // workgroup: indices with equal wg_raw & 7 still share one physical XCD.)
  int wg = wg_raw;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  uint32_t ub, utile, ubx, uw, ud, uh;
  a.fT.divmod((uint32_t)wg, ub, utile);
  a.fTw.divmod(utile, ubx, uw);
  a.fTh.divmod(ubx, ud, uh);
  const int b = (int)ub, tile_id = (int)utile;
  const int w0 = (int)uw * GW, h0 = (int)uh * GH, d0 = (int)ud * TD;
  const int64_t N = (int64_t)a.D * a.H * a.W;
  const float* srcb = a.src + (int64_t)b * N * a.sld;
  const int j = lane & 3;

  // lane -> voxel: row hh = lane/16; odd rows are rotated by LW % 16 voxels so that the fixed 16-lane groups a
  // ds_read_b128 is serviced in ({0-3,12-15,20-27}, ...: 8 lanes of an even row + 8 of the next odd row) fall on
  // 64 distinct banks with the (16 + 2*DIL)-float4 row pitch (2-way conflicts on every A read otherwise)
  const int hh = lane >> 4, ww = ((lane & 15) - (hh & 1) * (LW % 16)) & 15;
  const int hrow = 4 * wave + hh;  // row inside the workgroup's tile
  // bias and (accumulate mode) the previous output values are ordinary global loads: issued BEFORE the LDS-DMA
  // fill so that the single vmcnt(0) below covers them (a load after the fill would add a second memory latency)
  float* dstb = a.dst + (int64_t)b * N * a.dld;
  const bool accum = a.flags & N3D_ACCUMULATE;
  const int64_t vox_off = ((int64_t)(h0 + 4 * wave + hh) * a.W + w0 + ww);
  float4 biasv[Q], prevv[TD][Q];
#pragma unroll
  for (int hf = 0; hf < Q; ++hf) {
    biasv[hf] = a.bias ? *reinterpret_cast<const float4*>(a.bias + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < TD; ++g)
      prevv[g][hf] = accum ? *reinterpret_cast<const float4*>(dstb + (((int64_t)(d0 + g) * a.H * a.W) + vox_off) * a.dld + hf * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- stage weights and the halo tile with LDS-DMA (global_load_lds_dwordx4: HBM/L2 -> LDS, no VGPR staging, no
  // ds_write traffic -- the register-staged fill spent ~1 us in ds_write_b128 issue with 8 waves per CU).  One
  // instruction fills 64 consecutive float4 slots from 64 per-lane addresses: lane l owns the in-plane positions
  // l, l+64, .. of every plane (plane stride padded to a multiple of 64); positions outside the volume (zero
  // padding) and pad slots read a 16-byte zero page instead.  Same wave reads what it wrote: vmcnt(0), no barrier.
  {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float4* __restrict__ wq4 = reinterpret_cast<const float4*>(a.wq);
    const float4* zp = reinterpret_cast<const float4*>(a.zero_page);
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int idx = lane + i * 64;
      if (NW == 1 || (i % NW) == wave)
        __builtin_amdgcn_global_load_lds((gptr_t)(idx < NW4 ? wq4 + idx : zp), (lptr_t)(wl + i * 64), 16, 0, 0);
    }
    const int64_t pstride = (int64_t)a.H * a.W * a.sld;
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
      const int pos = lane + i * 64;
      const int wx = pos % LW, hy = pos / LW;
      const int gh = h0 - DIL + hy, gw = w0 - DIL + wx;
      const bool okp = pos < PLANE && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
      const float* prow = srcb + ((int64_t)gh * a.W + gw) * a.sld;
      // the workgroup's waves take alternate planes (dz = m*NW + wave: no branch, only the plane address and the LDS
      // destination depend on the wave)
      static_assert(LD % NW == 0, "tile depth must split evenly over the waves");
#pragma unroll
      for (int m = 0; m < LD / NW; ++m) {
        const int dz = m * NW + (NW > 1 ? wave : 0);
        const int gd = d0 - DIL + dz;
        const bool inb = okp && gd >= 0 && gd < a.D;
        const float* p = prow + gd * pstride;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          __builtin_amdgcn_global_load_lds((gptr_t)(inb ? reinterpret_cast<const float4*>(p + q * 4) : zp),
                                           (lptr_t)(tile + q * QSTRIDE + dz * PSTRIDE + i * 64), 16, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (NW > 1) __syncthreads();  // each wave reads rows the other wave's DMA filled

  // accumulators: the WEIGHTS are the MFMA A operand (row i = output channel) and the voxels the B operand
  // (column j = voxel), so lane l ends up with all four channels of ITS OWN voxel in the four accumulator
  // registers: the epilogue is one float4 store per lane, no cross-lane transpose.
  f32x4 acc[TD][Q];
#pragma unroll
  for (int hf = 0; hf < Q; ++hf) {
    const f32x4 bv = {biasv[hf].x, biasv[hf].y, biasv[hf].z, biasv[hf].w};
#pragma unroll
    for (int g = 0; g < TD; ++g) acc[g][hf] = bv;
  }
  float cs[Q][4], cq[Q][4];  // per-lane GroupNorm partial sums of the lane's voxels, per channel
#pragma unroll
  for (int hf = 0; hf < Q; ++hf)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[hf][r] = cq[hf][r] = 0.f;
  // finished output plane g: statistics (of the convolution result itself) + one 16-byte store per lane and half
  float* const o_plane0 = dstb + ((int64_t)d0 * a.H * a.W + vox_off) * a.dld;
  const int64_t o_pstride = (int64_t)a.H * a.W * a.dld;
  auto emit_plane = [&](int g) {
    float* o = o_plane0 + g * o_pstride;
#pragma unroll
    for (int hf = 0; hf < Q; ++hf) {
      const f32x4 v = acc[g][hf];
#pragma unroll
      for (int r = 0; r < 4; ++r) { cs[hf][r] += v[r]; cq[hf][r] = fmaf(v[r], v[r], cq[hf][r]); }
      float4* op = reinterpret_cast<float4*>(o + hf * 4);
      float4 w4 = make_float4(v[0], v[1], v[2], v[3]);
      { const float4 pv = prevv[g][hf]; w4.x += pv.x; w4.y += pv.y; w4.z += pv.z; w4.w += pv.w; }
      *op = w4;
    }
  };
  if constexpr (Q == 1) {
    // C = 4: all 27 weight quads live in registers (108 VGPRs) and the loop is INPUT-PLANE major: plane dz feeds
    // the output planes dz, dz-DIL, dz-2*DIL, and an output plane is stored as soon as its last input plane is
    // done -- its store latency and epilogue VALU work overlap the partner wave's MFMAs instead of forming a tail.
    // weight quads are fetched from LDS one kd-slab ahead of the first input plane that needs them (kd = 0 before
    // plane 0, kd = 1 during plane 0, ...) so the start of the MFMA phase is not one 45-read LDS burst per wave
    float4 wr[27];
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t] = wl[t * 4 + j];
    f32x4 acc2[TD];
#pragma unroll
    for (int g = 0; g < TD; ++g) acc2[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the nine A reads of plane dz+1 are issued before the MFMAs of plane dz (register double buffer)
    float4 avb[2][9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) avb[0][t9] = tile[(hrow + (t9 / 3) * DIL) * LW + (ww + (t9 % 3) * DIL)];
#pragma unroll
    for (int dz = 0; dz < LD; ++dz) {
      if (dz + 1 < LD) {
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
          avb[(dz + 1) & 1][t9] = tile[(dz + 1) * PSTRIDE + (hrow + (t9 / 3) * DIL) * LW + (ww + (t9 % 3) * DIL)];
      }
      if (dz == 0 || dz == DIL) {
        const int kd = dz / DIL + 1;
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[kd * 9 + t] = wl[(kd * 9 + t) * 4 + j];
      }
      const float4* av = avb[dz & 1];
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xe = e == 0 ? av[t9].x : (e == 1 ? av[t9].y : (e == 2 ? av[t9].z : av[t9].w));
#pragma unroll
          for (int kd = 0; kd < 3; ++kd) {
            const int g = dz - kd * DIL;
            if (g >= 0 && g < TD) {
              const float4 wv = wr[kd * 9 + t9];
              const float we = e == 0 ? wv.x : (e == 1 ? wv.y : (e == 2 ? wv.z : wv.w));
              // two accumulator chains per output plane (even / odd input channel): a dependent 4x4x1 chain
              // issues one MFMA per ~15 cycles, and the first / last input planes feed a single output plane
              if (e & 1) acc2[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(we, xe, acc2[g], 0, 0, 0);
              else acc[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(we, xe, acc[g][0], 0, 0, 0);
            }
          }
        }
      }
      if (dz - 2 * DIL >= 0) {
        acc[dz - 2 * DIL][0] += acc2[dz - 2 * DIL];
        emit_plane(dz - 2 * DIL);
      }
    }
  } else {
    // C = 8: (kh,kw) outer with the three kd weight sets in registers, input plane inner (every A read feeds up
    // to 3 output planes on independent accumulator chains)
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int kh = t9 / 3, kw = t9 % 3;
      float4 wr[3][Q][Q];  // [kd][half][quad]
#pragma unroll
      for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int hf = 0; hf < Q; ++hf)
#pragma unroll
          for (int q = 0; q < Q; ++q) wr[kd][hf][q] = wl[((kd * 9 + t9) * C + hf * 4 + j) * Q + q];
      const int base = (hrow + kh * DIL) * LW + (ww + kw * DIL);
      float4 av[LD][Q];
#pragma unroll
      for (int dz = 0; dz < LD; ++dz)
#pragma unroll
        for (int q = 0; q < Q; ++q) av[dz][q] = tile[q * QSTRIDE + dz * PSTRIDE + base];
#pragma unroll
      for (int dz = 0; dz < LD; ++dz) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xe = e == 0 ? av[dz][q].x : (e == 1 ? av[dz][q].y : (e == 2 ? av[dz][q].z : av[dz][q].w));
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
              const int g = dz - kd * DIL;
              if (g >= 0 && g < TD) {
#pragma unroll
                for (int hf = 0; hf < Q; ++hf) {
                  const float4 wv = wr[kd][hf][q];
                  const float we = e == 0 ? wv.x : (e == 1 ? wv.y : (e == 2 ? wv.z : wv.w));
                  acc[g][hf] = __builtin_amdgcn_mfma_f32_4x4x1f32(we, xe, acc[g][hf], 0, 0, 0);
                }
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < TD; ++g) emit_plane(g);
  }
  // ---- GroupNorm partial row of this tile: 8 per-lane sums (4 channels x {sum, sum of squares}) per half are
  // folded with a halving butterfly (lane^1 keeps channels {0,1} | {2,3}, lane^2 keeps one of the two), then one
  // class sum over the 16 lanes that ended up with the same channel.  fp32 tree, rows are added in fp64 downstream.
  if (a.stats) {
    const bool odd = lane & 1, hi = lane & 2;
#pragma unroll
    for (int hf = 0; hf < Q; ++hf) {
      float u[2], uq[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float keep = odd ? cs[hf][2 + k] : cs[hf][k], send = odd ? cs[hf][k] : cs[hf][2 + k];
        u[k] = keep + dpp_f<0xB1>(send);
        const float keepq = odd ? cq[hf][2 + k] : cq[hf][k], sendq = odd ? cq[hf][k] : cq[hf][2 + k];
        uq[k] = keepq + dpp_f<0xB1>(sendq);
      }
      float v1 = (hi ? u[1] : u[0]) + dpp_f<0x4E>(hi ? u[0] : u[1]);
      float v2 = (hi ? uq[1] : uq[0]) + dpp_f<0x4E>(hi ? uq[0] : uq[1]);
      v1 = wave_classsum_f(v1, 4); v2 = wave_classsum_f(v2, 4);
      if (lane < 4) {
        const int ch = (lane & 1) * 2 + (lane >> 1);
        double* o = a.stats + (((int64_t)b * a.rows_per_sample + tile_id * NW + wave) * C + hf * 4 + ch) * 2;
        reinterpret_cast<double2*>(o)[0] = make_double2((double)v1, (double)v2);
      }
    }
  }
}

template <int C, int TD, int DIL, int NW>
__global__ __launch_bounds__(64 * NW, VOX_LB) void conv_vox64_kernel(VxArgs a) {
  N3D_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float4 vlds[];
  vox64_body<C, TD, DIL, NW>(a, blockIdx.x, gridDim.x, vlds);
}

// ------------------------------------------------------------------------------------------------
// vox_s2: the same 4x4x1-MFMA scheme for the STRIDE-2 3x3x3 convs with C = 4 / 8 (down_conv / down_dil_conv forward
// and the data gradient of up_conv / up_dil_conv, which is a strided gather of dy):
//   out[o] = bias + sum_tap W[tap] . x[2*o - pad + tap*dil]          (pad == dil)
// One wave = one workgroup = TD x 4 x 16 output voxels; the (2*TD-1+2*dil) x (7+2*dil) x (31+2*dil) input region is
// filled by LDS-DMA.  Lanes read the tile with a voxel stride of 2, so the fill DE-INTERLEAVES each row by W parity
// (the per-lane source address of the DMA is free): row = [even w | odd w], and the 16 lanes of a row read 16
// consecutive float4 again; odd tile rows are rotated so that the fixed 16-lane groups of ds_read_b128 hit 64
// distinct banks.  Loop: (kh,kw) outer with the three kd weight sets in registers, input plane inner.
// ------------------------------------------------------------------------------------------------
struct Vs2Args {
  const float* src; int64_t sld; int D, H, W;      // input tensor
  float* dst; int64_t dld; int oD, oH, oW;         // output tensor (= ceil(in / 2))
  const float* wq; const float* bias; int flags;
  double* stats; int rows_per_sample; int tiles; const void* zero_page;
  FastDiv fT, fTw, fTh;      // tiles per sample, tile columns (oW / 16), tile rows (oH / 4): see VxArgs
};

template <int C, int TD, int DIL>
__device__ __forceinline__ void vox_s2_body(const Vs2Args& a, const int wg_raw, const int nwg, float4* const vlds) {
  constexpr int Q = C / 4;
  constexpr int LD = 2 * (TD - 1) + 2 * DIL + 1, LH = 7 + 2 * DIL, LW = 31 + 2 * DIL;
  constexpr int HW = (LW + 1) / 2, RW = 2 * HW;              // half-row (one W parity) and row pitch in float4
  constexpr int PLANE = LH * RW, NPOS = (PLANE + 63) / 64, PSTRIDE = NPOS * 64, QSTRIDE = LD * PSTRIDE;
  constexpr int NW4 = 27 * C * Q, NWI = (NW4 + 63) / 64;
  constexpr int ROT = ((2 * RW * 4) % 64) / 4;               // voxels odd lane rows are rotated by (bank-conflict-free reads)
  float4* tile = vlds;
  float4* wl = vlds + Q * QSTRIDE;
  const int lane = threadIdx.x;
  int wg = wg_raw;
  {  // XCD-aware placement (see vox64_body)
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  uint32_t ub, utile, ubx, uw, ud, uh;
  a.fT.divmod((uint32_t)wg, ub, utile);
  a.fTw.divmod(utile, ubx, uw);
  a.fTh.divmod(ubx, ud, uh);
  const int b = (int)ub, tile_id = (int)utile;
  const int w0 = (int)uw * 16, h0 = (int)uh * 4, d0 = (int)ud * TD;
  const int64_t Ns = (int64_t)a.D * a.H * a.W, Nd = (int64_t)a.oD * a.oH * a.oW;
  const float* srcb = a.src + (int64_t)b * Ns * a.sld;
  float* dstb = a.dst + (int64_t)b * Nd * a.dld;
  const int j = lane & 3;
  const int hh = lane >> 4, ww = ((lane & 15) - (hh & 1) * ROT) & 15;
  const bool accum = a.flags & N3D_ACCUMULATE;
  const int64_t vox_off = ((int64_t)(h0 + hh) * a.oW + w0 + ww);
  float4 biasv[Q], prevv[TD][Q];
#pragma unroll
  for (int hf = 0; hf < Q; ++hf) {
    biasv[hf] = a.bias ? *reinterpret_cast<const float4*>(a.bias + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < TD; ++g)
      prevv[g][hf] = accum ? *reinterpret_cast<const float4*>(dstb + (((int64_t)(d0 + g) * a.oH * a.oW) + vox_off) * a.dld + hf * 4)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float4* __restrict__ wq4 = reinterpret_cast<const float4*>(a.wq);
    const float4* zp = reinterpret_cast<const float4*>(a.zero_page);
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int idx = lane + i * 64;
      __builtin_amdgcn_global_load_lds((gptr_t)(idx < NW4 ? wq4 + idx : zp), (lptr_t)(wl + i * 64), 16, 0, 0);
    }
    const int64_t pstride = (int64_t)a.H * a.W * a.sld;
    const int id0 = 2 * d0 - DIL, ih0 = 2 * h0 - DIL, iw0 = 2 * w0 - DIL;
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
      const int pos = lane + i * 64;                       // LDS slot in the plane: [row][parity][half]
      const int row = pos / RW, rem = pos - row * RW;
      const int par = rem / HW, half = rem - par * HW;
      const int wx = 2 * half + par;                       // local input w
      const int gh = ih0 + row, gw = iw0 + wx;
      const bool okp = pos < PLANE && wx < LW && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
      const float* prow = srcb + ((int64_t)gh * a.W + gw) * a.sld;
#pragma unroll
      for (int dz = 0; dz < LD; ++dz) {
        const int gd = id0 + dz;
        const bool inb = okp && gd >= 0 && gd < a.D;
        const float* p = prow + gd * pstride;
#pragma unroll
        for (int q = 0; q < Q; ++q)
          __builtin_amdgcn_global_load_lds((gptr_t)(inb ? reinterpret_cast<const float4*>(p + q * 4) : zp),
                                           (lptr_t)(tile + q * QSTRIDE + dz * PSTRIDE + i * 64), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  f32x4 acc[TD][Q];
#pragma unroll
  for (int hf = 0; hf < Q; ++hf) {
    const f32x4 bv = {biasv[hf].x, biasv[hf].y, biasv[hf].z, biasv[hf].w};
#pragma unroll
    for (int g = 0; g < TD; ++g) acc[g][hf] = bv;
  }
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) {
    const int kh = t9 / 3, kw = t9 % 3;
    float4 wr[3][Q][Q];  // [kd][half][quad]
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int hf = 0; hf < Q; ++hf)
#pragma unroll
        for (int q = 0; q < Q; ++q) wr[kd][hf][q] = wl[((kd * 9 + t9) * C + hf * 4 + j) * Q + q];
    // input voxel (2*hh + kh*DIL, 2*ww + kw*DIL): W parity and half index are wave-uniform / lane-linear
    const int base = (2 * hh + kh * DIL) * RW + ((kw * DIL) & 1) * HW + ww + ((kw * DIL) >> 1);
    float4 av[LD][Q];
#pragma unroll
    for (int dz = 0; dz < LD; ++dz)
#pragma unroll
      for (int q = 0; q < Q; ++q) av[dz][q] = tile[q * QSTRIDE + dz * PSTRIDE + base];
#pragma unroll
    for (int dz = 0; dz < LD; ++dz) {
#pragma unroll
      for (int q = 0; q < Q; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xe = e == 0 ? av[dz][q].x : (e == 1 ? av[dz][q].y : (e == 2 ? av[dz][q].z : av[dz][q].w));
#pragma unroll
          for (int kd = 0; kd < 3; ++kd) {
            const int g2 = dz - kd * DIL;                 // = 2 * output plane
            if (g2 >= 0 && (g2 & 1) == 0 && (g2 >> 1) < TD) {
#pragma unroll
              for (int hf = 0; hf < Q; ++hf) {
                const float4 wv = wr[kd][hf][q];
                const float we = e == 0 ? wv.x : (e == 1 ? wv.y : (e == 2 ? wv.z : wv.w));
                acc[g2 >> 1][hf] = __builtin_amdgcn_mfma_f32_4x4x1f32(we, xe, acc[g2 >> 1][hf], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  }
  // ---- epilogue: statistics of the convolution result + one 16-byte store per lane and half
  float cs[Q][4], cq[Q][4];
#pragma unroll
  for (int hf = 0; hf < Q; ++hf)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[hf][r] = cq[hf][r] = 0.f;
#pragma unroll
  for (int g = 0; g < TD; ++g) {
    float* o = dstb + (((int64_t)(d0 + g) * a.oH * a.oW) + vox_off) * a.dld;
#pragma unroll
    for (int hf = 0; hf < Q; ++hf) {
      const f32x4 v = acc[g][hf];
#pragma unroll
      for (int r = 0; r < 4; ++r) { cs[hf][r] += v[r]; cq[hf][r] = fmaf(v[r], v[r], cq[hf][r]); }
      const float4 pv = prevv[g][hf];
      *reinterpret_cast<float4*>(o + hf * 4) = make_float4(v[0] + pv.x, v[1] + pv.y, v[2] + pv.z, v[3] + pv.w);
    }
  }
  if (a.stats) {
    const bool odd = lane & 1, hi = lane & 2;
#pragma unroll
    for (int hf = 0; hf < Q; ++hf) {
      float u[2], uq[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float keep = odd ? cs[hf][2 + k] : cs[hf][k], send = odd ? cs[hf][k] : cs[hf][2 + k];
        u[k] = keep + dpp_f<0xB1>(send);
        const float keepq = odd ? cq[hf][2 + k] : cq[hf][k], sendq = odd ? cq[hf][k] : cq[hf][2 + k];
        uq[k] = keepq + dpp_f<0xB1>(sendq);
      }
      float v1 = (hi ? u[1] : u[0]) + dpp_f<0x4E>(hi ? u[0] : u[1]);
      float v2 = (hi ? uq[1] : uq[0]) + dpp_f<0x4E>(hi ? uq[0] : uq[1]);
      v1 = wave_classsum_f(v1, 4); v2 = wave_classsum_f(v2, 4);
      if (lane < 4) {
        const int ch = (lane & 1) * 2 + (lane >> 1);
        double* o = a.stats + (((int64_t)b * a.rows_per_sample + tile_id) * C + hf * 4 + ch) * 2;
        reinterpret_cast<double2*>(o)[0] = make_double2((double)v1, (double)v2);
      }
    }
  }
}

template <int C, int TD, int DIL>
__global__ __launch_bounds__(64, 2) void conv_vox_s2_kernel(Vs2Args a) {
  N3D_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float4 vlds[];
  vox_s2_body<C, TD, DIL>(a, blockIdx.x, gridDim.x, vlds);
}

struct Vs2Plan { bool ok; int C, td, dil, tiles; size_t lds; };

// forward-type gathers only (conv forward with stride 2; data gradient of a stride-2 transposed conv)
static Vs2Plan vs2_plan(const n3d_conv_geom* g, bool data_grad) {
  Vs2Plan p; p.ok = false;
  if (data_grad || g->depthwise || g->k != 3 || g->stride != 2 || g->Ci != g->Co || (g->Ci != 4 && g->Ci != 8)) return p;
  if (!(g->dil == 1 || g->dil == 2) || g->pad != g->dil) return p;
  if (g->Wo % 16 != 0 || g->Ho % 4 != 0) return p;
  p.C = g->Ci; p.dil = g->dil;
  p.td = (g->Ci == 4 && g->Do % 2 == 0) ? 2 : 1;
  p.tiles = (g->Wo / 16) * (g->Ho / 4) * (g->Do / p.td);
  const int Q = g->Ci / 4;
  const int LD = 2 * (p.td - 1) + 2 * g->dil + 1, LH = 7 + 2 * g->dil, LW = 31 + 2 * g->dil;
  const size_t pstride = ((size_t)LH * 2 * ((LW + 1) / 2) + 63) / 64 * 64;
  p.lds = ((size_t)Q * LD * pstride + ((size_t)27 * g->Ci * Q + 63) / 64 * 64) * 16;
  p.ok = p.lds <= 160 * 1024;
  return p;
}

template <int C, int TD, int DIL>
static void launch_vs2_t(Vs2Args& a, const Vs2Plan& p, int B, hipStream_t s) {
  hipLaunchKernelGGL((conv_vox_s2_kernel<C, TD, DIL>), dim3(p.tiles * B), dim3(64), p.lds, s, a);
}

static void launch_vs2(Vs2Args& a, const Vs2Plan& p, int B, hipStream_t s) {
  if (p.C == 4) {
    if (p.td == 2) { if (p.dil == 1) launch_vs2_t<4, 2, 1>(a, p, B, s); else launch_vs2_t<4, 2, 2>(a, p, B, s); }
    else { if (p.dil == 1) launch_vs2_t<4, 1, 1>(a, p, B, s); else launch_vs2_t<4, 1, 2>(a, p, B, s); }
  } else {
    if (p.dil == 1) launch_vs2_t<8, 1, 1>(a, p, B, s); else launch_vs2_t<8, 1, 2>(a, p, B, s);
  }
}

// ------------------------------------------------------------------------------------------------
// vox_up: stride-2 3x3x3 TRANSPOSED conv forward (up_conv / up_dil_conv) and the data gradient of the stride-2 convs,
// C = 4 / 8: the output grid is twice the source grid, out[o] = bias + sum_k W[k] . x[(o + pad - k*dil) / 2] over the taps
// whose index is even.  A wave owns 4 x 16 SOURCE voxels j of one source plane and produces all 8 output parity
// classes (2*j + p) of them from one LDS halo tile (3 x 6 x 18): per dimension a source shift s feeds
//   dil = 1:  s =  0 -> (p = 0, k = 1) and (p = 1, k = 2);  s = +1 -> (p = 1, k = 0)
//   dil = 2:  s in {-1, 0, +1} -> (p = 0, k = 1 - s)         (odd outputs receive the bias only)
// so the 27 taps are spread over 8 accumulator sets (27 * Cin * Cout/4 MFMAs per 64 source voxels, none wasted).
// The generic gather walked every tap for every output voxel and masked 19 of 27.
// ------------------------------------------------------------------------------------------------
struct VupArgs {
  const float* src; int64_t sld; int D, H, W;      // source tensor (the small grid)
  float* dst; int64_t dld;                         // output tensor (2D x 2H x 2W)
  const float* wq; const float* bias; int flags;
  double* stats; int rows_per_sample; int tiles; const void* zero_page;
  FastDiv fT, fTw, fTh;      // tiles per sample, tile columns (W / 16), tile rows (H / 4): see VxArgs
};

__host__ __device__ constexpr int vup_count(int dil, int s) { return dil == 2 ? 1 : (s == 0 ? 2 : (s == 1 ? 1 : 0)); }
__host__ __device__ constexpr int vup_p(int dil, int s, int i) { return dil == 2 ? 0 : (s == 0 ? i : 1); }
__host__ __device__ constexpr int vup_k(int dil, int s, int i) { return dil == 2 ? 1 - s : (s == 0 ? 1 + i : 0); }

template <int C, int DIL>
__global__ __launch_bounds__(64, 2) void conv_vox_up_kernel(VupArgs a) {
  N3D_CHAIN_PRIO();
  constexpr int Q = C / 4;
  constexpr int LD = 3, LH = 6, LW = 18;
  constexpr int PLANE = LH * LW, NPOS = (PLANE + 63) / 64, PSTRIDE = NPOS * 64, QSTRIDE = LD * PSTRIDE;
  constexpr int NW4 = 27 * C * Q, NWI = (NW4 + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) float4 vlds[];
  float4* tile = vlds;
  float4* wl = vlds + Q * QSTRIDE;
  const int lane = threadIdx.x;
  int wg = blockIdx.x;
  {  // XCD-aware placement (see conv_vox64_kernel)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  uint32_t ub, utile, ubx, uw, ud, uh;
  a.fT.divmod((uint32_t)wg, ub, utile);
  a.fTw.divmod(utile, ubx, uw);
  a.fTh.divmod(ubx, ud, uh);
  const int b = (int)ub, tile_id = (int)utile;
  const int w0 = (int)uw * 16, h0 = (int)uh * 4, d0 = (int)ud;
  const int64_t Ns = (int64_t)a.D * a.H * a.W;
  const int oH = 2 * a.H, oW = 2 * a.W;
  const float* srcb = a.src + (int64_t)b * Ns * a.sld;
  float* dstb = a.dst + (int64_t)b * 8 * Ns * a.dld;
  const int j = lane & 3;
  const int hh = lane >> 4, ww = ((lane & 15) - (hh & 1) * (LW % 16)) & 15;   // bank-conflict-free row rotation (vox64)
  const bool accum = a.flags & N3D_ACCUMULATE;
  // output voxel of class (pd, ph, pw): (2*d0 + pd, 2*(h0+hh) + ph, 2*(w0+ww) + pw)
  const int64_t obase = (((int64_t)(2 * d0) * oH + 2 * (h0 + hh)) * oW + 2 * (w0 + ww)) * a.dld;
  auto ooff = [&](int cls) { return (((int64_t)(cls >> 2) * oH + ((cls >> 1) & 1)) * oW + (cls & 1)) * a.dld; };
  float4 biasv[Q], prevv[8][Q];
#pragma unroll
  for (int hf = 0; hf < Q; ++hf) {
    biasv[hf] = a.bias ? *reinterpret_cast<const float4*>(a.bias + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int cls = 0; cls < 8; ++cls)
      prevv[cls][hf] = accum ? *reinterpret_cast<const float4*>(dstb + obase + ooff(cls) + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float4* __restrict__ wq4 = reinterpret_cast<const float4*>(a.wq);
    const float4* zp = reinterpret_cast<const float4*>(a.zero_page);
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int idx = lane + i * 64;
      __builtin_amdgcn_global_load_lds((gptr_t)(idx < NW4 ? wq4 + idx : zp), (lptr_t)(wl + i * 64), 16, 0, 0);
    }
    const int64_t pstride = (int64_t)a.H * a.W * a.sld;
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
      const int pos = lane + i * 64;
      const int wx = pos % LW, hy = pos / LW;
      const int gh = h0 - 1 + hy, gw = w0 - 1 + wx;
      const bool okp = pos < PLANE && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
      const float* prow = srcb + ((int64_t)gh * a.W + gw) * a.sld;
#pragma unroll
      for (int dz = 0; dz < LD; ++dz) {
        const int gd = d0 - 1 + dz;
        const bool inb = okp && gd >= 0 && gd < a.D;
        const float* p = prow + gd * pstride;
#pragma unroll
        for (int q = 0; q < Q; ++q)
          __builtin_amdgcn_global_load_lds((gptr_t)(inb ? reinterpret_cast<const float4*>(p + q * 4) : zp),
                                           (lptr_t)(tile + q * QSTRIDE + dz * PSTRIDE + i * 64), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  f32x4 acc[8][Q];
#pragma unroll
  for (int hf = 0; hf < Q; ++hf) {
    const f32x4 bv = {biasv[hf].x, biasv[hf].y, biasv[hf].z, biasv[hf].w};
#pragma unroll
    for (int cls = 0; cls < 8; ++cls) acc[cls][hf] = bv;
  }
#pragma unroll
  for (int sd = -1; sd <= 1; ++sd) {
    if (vup_count(DIL, sd) == 0) continue;
    // the source positions of this plane that feed anything are requested first, then consumed
    float4 av[3][3][Q];
#pragma unroll
    for (int sh = -1; sh <= 1; ++sh)
#pragma unroll
      for (int sw = -1; sw <= 1; ++sw)
        if (vup_count(DIL, sh) > 0 && vup_count(DIL, sw) > 0) {
#pragma unroll
          for (int q = 0; q < Q; ++q) av[sh + 1][sw + 1][q] = tile[q * QSTRIDE + (1 + sd) * PSTRIDE + (hh + 1 + sh) * LW + (ww + 1 + sw)];
        }
#pragma unroll
    for (int sh = -1; sh <= 1; ++sh)
#pragma unroll
      for (int sw = -1; sw <= 1; ++sw) {
        if (vup_count(DIL, sh) == 0 || vup_count(DIL, sw) == 0) continue;
#pragma unroll
        for (int id = 0; id < vup_count(DIL, sd); ++id)
#pragma unroll
          for (int ih = 0; ih < vup_count(DIL, sh); ++ih)
#pragma unroll
            for (int iw = 0; iw < vup_count(DIL, sw); ++iw) {
              const int cls = vup_p(DIL, sd, id) * 4 + vup_p(DIL, sh, ih) * 2 + vup_p(DIL, sw, iw);
              const int tap = vup_k(DIL, sd, id) * 9 + vup_k(DIL, sh, ih) * 3 + vup_k(DIL, sw, iw);
              float4 wr[Q][Q];
#pragma unroll
              for (int hf = 0; hf < Q; ++hf)
#pragma unroll
                for (int q = 0; q < Q; ++q) wr[hf][q] = wl[(tap * C + hf * 4 + j) * Q + q];
#pragma unroll
              for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float4 x4 = av[sh + 1][sw + 1][q];
                  const float xe = e == 0 ? x4.x : (e == 1 ? x4.y : (e == 2 ? x4.z : x4.w));
#pragma unroll
                  for (int hf = 0; hf < Q; ++hf) {
                    const float4 wv = wr[hf][q];
                    const float we = e == 0 ? wv.x : (e == 1 ? wv.y : (e == 2 ? wv.z : wv.w));
                    acc[cls][hf] = __builtin_amdgcn_mfma_f32_4x4x1f32(we, xe, acc[cls][hf], 0, 0, 0);
                  }
                }
            }
      }
  }
  // ---- epilogue: 8 stores of 16 bytes per lane and half (the two W classes of a lane are adjacent voxels)
  float cs[Q][4], cq[Q][4];
#pragma unroll
  for (int hf = 0; hf < Q; ++hf)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[hf][r] = cq[hf][r] = 0.f;
#pragma unroll
  for (int cls = 0; cls < 8; ++cls) {
    float* o = dstb + obase + ooff(cls);
#pragma unroll
    for (int hf = 0; hf < Q; ++hf) {
      const f32x4 v = acc[cls][hf];
#pragma unroll
      for (int r = 0; r < 4; ++r) { cs[hf][r] += v[r]; cq[hf][r] = fmaf(v[r], v[r], cq[hf][r]); }
      const float4 pv = prevv[cls][hf];
      *reinterpret_cast<float4*>(o + hf * 4) = make_float4(v[0] + pv.x, v[1] + pv.y, v[2] + pv.z, v[3] + pv.w);
    }
  }
  if (a.stats) {
    const bool odd = lane & 1, hi = lane & 2;
#pragma unroll
    for (int hf = 0; hf < Q; ++hf) {
      float u[2], uq[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float keep = odd ? cs[hf][2 + k] : cs[hf][k], send = odd ? cs[hf][k] : cs[hf][2 + k];
        u[k] = keep + dpp_f<0xB1>(send);
        const float keepq = odd ? cq[hf][2 + k] : cq[hf][k], sendq = odd ? cq[hf][k] : cq[hf][2 + k];
        uq[k] = keepq + dpp_f<0xB1>(sendq);
      }
      float v1 = (hi ? u[1] : u[0]) + dpp_f<0x4E>(hi ? u[0] : u[1]);
      float v2 = (hi ? uq[1] : uq[0]) + dpp_f<0x4E>(hi ? uq[0] : uq[1]);
      v1 = wave_classsum_f(v1, 4); v2 = wave_classsum_f(v2, 4);
      if (lane < 4) {
        const int ch = (lane & 1) * 2 + (lane >> 1);
        double* o = a.stats + (((int64_t)b * a.rows_per_sample + tile_id) * C + hf * 4 + ch) * 2;
        reinterpret_cast<double2*>(o)[0] = make_double2((double)v1, (double)v2);
      }
    }
  }
}

struct VupPlan { bool ok; int C, dil, tiles; size_t lds; };

// gather form with den = 2 only (transposed conv forward with stride 2; data gradient of a stride-2 conv) on exactly doubled grids
static VupPlan vup_plan(const n3d_conv_geom* g, bool data_grad) {
  VupPlan p; p.ok = false;
  if (!data_grad || g->depthwise || g->k != 3 || g->stride != 2 || g->Ci != g->Co || (g->Ci != 4 && g->Ci != 8)) return p;
  if (!(g->dil == 1 || g->dil == 2) || g->pad != g->dil) return p;
  if (g->Di != 2 * g->Do || g->Hi != 2 * g->Ho || g->Wi != 2 * g->Wo) return p;
  if (g->Wo % 16 != 0 || g->Ho % 4 != 0) return p;
  p.C = g->Ci; p.dil = g->dil;
  p.tiles = (g->Wo / 16) * (g->Ho / 4) * g->Do;
  const int Q = g->Ci / 4;
  p.lds = ((size_t)Q * 3 * 128 + ((size_t)27 * g->Ci * Q + 63) / 64 * 64) * 16;
  p.ok = true;
  return p;
}

struct VxPlan { bool ok; int C, td, dil, tiles, nw; size_t lds; };

static VxPlan vx_plan(const n3d_conv_geom* g) {
  VxPlan p; p.ok = false;
  if (g->depthwise || g->k != 3 || g->stride != 1 || g->Ci != g->Co || (g->Ci != 4 && g->Ci != 8)) return p;
  if (!(g->dil == 1 || g->dil == 2) || g->pad != g->dil) return p;
  const int W = g->Wi, H = g->Hi, D = g->Di;
  if (W % 16 != 0 || H % 4 != 0) return p;
  const int64_t groups = (int64_t)g->B * D * (H / 4) * (W / 16);
  // tile depth: as deep as possible while keeping >= ~8 single-wave workgroups per CU
  int td = 1;
  if (D % 4 == 0 && groups / 4 >= 2048 && g->Ci == 4) td = 4;
  else if (D % 2 == 0 && groups / 2 >= 2048) td = 2;
  p.ok = true; p.C = g->Ci; p.td = td; p.dil = g->dil;
  // two waves per workgroup on an 8-row tile (shared halo): pays for dilation 2, whose +-2 halo makes the single-wave
  // tile 8 x 8 x 20 positions for 256 outputs (measured at (2,4,64^3): 12.8 -> 8.0 us); for dilation 1 it is neutral at
  // 64^3 and 4 % slower at 128^3, so those keep one wave per workgroup
  p.nw = (g->Ci == 4 && td == 4 && H % 8 == 0 && g->dil == 2) ? 2 : 1;
  p.tiles = (W / 16) * (H / (4 * p.nw)) * (D / td);
  const int Q = g->Ci / 4;
  const size_t pstride = ((size_t)(4 * p.nw + 2 * g->dil) * (16 + 2 * g->dil) + 63) / 64 * 64;
  p.lds = ((size_t)Q * (td + 2 * g->dil) * pstride + ((size_t)27 * g->Ci * Q + 63) / 64 * 64) * 16;
  return p;
}

template <int C, int TD, int DIL>
static int launch_vox_t(VxArgs& a, const VxPlan& p, int B, hipStream_t s) {
  a.tiles = p.tiles;
  a.fT = FastDiv((uint32_t)p.tiles); a.fTw = FastDiv((uint32_t)(a.W / 16)); a.fTh = FastDiv((uint32_t)(a.H / (4 * p.nw)));
  a.zero_page = zero_page_ptr();
  if (!a.zero_page) return 0;
  if constexpr (C == 4 && TD == 4) {
    if (p.nw == 2) { hipLaunchKernelGGL((conv_vox64_kernel<C, TD, DIL, 2>), dim3(p.tiles * B), dim3(128), p.lds, s, a); return 1; }
  }
  hipLaunchKernelGGL((conv_vox64_kernel<C, TD, DIL, 1>), dim3(p.tiles * B), dim3(64), p.lds, s, a);
  return 1;
}

template <int C>
static int launch_vox_c(VxArgs& a, const VxPlan& p, int B, hipStream_t s) {
  if (p.dil == 1) {
    if (p.td == 4) return launch_vox_t<C, 4, 1>(a, p, B, s);
    if (p.td == 2) return launch_vox_t<C, 2, 1>(a, p, B, s);
    return launch_vox_t<C, 1, 1>(a, p, B, s);
  }
  if (p.td == 4) return launch_vox_t<C, 4, 2>(a, p, B, s);
  if (p.td == 2) return launch_vox_t<C, 2, 2>(a, p, B, s);
  return launch_vox_t<C, 1, 2>(a, p, B, s);
}

// ------------------------------------------------------------------------------------------------
// Several INDEPENDENT single-wave convs of the vox family in one launch -- the plain-conv primitives of a supernet node at the C = 8
// level of 64^3 patches (cell.py:76-81: dil_conv / conv / down_conv / down_dil_conv on 2 x 8 x 32^3): each is 512-1024 one-wave
// workgroups, one per SIMD at best, so a launch lasts one fill + 216-432 MFMAs + one store whatever the chip has idle; four of them
// back to back cost 4 x 6.5 us, side by side ~9 (profiles/r05_search_table.log).  Workgroup L belongs to the job whose range
// [start[k], start[k+1]) holds it and runs that job's kernel BODY (vox64_body / vox_s2_body) on the job's own arguments: results, statistics
// rows and workspace use are exactly those of the single launches.  Only the one-plane tile forms (38-59 VGPRs) are folded: the
// deep-tile instantiations hold 130-190 registers and would set the occupancy of every job in the launch.
// ------------------------------------------------------------------------------------------------
struct VoxMultiArgs { VxArgs x[4]; Vs2Args s[4]; int kind[4]; int start[5]; };
enum { VOXK_S1_D1 = 0, VOXK_S1_D2 = 1, VOXK_S2_D1 = 2, VOXK_S2_D2 = 3, VOXK_S2_TD2_D1 = 4, VOXK_S2_TD2_D2 = 5 };

template <int C>
__global__ __launch_bounds__(64, 2) void conv_vox_multi_kernel(VoxMultiArgs q) {
  N3D_CHAIN_PRIO();
  extern __shared__ __attribute__((aligned(16))) float4 vlds[];
  const int L = blockIdx.x;
  const int k = (L >= q.start[1]) + (L >= q.start[2]) + (L >= q.start[3]);
  // the job's descriptor is copied out of the kernel arguments by a switch (static indexing), then ONE body per kind
  VxArgs xa; Vs2Args sa; int kind, s0, s1;
  switch (k) {
    case 0: xa = q.x[0]; sa = q.s[0]; kind = q.kind[0]; s0 = q.start[0]; s1 = q.start[1]; break;
    case 1: xa = q.x[1]; sa = q.s[1]; kind = q.kind[1]; s0 = q.start[1]; s1 = q.start[2]; break;
    case 2: xa = q.x[2]; sa = q.s[2]; kind = q.kind[2]; s0 = q.start[2]; s1 = q.start[3]; break;
    default: xa = q.x[3]; sa = q.s[3]; kind = q.kind[3]; s0 = q.start[3]; s1 = q.start[4]; break;
  }
  const int wg = L - s0, nwg = s1 - s0;
  switch (kind) {
    case VOXK_S1_D1: vox64_body<C, 1, 1, 1>(xa, wg, nwg, vlds); break;
    case VOXK_S1_D2: vox64_body<C, 1, 2, 1>(xa, wg, nwg, vlds); break;
    case VOXK_S2_D1: vox_s2_body<C, 1, 1>(sa, wg, nwg, vlds); break;
    case VOXK_S2_D2: vox_s2_body<C, 1, 2>(sa, wg, nwg, vlds); break;
    case VOXK_S2_TD2_D1: if constexpr (C == 4) vox_s2_body<4, 2, 1>(sa, wg, nwg, vlds); break;
    default: if constexpr (C == 4) vox_s2_body<4, 2, 2>(sa, wg, nwg, vlds); break;
  }
}

// one conv of a multi launch as the entry points hand it over (forward-type gather: data_grad as run_gather's)
struct VoxCall {
  const n3d_conv_geom* g; bool data_grad; const float* src; int64_t sld; const float* w; const float* bias; float* dst; int64_t dld; int flags;
  const float* in_gate; const float* relu_src; const float* out_gate; double* stats; void* ws; size_t ws_bytes;
};

// kind of the job, -1 = this conv is not one the multi launch folds (pure function of geometry, flags and alignment)
static int vox_multi_kind(const VoxCall& c, int* tiles, size_t* lds) {
  if (c.flags & (N3D_SRC_BF16 | N3D_DST_BF16 | N3D_NO_MFMA | N3D_RELU_IN)) return -1;
  if (c.in_gate || c.relu_src || c.out_gate || c.sld % 4 != 0 || c.dld % 4 != 0 || !aligned16(c.src) || !aligned16(c.dst)) return -1;
  const Vs2Plan v2 = vs2_plan(c.g, c.data_grad);
  if (v2.ok) {
    if (v2.td == 2 && v2.C != 4) return -1;
    *tiles = v2.tiles; *lds = v2.lds;
    return (v2.td == 2 ? VOXK_S2_TD2_D1 : VOXK_S2_D1) + (v2.dil == 2 ? 1 : 0);
  }
  if (vup_plan(c.g, c.data_grad).ok) return -1;
  const VxPlan v = vx_plan(c.g);
  if (!v.ok || v.td != 1 || v.nw != 1) return -1;
  *tiles = v.tiles; *lds = v.lds;
  return v.dil == 2 ? VOXK_S1_D2 : VOXK_S1_D1;
}

// n = 2..4 independent convs (distinct destinations) in one launch; 1 = launched, 0 = not foldable (nothing touched), < 0 error
int mfma_vox_multi_try(int n, const VoxCall* c, hipStream_t s) {
  if (n < 2 || n > 4) return 0;
  int kind[4], tiles[4];
  size_t lds = 0;
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    size_t l = 0;
    kind[i] = vox_multi_kind(c[i], &tiles[i], &l);
    if (kind[i] < 0 || c[i].g->Ci != c[0].g->Ci) return 0;
    if (!c[i].ws || c[i].ws_bytes < (size_t)27 * c[i].g->Ci * c[i].g->Ci * 4) return 0;
    for (int j = 0; j < i; ++j) {
      if (c[j].dst == c[i].dst) return 0;
      // ONE kernel reads every call's packed weights: two calls that pack into the same workspace (a C-ABI caller re-using one
      // scratch buffer, fine on the sequential path) would both see the last call's weights -> the sequential path
      if (c[j].ws == c[i].ws && !((c[j].flags & c[i].flags) & N3D_PREPACKED)) return 0;
    }
    if (l > lds) lds = l;
    total += (int64_t)tiles[i] * c[i].g->B;
  }
  if (total >= (1ll << 31)) return 0;
  const void* zp = zero_page_ptr();
  if (!zp) return 0;
  const int C = c[0].g->Ci;
  VoxMultiArgs q;
  int at = 0;
  for (int i = 0; i < 4; ++i) {
    const int k = i < n ? i : n - 1;      // unused slots repeat the last job (never selected: their range is empty)
    const VoxCall& cc = c[k];
    const n3d_conv_geom* g = cc.g;
    float* wq = (float*)cc.ws;
    const bool s2 = kind[k] >= VOXK_S2_D1;
    if (i < n && !(cc.flags & N3D_PREPACKED))
      hipLaunchKernelGGL(pack_vox_kernel, dim3((unsigned)cdiv(27 * C * C, 256)), dim3(256), 0, s, cc.w, wq, C, (!s2 && cc.data_grad) ? 1 : 0);
    VxArgs& x = q.x[i];
    x.src = cc.src; x.sld = cc.sld; x.dst = cc.dst; x.dld = cc.dld; x.wq = wq; x.bias = cc.bias; x.D = g->Di; x.H = g->Hi; x.W = g->Wi;
    x.flags = cc.flags; x.stats = cc.stats; x.rows_per_sample = tiles[k]; x.tiles = tiles[k]; x.zero_page = zp;
    x.fT = FastDiv((uint32_t)tiles[k]); x.fTw = FastDiv((uint32_t)(g->Wi / 16)); x.fTh = FastDiv((uint32_t)(g->Hi / 4));
    Vs2Args& v = q.s[i];
    v.src = cc.src; v.sld = cc.sld; v.D = g->Di; v.H = g->Hi; v.W = g->Wi; v.dst = cc.dst; v.dld = cc.dld; v.oD = g->Do; v.oH = g->Ho; v.oW = g->Wo;
    v.wq = wq; v.bias = cc.bias; v.flags = cc.flags; v.stats = cc.stats; v.rows_per_sample = tiles[k]; v.tiles = tiles[k]; v.zero_page = zp;
    v.fT = FastDiv((uint32_t)tiles[k]); v.fTw = FastDiv((uint32_t)(g->Wo / 16)); v.fTh = FastDiv((uint32_t)(g->Ho / 4));
    q.kind[i] = kind[k];
    q.start[i] = at;
    if (i < n) at += tiles[k] * g->B;
  }
  q.start[4] = at;
  if (C == 4) hipLaunchKernelGGL(conv_vox_multi_kernel<4>, dim3((unsigned)at), dim3(64), lds, s, q);
  else hipLaunchKernelGGL(conv_vox_multi_kernel<8>, dim3((unsigned)at), dim3(64), lds, s, q);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("conv(vox multi) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
  return 1;
}

// ------------------------------------------------------------------------------------------------
// vox64 weight gradient: 3x3x3 stride-1 (dilation 1/2) conv with C = 4 or 8:
//   dW[tap][ci][co] = sum_v X[v + (tap-1)*dil][ci] * dY[v][co].
// v_mfma_f32_4x4x1_16b_f32 again, now with the 16 blocks = 16 consecutive W voxels: block b multiplies the column
// X[v_b + tap][ci 0..3] by the row dY[v_b][co 0..3]; the hardware accumulates one 4x4 tile per block over
// successive issues (= over voxels), and the 16 block tiles are added with DPP class sums at the very end.
//   A operand: ds_read_b32 from the LDS halo tile of X (16 voxels x 16 B contiguous -> conflict free);
//   B operand: dY straight from HBM, one dword per lane and 16-voxel row, reused by all taps of the wave.
// The 27 taps are split over the 4 waves (7/7/7/6), so each wave keeps 7*Q*Q accumulator tiles (28 / 112 VGPRs);
// a workgroup walks several 4x4x16 tiles of one column before it reduces, which amortises the reduction and
// keeps the number of partial slabs (= workgroups) at a few hundred.  Compared with the VALU kernel this is ~8x
// fewer instructions per voxel, which is what matters at these sizes (a lone wave issues ~1 instruction / 4 cycles).
// ------------------------------------------------------------------------------------------------
struct VwArgs {
  const void* x; int64_t xld;       // T elements (fp32, or bf16 storage: one 16-byte LDS slot per voxel, operands widened on read)
  const void* dy; int64_t dyld;
  float* partial;       // [workgroup][27][C*C]
  int D, H, W, dchunk;  // dchunk: planes per workgroup (multiple of 4)
  const void* zero_page;
};

// Epilogue of the vox weight-gradient kernels: every wave holds 7 * QC * QC accumulator tiles acc[t][qa][qb] of the 4x4x1 MFMA --
// register r of lane (block b = lane >> 2, j = lane & 3) is the partial sum of dW[tap0 + t][ci = 4 qa + r][co = 4 qb + j] over the
// voxels block b saw -- and the slab wants the sum over the 16 blocks.  The four registers of a tile are reduced TOGETHER: a
// v_permlane32_swap of (r0, r1) puts the two wave halves of r0 side by side in the low half of the pair and those of r1 in the
// high half, so ONE add folds lane ^ 32 for both; the same with (r2, r3), then a v_permlane16_swap of the two results folds lane ^ 16
// for all four and leaves row 0 / 1 / 2 / 3 of the wave with r0 / r2 / r1 / r3; two DPP row rotations finish inside the rows.
// 3 swaps + 5 adds + 1 store of 16 lanes per tile instead of 4 x (2 swaps + 2 DPP + 4 adds + a 4-lane store): the epilogue was
// 3 (C = 4) / 7 us (C = 8) of these 12 us kernels (round-3 ablation: profiles/r03_wgrad_anatomy.log).
template <int C, int QC>
__device__ __forceinline__ void vw_store_tiles(const f32x4 (&acc)[7][QC][QC], float* __restrict__ out, const int tap0, const int lane,
                                               const int tap_stride = C * C) {
  const int rsel = ((lane >> 4) & 1) * 2 + (lane >> 5);     // the accumulator register this lane's row ends up with
  const bool writer = (lane & 15) < 4;
  float* o = out + rsel * C + (lane & 3);
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    if (tap0 + t < 27) {
#pragma unroll
      for (int qa = 0; qa < QC; ++qa)
#pragma unroll
        for (int qb = 0; qb < QC; ++qb) {
          // (scalars first: __builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler)
          const float v0 = acc[t][qa][qb][0], v1 = acc[t][qa][qb][1], v2 = acc[t][qa][qb][2], v3 = acc[t][qa][qb][3];
          const auto s01 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v0), __builtin_bit_cast(int, v1), false, false);
          const auto s23 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(int, v2), __builtin_bit_cast(int, v3), false, false);
          const float ab = __builtin_bit_cast(float, (int)s01[0]) + __builtin_bit_cast(float, (int)s01[1]);
          const float cd = __builtin_bit_cast(float, (int)s23[0]) + __builtin_bit_cast(float, (int)s23[1]);
          const auto sx = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(int, ab), __builtin_bit_cast(int, cd), false, false);
          float x = __builtin_bit_cast(float, (int)sx[0]) + __builtin_bit_cast(float, (int)sx[1]);
          x += dpp_f<0x124>(x);   // row_ror:4
          x += dpp_f<0x128>(x);   // row_ror:8
          if (writer) o[(tap0 + t) * tap_stride + (qa * 4) * C + qb * 4] = x;
        }
    }
  }
}

// wait until at most `n` of this wave's vector-memory operations are outstanding (they retire in order)
__device__ __forceinline__ void wait_vmcnt(const int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 36: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// NS: 0 = the workgroup walks a.dchunk / 4 tiles through two buffers (tile k+1 is requested when tile k has landed); NS = 1 / 2 / 4 =
// NS buffers, the tiles go in groups of NS that are requested together.  At 64^3 / 32^3 every workgroup of these launches is resident
// at once (a few hundred of them), so the kernel lasts as long as ONE workgroup's chain of memory round trips, and a tile's MFMAs
// (0.4 us) are far shorter than a round trip: with NS buffers the chain is a.dchunk / (4 NS) round trips instead of a.dchunk / 4.
typedef short vw_bf16x4 __attribute__((ext_vector_type(4)));
template <int C, int DIL, typename T = float, int NS = 0>
__global__ __launch_bounds__(256, 2) void vox_wgrad_kernel(VwArgs a) {
  constexpr bool B16 = sizeof(T) == 2;
  constexpr int QC = C / 4;                 // channel quads = accumulator tiles per side
  constexpr int Q = B16 ? 1 : QC;           // 16-byte LDS slots per voxel (a bf16 voxel of 4 or 8 channels takes one)
  constexpr int TD = 4, GH = 4, GW = 16;
  constexpr int LD = TD + 2 * DIL, LH = GH + 2 * DIL, LW = GW + 2 * DIL;
  constexpr int PLANE = LH * LW, NVOX = LD * PLANE;
  // One staged tile = the X halo tile ([Q][NVOX] float4) followed by the dY tile ([TD*GH rows][16 voxels][Q] float4), filled
  // by LDS-DMA in 1 KiB chunks that the four waves share; two such buffers, so the DMA of tile k+1 runs under the
  // MFMAs of tile k (ONE barrier per tile).  Both operands come from LDS: no ordinary global load sits in the loop,
  // whose wait would also drain the prefetch (VMEM counters retire in order).
  constexpr int NXC = (Q * NVOX + 63) / 64;            // chunks of the X tile
  constexpr int NYC = (TD * GH * GW * Q + 63) / 64;    // chunks of the dY tile
  constexpr int NCH = (NXC + NYC + 3) / 4 * 4;         // padded to a multiple of the 4 waves
  constexpr int BUF = NCH * 64;                        // float4 per buffer
  extern __shared__ __attribute__((aligned(16))) float4 wtile[];  // [2][BUF]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order of the workgroups (see conv_vox64_kernel); the slab index stays the physical workgroup id
  int wg = blockIdx.y * gridDim.x + blockIdx.x;
  {
    const int nwg = gridDim.x * gridDim.y, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  const int b = wg / (int)gridDim.x;
  const int tw_n = a.W / GW, th_n = a.H / GH;
  int bx = wg - b * (int)gridDim.x;
  const int w0 = (bx % tw_n) * GW; bx /= tw_n;
  const int h0 = (bx % th_n) * GH;
  const int dbeg = (bx / th_n) * a.dchunk;
  const int64_t N = (int64_t)a.D * a.H * a.W;
  const T* xb = reinterpret_cast<const T*>(a.x) + (int64_t)b * N * a.xld;
  const T* dyb = reinterpret_cast<const T*>(a.dy) + (int64_t)b * N * a.dyld;
  const float4* zp = reinterpret_cast<const float4*>(a.zero_page);
  const int blk = lane >> 2, i4 = lane & 3;
  const int tap0 = wave * 7;
  constexpr int EPS = B16 ? 8 : 4;          // LDS elements per slot
  int toff[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = (tap0 + t < 27) ? tap0 + t : 26;
    const int kw = tap % 3, kh = (tap / 3) % 3, kd = tap / 9;
    toff[t] = ((kd * DIL) * PLANE + (kh * DIL) * LW + kw * DIL + blk) * EPS + i4;
  }
  f32x4 acc[7][QC][QC];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int qa = 0; qa < QC; ++qa)
#pragma unroll
      for (int qb = 0; qb < QC; ++qb) acc[t][qa][qb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // stage tile `d0` into buffer `buf`: this wave's chunks are wave, wave+4, ...  Where a lane's slot of chunk m lies relative to
  // plane d0 does not depend on the tile, so the slot -> voxel arithmetic (divisions by the tile extents) is done ONCE per workgroup:
  // soff[m] = element offset from plane d0 of the chunk's tensor, sdz[m] = the slot's plane relative to d0 (a value that fails the
  // plane test for slots outside the volume in H / W and for padding slots).  Per tile and chunk that leaves a compare, a select and
  // a 64-bit add in front of the DMA -- the arithmetic was 3.9 of this kernel's 13.5 us at (2,4,64^3) and 26 of 83 us at (2,4,128^3)
  // (profiles/r03_wgrad_anatomy.log).
  constexpr int M = NCH / 4;
  int soff[M], sdz[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int c = m * 4 + wave;                       // uniform
    const int slot = c * 64 + lane;
    soff[m] = 0; sdz[m] = -(1 << 20);
    if (c < NXC) {
      // X tile: slot = q * NVOX + idx, idx = (dz * LH + hy) * LW + wx
      const int q = slot / NVOX, idx = slot - q * NVOX;
      const int wx = idx % LW, hy = (idx / LW) % LH, dz = idx / PLANE;
      const int gh = h0 - DIL + hy, gw = w0 - DIL + wx;
      if (slot < Q * NVOX && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W) {
        soff[m] = (((dz - DIL) * a.H + gh) * a.W + gw) * (int)a.xld + q * 4;   // bf16: 16 bytes from the voxel
        sdz[m] = dz - DIL;
      }
    } else {
      // dY tile: slot' = (row * 16 + voxel) * Q + q, row = g * GH + hh
      const int sl = slot - NXC * 64;
      const int q = sl % Q, vx = (sl / Q) % GW, row = sl / (Q * GW);
      if (row < TD * GH) {
        const int g = row / GH, hh = row - g * GH;
        soff[m] = ((g * a.H + h0 + hh) * a.W + w0 + vx) * (int)a.dyld + q * 4;
        sdz[m] = g;
      }
    }
  }
  const int64_t xplane = (int64_t)a.H * a.W * a.xld, yplane = (int64_t)a.H * a.W * a.dyld;
  auto stage = [&](int d0, float4* buf) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int c = m * 4 + wave;                       // uniform
      const bool isx = c < NXC;
      const T* base = (isx ? xb + (int64_t)d0 * xplane : dyb + (int64_t)d0 * yplane) + soff[m];
      const float4* srcp = (unsigned)(d0 + sdz[m]) < (unsigned)a.D ? reinterpret_cast<const float4*>(base) : zp;
      __builtin_amdgcn_global_load_lds((gptr_t)srcp, (lptr_t)(buf + c * 64), 16, 0, N3D_WGRAD_AUX);
    }
  };
  auto compute = [&](const float4* buf) {
    const float* tf = reinterpret_cast<const float*>(buf);
    const float* yf = tf + NXC * 64 * 4;
    const bf16_t* th = reinterpret_cast<const bf16_t*>(buf);
    const bf16_t* yh = th + NXC * 64 * 8;
    if constexpr (B16) {
      // bf16 storage (round 4): v_mfma_f32_4x4x4_16b_bf16 with K = the FOUR ROWS of an output plane.  Block b is still W voxel b; lane
      // (b, i) holds channel i of that voxel column in rows hh = 0..3 -- four 2-byte LDS reads packed into two registers, no widening
      // shift -- so one instruction does the work of four 4x4x1 issues.  The fp32-widening form was MFMA-issue bound like the fp32
      // kernel (SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVE_CYCLES = 0.89 at (2,4,128^3), 0.96 for fp32) with twice its LDS instructions and 50 %
      // more VALU: 63 us (d = 1) / 83 us (d = 2) against 59 / 59 for fp32 (profiles/r04_pmc_wgrad.json).
      auto pack4 = [](const bf16_t* p0, const int stride) {
        const uint32_t v0 = p0[0], v1 = p0[stride], v2 = p0[2 * stride], v3 = p0[3 * stride];
        return make_uint2(v0 | (v1 << 16), v2 | (v3 << 16));
      };
#pragma unroll
      for (int g = 0; g < TD; ++g) {
        // (requesting plane g + 1's operands ahead of plane g's MFMAs measured no better: 52.5 vs 50.1 us at (2,4,128^3))
        uint2 bq[QC], aq[7][QC];
#pragma unroll
        for (int qb = 0; qb < QC; ++qb) bq[qb] = pack4(yh + ((g * GH) * GW + blk) * 8 + qb * 4 + i4, GW * 8);
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa) aq[t][qa] = pack4(th + (g * PLANE) * EPS + toff[t] + qa * 4, LW * EPS);
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa)
#pragma unroll
            for (int qb = 0; qb < QC; ++qb)
              acc[t][qa][qb] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(vw_bf16x4, aq[t][qa]), __builtin_bit_cast(vw_bf16x4, bq[qb]),
                                                                      acc[t][qa][qb], 0, 0, 0);
      }
      return;
    }
    if constexpr (!B16) {
      // fp32: the operands of row r + PF are requested before the MFMAs of row r are issued (three register sets, rotation resolved
      // by the full unroll).  hipcc's own schedule asked for a row's operands 3 - 7 MFMAs ahead of their use and then sat on
      // s_waitcnt lgkmcnt(0) (78 VGPRs of the 256 two workgroups per compute unit leave a wave); with two waves per SIMD the LDS
      // latency was not covered.
      constexpr int PF = 2, NR = TD * GH;
      float avs[PF + 1][7][QC], bvs[PF + 1][QC];
      auto load_row = [&](const int r, float (&av)[7][QC], float (&bv)[QC]) {
        const int rbase = ((r >> 2) * PLANE + (r & 3) * LW) * EPS;
#pragma unroll
        for (int qb = 0; qb < QC; ++qb) bv[qb] = yf[((r * GW + blk) * Q + qb) * 4 + i4];
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa) av[t][qa] = tf[qa * NVOX * 4 + rbase + toff[t]];
      };
#pragma unroll
      for (int r = 0; r < PF; ++r) load_row(r, avs[r], bvs[r]);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r + PF < NR) load_row(r + PF, avs[(r + PF) % (PF + 1)], bvs[(r + PF) % (PF + 1)]);
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa)
#pragma unroll
            for (int qb = 0; qb < QC; ++qb)
              acc[t][qa][qb] = __builtin_amdgcn_mfma_f32_4x4x1f32(avs[r % (PF + 1)][t][qa], bvs[r % (PF + 1)][qb], acc[t][qa][qb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    // ---- 16 rows of 16 voxels: row r = (g, hh) -> output plane d0+g, row h0+hh
#pragma unroll
    for (int r = 0; r < TD * GH; ++r) {
      const int g = r >> 2, hh = r & 3;
      const int rbase = (g * PLANE + hh * LW) * EPS;
      float avs[7][QC], bvs[QC];
      if constexpr (B16) {
        // one slot per voxel, channel c at element c of the slot; widened to fp32 (a shift) on the way to the MFMA
#pragma unroll
        for (int qb = 0; qb < QC; ++qb) bvs[qb] = ld1(yh + (r * GW + blk) * 8 + qb * 4 + i4);
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa) avs[t][qa] = ld1(th + rbase + toff[t] + qa * 4);
      } else {
#pragma unroll
        for (int qb = 0; qb < QC; ++qb) bvs[qb] = yf[((r * GW + blk) * Q + qb) * 4 + i4];
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa) avs[t][qa] = tf[qa * NVOX * 4 + rbase + toff[t]];
      }
#pragma unroll
      for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int qa = 0; qa < QC; ++qa)
#pragma unroll
          for (int qb = 0; qb < QC; ++qb) acc[t][qa][qb] = __builtin_amdgcn_mfma_f32_4x4x1f32(avs[t][qa], bvs[qb], acc[t][qa][qb], 0, 0, 0);
    }
  };
  if constexpr (NS > 0) {
    // NS tiles in flight at once; tile k is complete when all but the (NS - 1 - k) * NCH / 4 youngest requests of every wave are
    const int ntile = a.dchunk / TD;           // a multiple of NS (host)
    for (int t0 = 0; t0 < ntile; t0 += NS) {
      if (t0) __syncthreads();                 // everyone is done with the buffers of the previous group
#pragma unroll
      for (int k = 0; k < NS; ++k) stage(dbeg + (t0 + k) * TD, wtile + k * BUF);
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        wait_vmcnt((NS - 1 - k) * (NCH / 4));
        __syncthreads();
        compute(wtile + k * BUF);
      }
    }
  } else {
    const int ntile = a.dchunk / TD;
    stage(dbeg, wtile);
    for (int k = 0; k < ntile; ++k) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // tile k landed for every wave; everyone is done with the buffer tile k+1 is about to overwrite
      if (k + 1 < ntile) stage(dbeg + (k + 1) * TD, wtile + ((k + 1) & 1) * BUF);
      compute(wtile + (k & 1) * BUF);
    }
  }
  // ---- add the 16 block tiles (lanes with equal lane&3) and write this workgroup's slab
  float* out = a.partial + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 27) * (C * C);
  vw_store_tiles<C, QC>(acc, out, tap0, lane);
}

// ------------------------------------------------------------------------------------------------
// vox_wgrad_s2: weight gradient of the STRIDE-2 3x3x3 convs with C = 4 / 8 (and, with the operand roles swapped by the
// caller, of the stride-2 transposed convs):  dW[tap][ci][co] = sum_o X[2*o - pad + tap*dil][ci] * dY[o][co].
// Same scheme as vox_wgrad_kernel (4x4x1 MFMA, taps split over 4 waves, double-buffered LDS-DMA staging); the tile
// walks TD = 2 planes of the SMALL grid, the X region is (3 + 2*dil) x (7 + 2*dil) x (31 + 2*dil) voxels of the big grid,
// stored with every row de-interleaved by W parity so that the 16 blocks of an MFMA read 16 consecutive voxels.
// The generic kernel gave every tap its own workgroups, i.e. re-read dY 27 times.
// ------------------------------------------------------------------------------------------------
struct Vw2Args {
  const void* x; int64_t xld; int D, H, W;         // big grid (conv input / transposed-conv output gradient); T elements
  const void* dy; int64_t dyld; int oD, oH, oW;    // small grid
  float* partial; int dchunk; const void* zero_page;
  int tco;   // output-channel tiles of C channels (grid.z): dY has tco * C channels, workgroup z takes channels [z C, z C + C) -- the
             // stem's 4 -> 12 conv as three 4 -> 4 problems in one launch; slabs [workgroup][tap][z][C][C] (final job: tco tiles per tap)
};

// T / TY: storage types of X and dY (TY = T except for the stem in the bf16 configuration: fp32 net input, bf16 gradient)
template <int C, int DIL, typename T = float, typename TY = T>
__global__ __launch_bounds__(256, 2) void vox_wgrad_s2_kernel(Vw2Args a) {
  constexpr bool B16 = sizeof(T) == 2, BY16 = sizeof(TY) == 2;
  constexpr int QC = C / 4, Q = B16 ? 1 : QC, EPS = B16 ? 8 : 4;   // accumulator quads; LDS slots per voxel; elements per slot (X image)
  constexpr int QY = BY16 ? 1 : QC;                                 // LDS slots per voxel of the dY image
  constexpr int TD = 2, GH = 4, GW = 16;
  constexpr int LD = 2 * (TD - 1) + 2 * DIL + 1, LH = 7 + 2 * DIL, LW = 31 + 2 * DIL;
  constexpr int HW = (LW + 1) / 2, RW = 2 * HW, PLANE = LH * RW, NVOX = LD * PLANE;
  constexpr int NXC = (Q * NVOX + 63) / 64, NYC = (TD * GH * GW * QY + 63) / 64, NCH = (NXC + NYC + 3) / 4 * 4, BUF = NCH * 64;
  extern __shared__ __attribute__((aligned(16))) float4 wtile[];  // [2][BUF]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wg = blockIdx.y * gridDim.x + blockIdx.x;
  {
    const int nwg = gridDim.x * gridDim.y, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  const int b = wg / (int)gridDim.x;
  const int tw_n = a.oW / GW, th_n = a.oH / GH;
  int bx = wg - b * (int)gridDim.x;
  const int w0 = (bx % tw_n) * GW; bx /= tw_n;
  const int h0 = (bx % th_n) * GH;
  const int dbeg = (bx / th_n) * a.dchunk;
  const int64_t Nx = (int64_t)a.D * a.H * a.W, Ny = (int64_t)a.oD * a.oH * a.oW;
  const T* xb = reinterpret_cast<const T*>(a.x) + (int64_t)b * Nx * a.xld;
  const TY* dyb = reinterpret_cast<const TY*>(a.dy) + (int64_t)b * Ny * a.dyld + (int)blockIdx.z * C;
  const float4* zp = reinterpret_cast<const float4*>(a.zero_page);
  const int blk = lane >> 2, i4 = lane & 3;
  const int tap0 = wave * 7;
  int toff[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int tap = (tap0 + t < 27) ? tap0 + t : 26;
    const int kw = tap % 3, kh = (tap / 3) % 3, kd = tap / 9;
    // X voxel (2g + kd*DIL, 2hh + kh*DIL, 2blk + kw*DIL): W parity and half index
    toff[t] = ((kd * DIL) * PLANE + (kh * DIL) * RW + ((kw * DIL) & 1) * HW + ((kw * DIL) >> 1) + blk) * EPS + i4;
  }
  f32x4 acc[7][QC][QC];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int qa = 0; qa < QC; ++qa)
#pragma unroll
      for (int qb = 0; qb < QC; ++qb) acc[t][qa][qb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  // slot -> voxel arithmetic once per workgroup (see vox_wgrad_kernel): soff = element offset from plane 2 d0 of X / plane d0 of dY,
  // sdz = the slot's plane relative to that (a value that fails the plane test for slots outside the volume in H / W and padding)
  constexpr int M = NCH / 4;
  constexpr bool PRE = M <= 16;       // (C = 8, dilation 2, fp32: 23 chunks per wave -- the two tables would spill; arithmetic per tile there)
  int soff[PRE ? M : 1], sdz[PRE ? M : 1];
  if constexpr (PRE) {
    const int ih0 = 2 * h0 - DIL, iw0 = 2 * w0 - DIL;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int c = m * 4 + wave;
      const int slot = c * 64 + lane;
      soff[m] = 0; sdz[m] = -(1 << 20);
      if (c < NXC) {
        // X tile: slot = q * NVOX + (dz * LH + row) * RW + par * HW + half
        const int q = slot / NVOX, idx = slot - q * NVOX;
        const int dz = idx / PLANE, rem = idx - dz * PLANE;
        const int row = rem / RW, r2 = rem - row * RW;
        const int par = r2 / HW, half = r2 - par * HW;
        const int wx = 2 * half + par;
        const int gh = ih0 + row, gw = iw0 + wx;
        if (slot < Q * NVOX && wx < LW && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W) {
          soff[m] = (((dz - DIL) * a.H + gh) * a.W + gw) * (int)a.xld + q * 4;
          sdz[m] = dz - DIL;
        }
      } else {
        const int sl = slot - NXC * 64;
        const int q = sl % QY, vx = (sl / QY) % GW, row = sl / (QY * GW);
        if (row < TD * GH) {
          const int g = row / GH, hh = row - g * GH;
          soff[m] = ((g * a.oH + h0 + hh) * a.oW + w0 + vx) * (int)a.dyld + q * 4;
          sdz[m] = g;
        }
      }
    }
  }
  const int64_t xplane = (int64_t)a.H * a.W * a.xld, yplane = (int64_t)a.oH * a.oW * a.dyld;
  auto stage = [&](int d0, float4* buf) {
    if constexpr (PRE) {
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const int c = m * 4 + wave;
        const bool isx = c < NXC;
        const void* base = isx ? static_cast<const void*>(xb + (int64_t)(2 * d0) * xplane + soff[PRE ? m : 0])
                               : static_cast<const void*>(dyb + (int64_t)d0 * yplane + soff[PRE ? m : 0]);
        const bool inb = isx ? (unsigned)(2 * d0 + sdz[PRE ? m : 0]) < (unsigned)a.D : sdz[PRE ? m : 0] >= 0;
        const float4* srcp = inb ? reinterpret_cast<const float4*>(base) : zp;
        __builtin_amdgcn_global_load_lds((gptr_t)srcp, (lptr_t)(buf + c * 64), 16, 0, N3D_WGRAD_AUX);
      }
    } else {
      const int id0 = 2 * d0 - DIL, ih0 = 2 * h0 - DIL, iw0 = 2 * w0 - DIL;
#pragma unroll 1
      for (int m = 0; m < M; ++m) {
        const int c = m * 4 + wave;
        const int slot = c * 64 + lane;
        const float4* srcp = zp;
        if (c < NXC) {
          const int q = slot / NVOX, idx = slot - q * NVOX;
          const int dz = idx / PLANE, rem = idx - dz * PLANE;
          const int row = rem / RW, r2 = rem - row * RW;
          const int par = r2 / HW, half = r2 - par * HW;
          const int wx = 2 * half + par;
          const int gd = id0 + dz, gh = ih0 + row, gw = iw0 + wx;
          const bool inb = slot < Q * NVOX && wx < LW && gd >= 0 && gd < a.D && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
          if (inb) srcp = reinterpret_cast<const float4*>(xb + (((int64_t)gd * a.H + gh) * a.W + gw) * a.xld + q * 4);
        } else {
          const int sl = slot - NXC * 64;
          const int q = sl % QY, vx = (sl / QY) % GW, row = sl / (QY * GW);
          if (row < TD * GH) {
            const int g = row / GH, hh = row - g * GH;
            srcp = reinterpret_cast<const float4*>(dyb + (((int64_t)(d0 + g) * a.oH + h0 + hh) * a.oW + w0 + vx) * a.dyld + q * 4);
          }
        }
        __builtin_amdgcn_global_load_lds((gptr_t)srcp, (lptr_t)(buf + c * 64), 16, 0, N3D_WGRAD_AUX);
      }
    }
  };
  const int ntile = a.dchunk / TD;
  stage(dbeg, wtile);
  for (int k = 0; k < ntile; ++k) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (k + 1 < ntile) stage(dbeg + (k + 1) * TD, wtile + ((k + 1) & 1) * BUF);
    const float* tf = reinterpret_cast<const float*>(wtile + (k & 1) * BUF);
    const float* yf = tf + NXC * 64 * 4;
    const bf16_t* th = reinterpret_cast<const bf16_t*>(wtile + (k & 1) * BUF);
    const bf16_t* yh = th + NXC * 64 * 8;
    if constexpr (B16 && BY16) {
      // both images bf16: the four rows of an output plane are K of v_mfma_f32_4x4x4_16b_bf16 (see vox_wgrad_kernel)
      auto pack4 = [](const bf16_t* p0, const int stride) {
        const uint32_t v0 = p0[0], v1 = p0[stride], v2 = p0[2 * stride], v3 = p0[3 * stride];
        return make_uint2(v0 | (v1 << 16), v2 | (v3 << 16));
      };
#pragma unroll
      for (int g = 0; g < TD; ++g) {
        uint2 bq[QC], aq[7][QC];
#pragma unroll
        for (int qb = 0; qb < QC; ++qb) bq[qb] = pack4(yh + ((g * GH) * GW + blk) * 8 + qb * 4 + i4, GW * 8);
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa) aq[t][qa] = pack4(th + ((2 * g) * PLANE) * EPS + toff[t] + qa * 4, 2 * RW * EPS);
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa)
#pragma unroll
            for (int qb = 0; qb < QC; ++qb)
              acc[t][qa][qb] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(vw_bf16x4, aq[t][qa]), __builtin_bit_cast(vw_bf16x4, bq[qb]),
                                                                      acc[t][qa][qb], 0, 0, 0);
      }
      continue;
    }
    if constexpr (!B16 && !BY16) {
      // fp32: operands of row r + 2 requested before the MFMAs of row r (see vox_wgrad_kernel)
      constexpr int PF = 2, NR = TD * GH;
      float avs[PF + 1][7][QC], bvs[PF + 1][QC];
      auto load_row = [&](const int r, float (&av)[7][QC], float (&bv)[QC]) {
        const int rbase = ((2 * (r >> 2)) * PLANE + (2 * (r & 3)) * RW) * EPS;
#pragma unroll
        for (int qb = 0; qb < QC; ++qb) bv[qb] = yf[((r * GW + blk) * QY + qb) * 4 + i4];
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa) av[t][qa] = tf[qa * NVOX * 4 + rbase + toff[t]];
      };
#pragma unroll
      for (int r = 0; r < PF; ++r) load_row(r, avs[r], bvs[r]);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r + PF < NR) load_row(r + PF, avs[(r + PF) % (PF + 1)], bvs[(r + PF) % (PF + 1)]);
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa)
#pragma unroll
            for (int qb = 0; qb < QC; ++qb)
              acc[t][qa][qb] = __builtin_amdgcn_mfma_f32_4x4x1f32(avs[r % (PF + 1)][t][qa], bvs[r % (PF + 1)][qb], acc[t][qa][qb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < TD * GH; ++r) {
      const int g = r >> 2, hh = r & 3;
      const int rbase = ((2 * g) * PLANE + (2 * hh) * RW) * EPS;
      float avs[7][QC], bvs[QC];
      if constexpr (BY16) {
#pragma unroll
        for (int qb = 0; qb < QC; ++qb) bvs[qb] = ld1(yh + (r * GW + blk) * 8 + qb * 4 + i4);
      } else {
#pragma unroll
        for (int qb = 0; qb < QC; ++qb) bvs[qb] = yf[((r * GW + blk) * QY + qb) * 4 + i4];
      }
      if constexpr (B16) {
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa) avs[t][qa] = ld1(th + rbase + toff[t] + qa * 4);
      } else {
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
          for (int qa = 0; qa < QC; ++qa) avs[t][qa] = tf[qa * NVOX * 4 + rbase + toff[t]];
      }
#pragma unroll
      for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int qa = 0; qa < QC; ++qa)
#pragma unroll
          for (int qb = 0; qb < QC; ++qb) acc[t][qa][qb] = __builtin_amdgcn_mfma_f32_4x4x1f32(avs[t][qa], bvs[qb], acc[t][qa][qb], 0, 0, 0);
    }
  }
  float* out = a.partial + (((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 27) * a.tco + blockIdx.z) * (C * C);
  vw_store_tiles<C, QC>(acc, out, tap0, lane, a.tco * C * C);
}

// returns 1 if handled (slab layout as vox_wgrad_try: ci_t = co_t = C, tci = 1, ntiles = 27 * tco; tco = Co / Ci = 1 except for the
// stem's 4 -> 12 conv)
int vox_wgrad_s2_try(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, int flags, const float* in_gate,
                     float* partial, size_t avail_floats, int* nchunks_out, hipStream_t s) {
  if ((flags & (N3D_RELU_IN | N3D_NO_MFMA)) || in_gate) return 0;
  if (g->depthwise || g->k != 3 || g->stride != 2 || (g->Ci != 4 && g->Ci != 8)) return 0;
  // square, or (fp32) 4 -> 8 / 12 / 16 channels as Co / 4 problems of 4 -> 4 in one launch (stem1, nas.py:29 / searched.py:70)
  const int tco = g->Co / g->Ci;
  constexpr bool no_cotile = false;
  // (the stem reads the fp32 net input also in the bf16 configuration: X fp32, dY fp32 or bf16)
  if (g->Co != g->Ci && (no_cotile || !(g->Ci == 4 && g->Co % 4 == 0 && tco >= 2 && tco <= 4 && !(flags & N3D_SRC_BF16)))) return 0;
  if (!(g->dil == 1 || g->dil == 2) || g->pad != g->dil) return 0;
  if (g->Wo % 16 != 0 || g->Ho % 4 != 0 || g->Do % 2 != 0) return 0;
  // small problems (16^3 outputs) leave this tile scheme with a few dozen long-running workgroups: the per-tap generic
  // kernel is faster there (measured 8.7 vs 12.7 us at (2,8,16^3)); from 32^3 outputs on it is 2x faster and reads dY once
  if ((int64_t)g->B * g->Do * g->Ho * g->Wo < 32768) return 0;
  const bool b16 = (flags & N3D_SRC_BF16) && (flags & N3D_DST_BF16);
  const bool mixed = !(flags & N3D_SRC_BF16) && (flags & N3D_DST_BF16) && g->Ci == 4;   // fp32 X, bf16 dY (C = 4 only: the stem)
  if (!b16 && !mixed && (flags & (N3D_SRC_BF16 | N3D_DST_BF16))) return 0;   // other mixed storage: the generic kernel
  if (xld % 4 != 0 || dyld % 4 != 0) return 0;
  if (mixed && (!aligned16(x) || (reinterpret_cast<uintptr_t>(dy) & 7) != 0)) return 0;
  if (!b16 && !mixed && (!aligned16(x) || !aligned16(dy))) return 0;
  if (b16 && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 7) != 0 ||
              (g->Ci == 8 && (xld % 8 != 0 || dyld % 8 != 0 || !aligned16(x) || !aligned16(dy))))) return 0;
  const int columns = g->B * (g->Ho / 4) * (g->Wo / 16);
  int nd = g->Do / 2, dsplit = 1;
  while (columns * dsplit < 384 && dsplit * 2 <= nd && nd % (dsplit * 2) == 0) dsplit *= 2;
  const int tiles = (g->Wo / 16) * (g->Ho / 4) * dsplit;
  const int nwg = tiles * g->B;
  if ((size_t)nwg * 27 * g->Ci * g->Co > avail_floats) return 0;
  const size_t Qn = b16 ? 1 : g->Ci / 4;   // LDS slots per voxel (X image; dY: one slot per voxel in bf16)
  const size_t Qy = (b16 || mixed) ? 1 : g->Ci / 4;
  const size_t LDn = 2 + 2 * g->dil + 1, LHn = 7 + 2 * g->dil, LWn = 31 + 2 * g->dil;
  const size_t nvox = LDn * LHn * 2 * ((LWn + 1) / 2);
  const size_t nxc = (Qn * nvox + 63) / 64, nyc = (2 * 4 * 16 * Qy + 63) / 64, nch = (nxc + nyc + 3) / 4 * 4;
  const size_t lds = 2 * nch * 64 * 16;
  if (lds > 160 * 1024) return 0;
  Vw2Args a;
  a.x = x; a.xld = xld; a.D = g->Di; a.H = g->Hi; a.W = g->Wi; a.dy = dy; a.dyld = dyld; a.oD = g->Do; a.oH = g->Ho; a.oW = g->Wo;
  a.partial = partial; a.dchunk = g->Do / dsplit; a.zero_page = zero_page_ptr(); a.tco = tco;
  if (!a.zero_page) return 0;
  dim3 grid(tiles, g->B, tco);
  if (mixed) {
    if (g->dil == 1) hipLaunchKernelGGL((vox_wgrad_s2_kernel<4, 1, float, bf16_t>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((vox_wgrad_s2_kernel<4, 2, float, bf16_t>), grid, dim3(256), lds, s, a);
  } else if (b16) {
    if (g->Ci == 4) {
      if (g->dil == 1) hipLaunchKernelGGL((vox_wgrad_s2_kernel<4, 1, bf16_t>), grid, dim3(256), lds, s, a);
      else hipLaunchKernelGGL((vox_wgrad_s2_kernel<4, 2, bf16_t>), grid, dim3(256), lds, s, a);
    } else {
      if (g->dil == 1) hipLaunchKernelGGL((vox_wgrad_s2_kernel<8, 1, bf16_t>), grid, dim3(256), lds, s, a);
      else hipLaunchKernelGGL((vox_wgrad_s2_kernel<8, 2, bf16_t>), grid, dim3(256), lds, s, a);
    }
  } else if (g->Ci == 4) {
    if (g->dil == 1) hipLaunchKernelGGL((vox_wgrad_s2_kernel<4, 1>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((vox_wgrad_s2_kernel<4, 2>), grid, dim3(256), lds, s, a);
  } else {
    if (g->dil == 1) hipLaunchKernelGGL((vox_wgrad_s2_kernel<8, 1>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((vox_wgrad_s2_kernel<8, 2>), grid, dim3(256), lds, s, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("conv(vox_wgrad_s2) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
  *nchunks_out = nwg;
  return 1;
}

struct VwPlan { bool ok; int C, dil, dchunk, tiles, ns; size_t lds; };

// dynamic LDS of vox_wgrad_kernel with `slots` 16-byte LDS slots per voxel (fp32: C / 4; bf16 storage: 1)
static size_t vw_stage_lds(int slots, int dil) {
  const size_t Qn = slots, nvox = (size_t)(4 + 2 * dil) * (4 + 2 * dil) * (16 + 2 * dil);
  const size_t nxc = (Qn * nvox + 63) / 64, nyc = (4 * 4 * 16 * Qn + 63) / 64, nch = (nxc + nyc + 3) / 4 * 4;
  return nch * 64 * 16;       // one staged tile (X halo tile + dY tile)
}
// tiles requested together (NS of vox_wgrad_kernel) for a workgroup of `ntile` tiles: as many as 64 KiB of LDS hold (two workgroups
// per compute unit), 0 = the two-buffer walk
static int vw_stages(int ntile, size_t stage_bytes) {
  constexpr int forced = -1;
  if (forced == 0) return 0;
  // (a workgroup with ONE tile runs the two-buffer form: measured 8.8 vs 10.6 us at (2,8,32^3) for the same work)
  for (int ns : {4, 2}) {
    if (forced > 0 && ns != forced) continue;
    if (ntile % ns == 0 && (size_t)ns * stage_bytes <= 64 * 1024) return ns;
  }
  return (forced == 1 && (size_t)stage_bytes <= 64 * 1024) ? 1 : 0;
}
static size_t vw_plan_lds(int slots, int dil, int ns) { return (size_t)(ns ? ns : 2) * vw_stage_lds(slots, dil); }

static VwPlan vw_plan(const n3d_conv_geom* g) {
  VwPlan p; p.ok = false;
  if (g->depthwise || g->k != 3 || g->stride != 1 || g->Ci != g->Co || (g->Ci != 4 && g->Ci != 8)) return p;
  if (!(g->dil == 1 || g->dil == 2) || g->pad != g->dil) return p;
  const int W = g->Wi, H = g->Hi, D = g->Di;
  if (W % 16 != 0 || H % 4 != 0 || D % 4 != 0) return p;
  const int columns = g->B * (H / 4) * (W / 16);
  int nd = D / 4, dsplit = 1;
  constexpr int want_wgs = 384;
  while (columns * dsplit < want_wgs && dsplit * 2 <= nd && nd % (dsplit * 2) == 0) dsplit *= 2;
  p.ok = true; p.C = g->Ci; p.dil = g->dil; p.dchunk = D / dsplit;
  p.tiles = (W / 16) * (H / 4) * dsplit;
  p.ns = 0; p.lds = 0;        // settled by the caller (the LDS image depends on the storage type)
  return p;
}

// returns 1 if handled: partial slabs laid out for conv_wgrad_final (ci_t = co_t = C, tci = tco = 1, ntiles = 27)
int vox_wgrad_try(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, int flags, const float* in_gate,
                  float* partial, size_t avail_floats, int* nchunks_out, hipStream_t s) {
  if ((flags & (N3D_RELU_IN | N3D_NO_MFMA)) || in_gate) return 0;
  const bool b16 = (flags & N3D_SRC_BF16) && (flags & N3D_DST_BF16);
  if (!b16 && (flags & (N3D_SRC_BF16 | N3D_DST_BF16))) return 0;   // mixed storage: the generic kernel
  VwPlan p = vw_plan(g);
  if (!p.ok || xld % 4 != 0 || dyld % 4 != 0 || (reinterpret_cast<uintptr_t>(x) & (b16 ? 7 : 15)) != 0) return 0;
  if (b16 && ((reinterpret_cast<uintptr_t>(dy) & 7) != 0 || (p.C == 8 && (xld % 8 != 0 || dyld % 8 != 0 || !aligned16(x) || !aligned16(dy))))) return 0;
  const int nwg = p.tiles * g->B;
  if ((size_t)nwg * 27 * p.C * p.C > avail_floats) return 0;
  VwArgs a;
  a.x = x; a.xld = xld; a.dy = dy; a.dyld = dyld; a.partial = partial; a.D = g->Di; a.H = g->Hi; a.W = g->Wi; a.dchunk = p.dchunk;
  a.zero_page = zero_page_ptr();
  if (!a.zero_page) return 0;
  dim3 grid(p.tiles, g->B);
  // bf16 storage: the LDS image is the fp32 C = 4 one (one slot per voxel) for both channel counts
  const int slots = b16 ? 1 : p.C / 4;
  const int ns = vw_stages(p.dchunk / 4, vw_stage_lds(slots, p.dil));
  const size_t lds = vw_plan_lds(slots, p.dil, ns);
#define N3D_VW_T(C_, D_, T_) do { \
    if (ns == 4) hipLaunchKernelGGL((vox_wgrad_kernel<C_, D_, T_, 4>), grid, dim3(256), lds, s, a); \
    else if (ns == 2) hipLaunchKernelGGL((vox_wgrad_kernel<C_, D_, T_, 2>), grid, dim3(256), lds, s, a); \
    else if (ns == 1) hipLaunchKernelGGL((vox_wgrad_kernel<C_, D_, T_, 1>), grid, dim3(256), lds, s, a); \
    else hipLaunchKernelGGL((vox_wgrad_kernel<C_, D_, T_, 0>), grid, dim3(256), lds, s, a); } while (0)
#define N3D_VW(T_) do { \
    if (p.C == 4) { if (p.dil == 1) N3D_VW_T(4, 1, T_); else N3D_VW_T(4, 2, T_); } \
    else { if (p.dil == 1) N3D_VW_T(8, 1, T_); else N3D_VW_T(8, 2, T_); } } while (0)
  if (b16) N3D_VW(bf16_t); else N3D_VW(float);
#undef N3D_VW
#undef N3D_VW_T
  *nchunks_out = nwg;
  return 1;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// tile16: the gemm16 problem with MANY voxels -- 3x3x3 stride-1 conv (dilation 1 / 2) with 16 -> 16 channels on >= 16k
// voxels (the C = 16 level of 128^3 patches: 2 x 32^3).  gemm16 gathers its A operand from global memory once per tap
// (27 x 64 bytes per voxel through L1/L2: 21 % of the fp32 MFMA peak at M = 65536); here a workgroup stages the halo tile
// of its 2 x 4 x 16 output voxels in LDS once (LDS-DMA, 64-byte voxel records) and every tap is one ds_read_b128 per
// 16-voxel row: lane (m = lane&15, kk = lane>>4) reads channels 4kk..4kk+3 of voxel m = the A operand of four
// v_mfma_f32_16x16x4_f32 steps, 1 KiB contiguous per wave-instruction (conflict-free).  The B operand (27 float4 per
// lane, the gemm16 packed layout Wp[tap][kk][cd][j]) lives in registers for the whole tile; an input gate scales it
// once per workgroup instead of the A values.  Same arguments, same epilogue (bias, ReLU mask, output gate,
// accumulate), same statistics rows (one per 128 voxels) as the gemm16 plan it replaces.
//   DG: data gradient = the same conv with mirrored taps (weights packed transposed by the pack kernel).
// ------------------------------------------------------------------------------------------------
template <int DIL, bool DG, bool BF>
__device__ __forceinline__ void tile16_body(const MfArgs& a, const void* zero_page, int tiles, const FastDiv& fT, const FastDiv& fTw, const FastDiv& fTh,
                                            float4* const t16, double* const red) {
  N3D_CHAIN_PRIO();
  constexpr int TD = 2, TH = 4, TW = 16;
  constexpr int LD = TD + 2 * DIL, LH = TH + 2 * DIL, LW = TW + 2 * DIL, NV = LD * LH * LW;
  constexpr int NP = NV * 4, NIT = (NP + 255) / 256;   // float4 pieces of the halo tile, DMA instructions per thread
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kk = lane >> 4;
  int wg = blockIdx.x;
  {  // XCD-aware placement (see conv_vox64_kernel): every XCD takes one contiguous run of tiles
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  uint32_t ub, utile, ubx, uw, ud, uh;       // (by tiles per sample, tile columns, tile rows: no run-time divisions, see VxArgs)
  fT.divmod((uint32_t)wg, ub, utile);
  fTw.divmod(utile, ubx, uw);
  fTh.divmod(ubx, ud, uh);
  const int b = (int)ub, tid = (int)utile;
  const int D = a.Dd, H = a.Hd, W = a.Wd;
  const int w0 = (int)uw * TW, h0 = (int)uh * TH, d0 = (int)ud * TD;
  const int lz = wave >> 1, ly0 = (wave & 1) * 2;   // this wave: plane lz, rows ly0 and ly0 + 1 of the tile
  const int64_t N = (int64_t)D * H * W;
  const bool accum = a.flags & N3D_ACCUMULATE;

  // epilogue operands first (ordinary loads, in flight behind the fill)
  const float e_bias = a.bias ? a.bias[m] : 0.f;
  const float e_gate = a.out_gate ? a.out_gate[(int64_t)b * 16 + m] : 1.f;
  float e_relu[2][4], e_prev[2][4];
  int64_t orow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    orow[t] = (int64_t)b * N + ((int64_t)(d0 + lz) * H + h0 + ly0 + t) * W + w0 + kk * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      e_relu[t][r] = a.relu_src ? a.relu_src[(orow[t] + r) * a.rld + m] : 1.f;
      e_prev[t][r] = accum ? a.dst[(orow[t] + r) * a.dld + m] : 0.f;
    }
  }
  // weights: Wp[tap][kk][cd = m][j], one float4 per lane and tap
  float4 bv[27];
  {
    const float4* __restrict__ wp4 = reinterpret_cast<const float4*>(a.wp);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) bv[tap] = wp4[(tap * 4 + kk) * 16 + m];
  }
  float4 gq = make_float4(1.f, 1.f, 1.f, 1.f);
  if (a.in_gate) gq = *reinterpret_cast<const float4*>(a.in_gate + (int64_t)b * 16 + kk * 4);

  // ---- halo tile by LDS-DMA: instruction i of wave w fills pieces [(4i + w) * 64, +64) = 16 voxel records
  {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float4* zp = reinterpret_cast<const float4*>(zero_page);
    const float* srcb = a.src + (int64_t)b * N * a.sld;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int pc = (i * 4 + wave) * 64 + lane;
      const int v = pc >> 2, q = pc & 3;
      const int x = v % LW, y = (v / LW) % LH, z = v / (LW * LH);
      const int gd = d0 - DIL + z, gh = h0 - DIL + y, gw = w0 - DIL + x;
      const bool ok = v < NV && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
      const float* sp = srcb + (((int64_t)gd * H + gh) * W + gw) * a.sld + q * 4;
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? reinterpret_cast<const float4*>(sp) : zp), (lptr_t)(t16 + (i * 4 + wave) * 64), 16, 0, 0);
    }
  }
  if (a.in_gate) {
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) { bv[tap].x *= gq.x; bv[tap].y *= gq.y; bv[tap].z *= gq.z; bv[tap].w *= gq.w; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 acc[2], acc2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) { acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float relu_floor = (a.flags & N3D_RELU_IN) ? 0.f : -INFINITY;
  const float4* arow = t16 + ((lz * LH + ly0) * LW + m) * 4 + kk;
  mm_bf16x4 bb[BF ? 27 : 1];     // N3D_MM_BF16: the weights rounded once per tile
  if constexpr (BF) {
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) bb[tap] = mm_cvt4(bv[tap]);
  }
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        constexpr int dummy = 0; (void)dummy;
        const int oz = (DG ? 2 - kd : kd) * DIL, oy = (DG ? 2 - kh : kh) * DIL, ox = (DG ? 2 - kw : kw) * DIL;
        const float4 w4 = bv[(kd * 3 + kh) * 3 + kw];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float4 av = arow[((oz * LH + oy + t) * LW + ox) * 4];
          av.x = fmaxf(av.x, relu_floor); av.y = fmaxf(av.y, relu_floor); av.z = fmaxf(av.z, relu_floor); av.w = fmaxf(av.w, relu_floor);
          if constexpr (BF) {
            // (two chains over the taps: even taps into acc, odd taps into acc2)
            if (((kd * 3 + kh) * 3 + kw) & 1) acc2[t] = mm_bf16(mm_cvt4(av), bb[(kd * 3 + kh) * 3 + kw], acc2[t]);
            else acc[t] = mm_bf16(mm_cvt4(av), bb[(kd * 3 + kh) * 3 + kw], acc[t]);
            continue;
          }
          // two accumulator chains per row (x,z / y,w): a dependent 16x16x4 MFMA has 40 cycles latency vs 32 issue
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, w4.x, acc[t], 0, 0, 0);
          acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, w4.y, acc2[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, w4.z, acc[t], 0, 0, 0);
          acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, w4.w, acc2[t], 0, 0, 0);
        }
      }

  // ---- epilogue: lane holds channel m of voxels 4kk .. 4kk+3 of each row
  float csum = 0.f, csq = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = acc[t][r] + acc2[t][r] + e_bias;
      if (!(e_relu[t][r] > 0.f)) v = 0.f;
      v = v * e_gate + e_prev[t][r];
      a.dst[(orow[t] + r) * a.dld + m] = v;
      csum += v; csq = fmaf(v, v, csq);
    }
  if (a.stats) {
    const double s = xsum32_d(xsum16_d((double)csum)), q = xsum32_d(xsum16_d((double)csq));
    if (kk == 0) { red[(wave * 16 + m) * 2] = s; red[(wave * 16 + m) * 2 + 1] = q; }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int q2 = threadIdx.x & 1, col = threadIdx.x >> 1;
      double tot = 0;
      for (int w = 0; w < 4; ++w) tot += red[(w * 16 + col) * 2 + q2];
      a.stats[(((int64_t)b * a.rows_per_sample + tid) * 16 + col) * 2 + q2] = tot;
    }
  }
}

// (BF = N3D_MM_BF16 is a kernel of its own: the fp32 form keeps its register budget -- 164 VGPRs, three waves per SIMD)
template <int DIL, bool DG, bool BF = false>
__global__ __launch_bounds__(256, 2) void conv_tile16_kernel(MfArgs a, const void* zero_page, int tiles, FastDiv fT, FastDiv fTw, FastDiv fTh) {
  extern __shared__ __attribute__((aligned(16))) float4 t16[];   // [NIT * 256] float4: voxel-major, 4 pieces per voxel
  __shared__ double red[4 * 16 * 2];
  tile16_body<DIL, DG, BF>(a, zero_page, tiles, fT, fTw, fTh, t16, red);
}

// ------------------------------------------------------------------------------------------------
// tile32 (round 6): the tile16 scheme for 32 -> 32 channels (3x3x3, stride 1, dilation 1 / 2) on volumes >= 16 wide -- the C = 32
// level of 128^3 patches (2 x 16^3), which the K-split gather plan ran at 0.20 of the fp32 MFMA peak, latency-bound on its 54
// gathered operand loads per wave (profiles/r06_pmc_gemm16_before.json: 58 % of the wave cycles parked in s_waitcnt).  A
// workgroup owns 1 x 4 x 16 output voxels and ONE half of the output channels (grid.y): 256 workgroups at 2 x 16^3, all compute
// units.  LDS: the halo tile as 128-byte voxel records (eight 16-byte channel quads, XOR-swizzled by the voxel index so that the
// 16 voxels of a row land on distinct banks: 41 KB at dilation 1, 102 KB at dilation 2) and the workgroup's weight columns
// Wp[tap][c16][kk][16 cd][j] (54 KB), both by LDS-DMA.  Wave w = row w of the tile: per (tap, 16-channel block) one ds_read_b128
// of the A operand (lane (m, kk): channels 4 kk .. 4 kk + 3 of voxel m), one of the B operand, four v_mfma_f32_16x16x4_f32
// (N3D_MM_BF16: one v_mfma_f32_16x16x16_bf16).  Same arguments, epilogue and packed-weight layout as gemm16; one statistics
// row per tile.   DG: data gradient = the same conv with mirrored taps (weights packed transposed by the pack kernel).
// ------------------------------------------------------------------------------------------------
template <int DIL, bool DG, bool BF>
__global__ __launch_bounds__(256, 1) void conv_tile32_kernel(MfArgs a, const void* zero_page, int tiles, FastDiv fT, FastDiv fTw, FastDiv fTh) {
  N3D_CHAIN_PRIO();
  constexpr int TH = 4, TW = 16;
  constexpr int LD = 1 + 2 * DIL, LH = TH + 2 * DIL, LW = TW + 2 * DIL, NV = LD * LH * LW;
  constexpr int NP = NV * 8, NIT = (NP + 255) / 256;     // float4 pieces of the halo tile (8 per voxel), DMA instructions per thread
  constexpr int WSL = 27 * 2 * 64, WIT = (WSL + 255) / 256;   // float4 slots of this workgroup's weight columns
  extern __shared__ __attribute__((aligned(16))) float4 t32[];   // [NIT * 256] halo pieces, [WIT * 256] weights, 64 float4 of reduction space
  float4* const wl = t32 + NIT * 256;
  double* const red = reinterpret_cast<double*>(wl + WIT * 256);   // [4 * 16 * 2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kk = lane >> 4;
  const int nh = blockIdx.y;                              // which 16 of the 32 output channels
  int wg = blockIdx.x;
  {  // XCD-aware placement (see conv_vox64_kernel): every XCD takes one contiguous run of tiles
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  uint32_t ub, utile, ubx, uw, ud, uh;
  fT.divmod((uint32_t)wg, ub, utile);
  fTw.divmod(utile, ubx, uw);
  fTh.divmod(ubx, ud, uh);
  const int b = (int)ub, tid = (int)utile;
  const int D = a.Dd, H = a.Hd, W = a.Wd;
  const int w0 = (int)uw * TW, h0 = (int)uh * TH, d0 = (int)ud;
  const int64_t N = (int64_t)D * H * W;
  const bool accum = a.flags & N3D_ACCUMULATE;
  const int cd = nh * 16 + m;

  // epilogue operands first (ordinary loads, in flight behind the fill)
  const float e_bias = a.bias ? a.bias[cd] : 0.f;
  const float e_gate = a.out_gate ? a.out_gate[(int64_t)b * 32 + cd] : 1.f;
  float e_relu[4], e_prev[4];
  const int64_t orow = (int64_t)b * N + ((int64_t)d0 * H + h0 + wave) * W + w0 + kk * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    e_relu[r] = a.relu_src ? a.relu_src[(orow + r) * a.rld + cd] : 1.f;
    e_prev[r] = accum ? a.dst[(orow + r) * a.dld + cd] : 0.f;
  }
  {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float4* zp = reinterpret_cast<const float4*>(zero_page);
    const float* srcb = a.src + (int64_t)b * N * a.sld;
    // halo: piece pc -> voxel v = pc >> 3, LDS quad x = pc & 7 holding channel quad q = x ^ ((v >> 1) & 7)
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int pc = (i * 4 + wave) * 64 + lane;
      const int v = pc >> 3, q = (pc & 7) ^ ((v >> 1) & 7);
      const int x = v % LW, y = (v / LW) % LH, z = v / (LW * LH);
      const int gd = d0 - DIL + z, gh = h0 - DIL + y, gw = w0 - DIL + x;
      const bool ok = v < NV && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
      const float* sp = srcb + (((int64_t)gd * H + gh) * W + gw) * a.sld + q * 4;
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? reinterpret_cast<const float4*>(sp) : zp), (lptr_t)(t32 + (i * 4 + wave) * 64), 16, 0, 0);
    }
    // weights: slot s = ((tap * 2 + c16) * 4 + kq) * 16 + mm  <-  Wp float4 ((tap * 2 + c16) * 4 + kq) * 32 + nh * 16 + mm
    const float4* wp4 = reinterpret_cast<const float4*>(a.wp);
#pragma unroll
    for (int i = 0; i < WIT; ++i) {
      const int sidx = (i * 4 + wave) * 64 + lane;
      const float4* gp = sidx < WSL ? wp4 + (sidx >> 4) * 32 + nh * 16 + (sidx & 15) : zp;
      __builtin_amdgcn_global_load_lds((gptr_t)gp, (lptr_t)(wl + (i * 4 + wave) * 64), 16, 0, 0);
    }
  }
  float4 gq[2];
  gq[0] = gq[1] = make_float4(1.f, 1.f, 1.f, 1.f);
  if (a.in_gate) {
    gq[0] = *reinterpret_cast<const float4*>(a.in_gate + (int64_t)b * 32 + kk * 4);
    gq[1] = *reinterpret_cast<const float4*>(a.in_gate + (int64_t)b * 32 + 16 + kk * 4);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  const float relu_floor = (a.flags & N3D_RELU_IN) ? 0.f : -INFINITY;
  const bool gated = a.in_gate != nullptr;
  const int vrow = (DIL * LH + wave + DIL) * LW + DIL + m;      // halo index of this lane's output voxel (tap offset 0,0,0 = centre - DIL)
  const float4* wlane = wl + kk * 16 + m;
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int oz = (DG ? 1 - kd : kd - 1) * DIL, oy = (DG ? 1 - kh : kh - 1) * DIL, ox = (DG ? 1 - kw : kw - 1) * DIL;
        const int tap = (kd * 3 + kh) * 3 + kw;
        const int v = vrow + (oz * LH + oy) * LW + ox;
        const int sw = (v >> 1) & 7;
#pragma unroll
        for (int c16 = 0; c16 < 2; ++c16) {
          float4 av = t32[v * 8 + ((c16 * 4 + kk) ^ sw)];
          const float4 w4 = wlane[(tap * 2 + c16) * 64];
          av.x = fmaxf(av.x, relu_floor); av.y = fmaxf(av.y, relu_floor); av.z = fmaxf(av.z, relu_floor); av.w = fmaxf(av.w, relu_floor);
          if (gated) { av.x *= gq[c16].x; av.y *= gq[c16].y; av.z *= gq[c16].z; av.w *= gq[c16].w; }
          if constexpr (BF) {
            if (c16) acc2 = mm_bf16(mm_cvt4(av), mm_cvt4(w4), acc2);
            else acc = mm_bf16(mm_cvt4(av), mm_cvt4(w4), acc);
          } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, w4.x, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, w4.y, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, w4.z, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, w4.w, acc2, 0, 0, 0);
          }
        }
      }

  // ---- epilogue: lane holds channel cd of voxels 4kk .. 4kk+3 of its row
  float csum = 0.f, csq = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v = acc[r] + acc2[r] + e_bias;
    if (!(e_relu[r] > 0.f)) v = 0.f;
    v = v * e_gate + e_prev[r];
    a.dst[(orow + r) * a.dld + cd] = v;
    csum += v; csq = fmaf(v, v, csq);
  }
  if (a.stats) {
    const double s = xsum32_d(xsum16_d((double)csum)), q = xsum32_d(xsum16_d((double)csq));
    if (kk == 0) { red[(wave * 16 + m) * 2] = s; red[(wave * 16 + m) * 2 + 1] = q; }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int q2 = threadIdx.x & 1, col = threadIdx.x >> 1;
      double tot = 0;
      for (int w = 0; w < 4; ++w) tot += red[(w * 16 + col) * 2 + q2];
      a.stats[(((int64_t)b * a.rows_per_sample + tid) * 32 + nh * 16 + col) * 2 + q2] = tot;
    }
  }
}

// geometry-only decision (n3d_conv_stats_rows must agree with the launch): tiles per sample, 0 = not this kernel
static int tile32_tiles(const n3d_conv_geom* g) {
  if (g->depthwise || g->k != 3 || g->stride != 1 || g->Ci != 32 || g->Co != 32) return 0;
  if ((g->dil != 1 && g->dil != 2) || g->pad != g->dil) return 0;
  if (g->Wi % 16 != 0 || g->Hi % 4 != 0) return 0;
  const int tiles = (g->Wi / 16) * (g->Hi / 4) * g->Di;
  if ((int64_t)tiles * g->B < 128) return 0;      // fewer tiles than half the compute units: the K-split plans serve the small levels
  return tiles;
}

static bool launch_tile32(const MfArgs& a, int tiles, hipStream_t s) {
  const void* zp = zero_page_ptr();
  if (!zp || a.sld % 4 != 0 || !aligned16(a.src) || !aligned16(a.wp)) return false;
  const int d = a.dt < 0 ? -a.dt : a.dt;
  const int nv = (1 + 2 * d) * (4 + 2 * d) * (16 + 2 * d);
  const size_t shm = ((size_t)((nv * 8 + 255) / 256) * 256 + (size_t)((27 * 2 * 64 + 255) / 256) * 256 + 64) * 16;
  const dim3 grid((unsigned)(tiles * a.B), 2);
  const FastDiv fT((uint32_t)tiles), fTw((uint32_t)(a.Wd / 16)), fTh((uint32_t)(a.Hd / 4));
  const bool bf = a.flags & N3D_MM_BF16;
  static bool attr_set = false;
  if (!attr_set) {
#define N3D_T32_ATTR(DIL_, DG_, BF_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tile32_kernel<DIL_, DG_, BF_>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    N3D_T32_ATTR(1, false, false); N3D_T32_ATTR(1, true, false); N3D_T32_ATTR(2, false, false); N3D_T32_ATTR(2, true, false);
    N3D_T32_ATTR(1, false, true); N3D_T32_ATTR(1, true, true); N3D_T32_ATTR(2, false, true); N3D_T32_ATTR(2, true, true);
#undef N3D_T32_ATTR
    attr_set = true;
  }
#define N3D_T32_LAUNCH(DIL_, DG_) do { if (bf) hipLaunchKernelGGL((conv_tile32_kernel<DIL_, DG_, true>), grid, dim3(256), shm, s, a, zp, tiles, fT, fTw, fTh); \
    else hipLaunchKernelGGL((conv_tile32_kernel<DIL_, DG_, false>), grid, dim3(256), shm, s, a, zp, tiles, fT, fTw, fTh); } while (0)
  if (d == 1) { if (a.dt > 0) N3D_T32_LAUNCH(1, false); else N3D_T32_LAUNCH(1, true); }
  else { if (a.dt > 0) N3D_T32_LAUNCH(2, false); else N3D_T32_LAUNCH(2, true); }
#undef N3D_T32_LAUNCH
  return true;
}

// ------------------------------------------------------------------------------------------------
// tile16_up: the den = 2 gather of the 16-channel level -- forward of a stride-2 transposed 3x3x3 conv and data gradient of a
// stride-2 conv (destination grid = 2 x source grid exactly):   dst[o] = sum_k W[k] . src[(o + pad - k*dil) / 2]   (if divisible).
// Through the gemm16 gather map every output voxel visits all 27 taps and masks the 23.6 that do not divide (27 us at 2 x 32^3
// outputs).  Here a workgroup owns 1 x 4 x 16 SOURCE voxels and produces all 8 output parity classes from one LDS halo tile:
// per dimension, o = 2s + p takes  dil 1: p = 0 -> (k 1, s), p = 1 -> (k 0, s + 1), (k 2, s);  dil 2: p = 0 -> (k 0, s + 1), (k 1, s),
// (k 2, s - 1), p = 1 -> nothing -- so the 27 taps are used exactly once over the 8 classes, each as ONE ds_read_b128 per 16-voxel
// source row feeding four v_mfma_f32_16x16x4_f32 (the A-operand scheme of conv_tile16).  Same arguments and epilogue as gemm16;
// one statistics row per workgroup (512 outputs).
// ------------------------------------------------------------------------------------------------
template <int DIL, bool BF>
__device__ __forceinline__ void tile16_up_body(const MfArgs& a, const void* zero_page, int tiles, const FastDiv& fT, const FastDiv& fTw, const FastDiv& fTh,
                                               float4* const t16, double* const red) {
  N3D_CHAIN_PRIO();
  constexpr int TH = 4, TW = 16;
  constexpr int LO = DIL == 2 ? 1 : 0;                   // source halo below (above: always 1)
  constexpr int LD = 1 + LO + 1, LH = TH + LO + 1, LW = TW + LO + 1, NV = LD * LH * LW;
  constexpr int NP = NV * 4, NIT = (NP + 255) / 256;
  constexpr int NCLS = DIL == 1 ? 8 : 1;                 // output parity classes that receive taps
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kk = lane >> 4;
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  uint32_t ub, utile, ubx, uw, ud, uh;
  fT.divmod((uint32_t)wg, ub, utile);
  fTw.divmod(utile, ubx, uw);
  fTh.divmod(ubx, ud, uh);
  const int b = (int)ub, tid = (int)utile;
  const int Ds = a.Ds, Hs = a.Hs, Ws = a.Ws, Dd = a.Dd, Hd = a.Hd, Wd = a.Wd;
  const int w0 = (int)uw * TW, h0 = (int)uh * TH, d0 = (int)ud;
  const int ly = wave;                                    // this wave: source row ly of the tile
  const bool accum = a.flags & N3D_ACCUMULATE;
  const float e_bias = a.bias ? a.bias[m] : 0.f;
  const float e_gate = a.out_gate ? a.out_gate[(int64_t)b * 16 + m] : 1.f;
  float4 bv[27];
  {
    const float4* __restrict__ wp4 = reinterpret_cast<const float4*>(a.wp);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) bv[tap] = wp4[(tap * 4 + kk) * 16 + m];
  }
  float4 gq = make_float4(1.f, 1.f, 1.f, 1.f);
  if (a.in_gate) gq = *reinterpret_cast<const float4*>(a.in_gate + (int64_t)b * 16 + kk * 4);
  {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float4* zp = reinterpret_cast<const float4*>(zero_page);
    const float* srcb = a.src + (int64_t)b * Ds * Hs * Ws * a.sld;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int pc = (i * 4 + wave) * 64 + lane;
      const int v = pc >> 2, q = pc & 3;
      const int x = v % LW, y = (v / LW) % LH, z = v / (LW * LH);
      const int gd = d0 - LO + z, gh = h0 - LO + y, gw = w0 - LO + x;
      const bool ok = v < NV && (unsigned)gd < (unsigned)Ds && (unsigned)gh < (unsigned)Hs && (unsigned)gw < (unsigned)Ws;
      const float* sp = srcb + (((int64_t)gd * Hs + gh) * Ws + gw) * a.sld + q * 4;
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? reinterpret_cast<const float4*>(sp) : zp), (lptr_t)(t16 + (i * 4 + wave) * 64), 16, 0, 0);
    }
  }
  if (a.in_gate) {
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) { bv[tap].x *= gq.x; bv[tap].y *= gq.y; bv[tap].z *= gq.z; bv[tap].w *= gq.w; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 acc[NCLS], acc2[NCLS];
#pragma unroll
  for (int c = 0; c < NCLS; ++c) { acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc2[c] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float relu_floor = (a.flags & N3D_RELU_IN) ? 0.f : -INFINITY;
  const float4* arow = t16 + ((LO * LH + ly + LO) * LW + LO + m) * 4 + kk;   // source voxel (d0, h0 + ly, w0 + m)
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        // per dimension: output parity and source shift of tap k
        const int pd = DIL == 1 ? (kd != 1) : 0, ph = DIL == 1 ? (kh != 1) : 0, pw = DIL == 1 ? (kw != 1) : 0;
        const int sd = DIL == 1 ? (kd == 0) : 1 - kd, sh = DIL == 1 ? (kh == 0) : 1 - kh, sw = DIL == 1 ? (kw == 0) : 1 - kw;
        const int cls = DIL == 1 ? pd * 4 + ph * 2 + pw : 0;
        const float4 w4 = bv[(kd * 3 + kh) * 3 + kw];
        float4 av = arow[((sd * LH + sh) * LW + sw) * 4];
        av.x = fmaxf(av.x, relu_floor); av.y = fmaxf(av.y, relu_floor); av.z = fmaxf(av.z, relu_floor); av.w = fmaxf(av.w, relu_floor);
        if constexpr (BF) {      // N3D_MM_BF16: one 16-deep bf16 MFMA per tap (operands rounded in registers)
          acc[cls] = mm_bf16(mm_cvt4(av), mm_cvt4(w4), acc[cls]);
          continue;
        }
        acc[cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, w4.x, acc[cls], 0, 0, 0);
        acc2[cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, w4.y, acc2[cls], 0, 0, 0);
        acc[cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, w4.z, acc[cls], 0, 0, 0);
        acc2[cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, w4.w, acc2[cls], 0, 0, 0);
      }

  // ---- epilogue: lane holds channel m of source voxels 4kk .. 4kk+3 of its row, for every output parity class
  float csum = 0.f, csq = 0.f;
  const int64_t Nd = (int64_t)Dd * Hd * Wd;
#pragma unroll
  for (int cls = 0; cls < 8; ++cls) {
    const int pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
    const int64_t orow = (int64_t)b * Nd + ((int64_t)(2 * d0 + pd) * Hd + 2 * (h0 + ly) + ph) * Wd + 2 * (w0 + kk * 4) + pw;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t o = orow + 2 * r;
      float v = e_bias;
      if (cls < NCLS) v += acc[cls < NCLS ? cls : 0][r] + acc2[cls < NCLS ? cls : 0][r];
      if (a.relu_src && !(a.relu_src[o * a.rld + m] > 0.f)) v = 0.f;
      v *= e_gate;
      if (accum) v += a.dst[o * a.dld + m];
      a.dst[o * a.dld + m] = v;
      csum += v; csq = fmaf(v, v, csq);
    }
  }
  if (a.stats) {
    const double s = xsum32_d(xsum16_d((double)csum)), q = xsum32_d(xsum16_d((double)csq));
    if (kk == 0) { red[(wave * 16 + m) * 2] = s; red[(wave * 16 + m) * 2 + 1] = q; }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int q2 = threadIdx.x & 1, col = threadIdx.x >> 1;
      double tot = 0;
      for (int w = 0; w < 4; ++w) tot += red[(w * 16 + col) * 2 + q2];
      a.stats[(((int64_t)b * tiles + tid) * 16 + col) * 2 + q2] = tot;
    }
  }
}

template <int DIL, bool BF = false>
__global__ __launch_bounds__(256, 2) void conv_tile16_up_kernel(MfArgs a, const void* zero_page, int tiles, FastDiv fT, FastDiv fTw, FastDiv fTh) {
  extern __shared__ __attribute__((aligned(16))) float4 t16[];
  __shared__ double red[4 * 16 * 2];
  tile16_up_body<DIL, BF>(a, zero_page, tiles, fT, fTw, fTh, t16, red);
}

// geometry-only decision (n3d_conv_stats_rows must agree with the launch): tiles per sample, 0 = not this kernel
static int tile16_up_tiles(const n3d_conv_geom* g, bool data_grad) {
  constexpr bool off = false;
  if (off || !data_grad || g->depthwise || g->k != 3 || g->stride != 2 || g->Ci != 16 || g->Co != 16) return 0;
  if ((g->dil != 1 && g->dil != 2) || g->pad != g->dil) return 0;
  if (g->Di != 2 * g->Do || g->Hi != 2 * g->Ho || g->Wi != 2 * g->Wo || g->Wo % 16 != 0 || g->Ho % 4 != 0) return 0;
  if ((int64_t)g->B * g->Di * g->Hi * g->Wi < 16384 * 2) return 0;    // the K-split plans serve the small levels
  return (g->Wo / 16) * (g->Ho / 4) * g->Do;
}

static bool tile16_applies(const MfArgs& a, int ksplit) {
  constexpr bool off = false;
  if (off || ksplit != 1 || a.k != 3 || a.sn != 1 || a.den != 1 || a.Cs != 16 || a.Cd != 16) return false;
  const int d = a.dt < 0 ? -a.dt : a.dt;
  if ((d != 1 && d != 2) || a.off != -a.dt) return false;
  if (a.Ds != a.Dd || a.Hs != a.Hd || a.Ws != a.Wd || a.Wd % 16 != 0 || a.Hd % 4 != 0 || a.Dd % 2 != 0) return false;
  if (a.stats && a.rows_per_sample != (a.Wd / 16) * (a.Hd / 4) * (a.Dd / 2)) return false;
  return a.sld % 4 == 0 && aligned16(a.src) && aligned16(a.wp);
}

static bool launch_tile16(const MfArgs& a, hipStream_t s) {
  const void* zp = zero_page_ptr();
  if (!zp) return false;
  const int d = a.dt < 0 ? -a.dt : a.dt;
  const int tiles = (a.Wd / 16) * (a.Hd / 4) * (a.Dd / 2);
  const int nv = (2 + 2 * d) * (4 + 2 * d) * (16 + 2 * d);
  const size_t shm = (size_t)((nv * 4 + 255) / 256) * 256 * 16;
  const dim3 grid((unsigned)(tiles * a.B));
  const FastDiv fT((uint32_t)tiles), fTw((uint32_t)(a.Wd / 16)), fTh((uint32_t)(a.Hd / 4));
  const bool bf = a.flags & N3D_MM_BF16;
#define N3D_T16_LAUNCH(DIL_, DG_) do { if (bf) hipLaunchKernelGGL((conv_tile16_kernel<DIL_, DG_, true>), grid, dim3(256), shm, s, a, zp, tiles, fT, fTw, fTh); \
    else hipLaunchKernelGGL((conv_tile16_kernel<DIL_, DG_, false>), grid, dim3(256), shm, s, a, zp, tiles, fT, fTw, fTh); } while (0)
  if (d == 1) {
    if (a.dt > 0) N3D_T16_LAUNCH(1, false); else N3D_T16_LAUNCH(1, true);
  } else {
    if (a.dt > 0) N3D_T16_LAUNCH(2, false); else N3D_T16_LAUNCH(2, true);
  }
#undef N3D_T16_LAUNCH
  return true;
}

struct G16Plan { int mt, nt, ksplit, rows_per_block; bool ok; };
static int tile32_tiles(const n3d_conv_geom* g);

static G16Plan g16_plan(const n3d_conv_geom* g, bool data_grad) {
  G16Plan p; p.ok = false; p.mt = 1; p.nt = 1; p.ksplit = 4; p.rows_per_block = 16;
  if (g->depthwise) return p;
  const int Cs = data_grad ? g->Co : g->Ci, Cd = data_grad ? g->Ci : g->Co;
  if (Cs % 16 != 0 || Cd % 16 != 0) return p;
  // the kernel's per-workgroup tap table holds 256 K-groups (taps x Cs / 16), its packed tap offsets 8 bits, its byte offsets 31
  if ((int64_t)g->k * g->k * g->k * (Cs / 16) > 256 || g->pad > 4 || g->dil > 2) return p;
  {
    const int64_t Ns = data_grad ? (int64_t)g->Do * g->Ho * g->Wo : (int64_t)g->Di * g->Hi * g->Wi;
    if ((int64_t)g->B * Ns * Cs * 4 >= (1ll << 30) || (int64_t)g->k * g->k * g->k * Cs * Cd * 4 >= (1ll << 30)) return p;
  }
  const int64_t Nd = data_grad ? (int64_t)g->Di * g->Hi * g->Wi : (int64_t)g->Do * g->Ho * g->Wo;
  const int64_t M = (int64_t)g->B * Nd;
  const int64_t tiles = cdiv(M, 16) * (Cd / 16);
  if (tiles <= 256) { p.mt = 1; p.nt = 1; p.ksplit = 16; p.rows_per_block = 16; }
  else if (tiles <= 1024) { p.mt = 1; p.nt = 1; p.ksplit = 4; p.rows_per_block = 16; }
  else { p.mt = 2; p.nt = (Cd % 32 == 0) ? 2 : 1; p.ksplit = 1; p.rows_per_block = 128; }
  p.ok = true;
  return p;
}

int mfma_pack_layout(const n3d_conv_geom* g, bool data_grad, int flags) {
  if (flags & (N3D_NO_MFMA | N3D_SRC_BF16 | N3D_DST_BF16)) return 0;
  if (vup_plan(g, data_grad).ok) return 3;   // vox layout, channels transposed, taps not flipped
  if (vx_plan(g).ok || vs2_plan(g, data_grad).ok) return 2;
  if (g16_plan(g, data_grad).ok) return 1;
  return 0;
}

int mfma_conv_stats_rows(const n3d_conv_geom* g, bool data_grad, int flags) {
  if (flags & (N3D_NO_MFMA | N3D_SRC_BF16 | N3D_DST_BF16)) return 0;   // bf16 storage: the streaming / gather kernels (conv_generic.hip)
  {
    VxPlan v = vx_plan(g);
    if (v.ok) return v.tiles * v.nw;  // one partial row per wave
    Vs2Plan v2 = vs2_plan(g, data_grad);
    if (v2.ok) return v2.tiles;
    VupPlan v3 = vup_plan(g, data_grad);
    if (v3.ok) return v3.tiles;
  }
  { const int t = tile32_tiles(g); if (t) return t; }     // the LDS-tile kernel of the 32-channel level: one row per 1 x 4 x 16 tile
  G16Plan p = g16_plan(g, data_grad);
  if (!p.ok) return 0;
  if (p.ksplit == 1) { const int t = tile16_up_tiles(g, data_grad); if (t) return t; }
  const int64_t Nd = data_grad ? (int64_t)g->Di * g->Hi * g->Wi : (int64_t)g->Do * g->Ho * g->Wo;
  if (Nd * 2 == p.rows_per_block && p.ksplit > 1) return 1;   // the 2^3 level: a 16-row tile holds two samples, one row each
  if (Nd % p.rows_per_block != 0) return -1;  // statistics not produced by this kernel: caller must use n3d_channel_stats
  return (int)(Nd / p.rows_per_block);
}

template <int MT, int NT, int KS>
static void launch_g16(const MfArgs& a, int64_t M, hipStream_t s) {
  constexpr int RPB = 16 * MT * (KS == 1 ? 4 : 1);
  dim3 grid((unsigned)cdiv(M, RPB), (unsigned)(a.Cd / (16 * NT)));
  size_t shm = (KS > 1 ? (size_t)(KS - 1) * MT * NT * 256 * sizeof(float) : 0) + (size_t)16 * NT * 16 * 2 * sizeof(double);
  hipLaunchKernelGGL((conv_gemm16_kernel<MT, NT, KS>), grid, dim3(KS == 16 ? 1024 : 256), shm, s, a);
}

void mfma_pack16(const float* w, float* wp, int Co, int Ci, int taps, int data_grad, hipStream_t s) {
  hipLaunchKernelGGL(pack16_kernel, dim3((unsigned)cdiv((int64_t)taps * Co * Ci, 256)), dim3(256), 0, s, w, wp, Co, Ci, taps, data_grad);
}

int g16_prepare(const n3d_conv_geom* g, bool data_grad, const float* src, int64_t sld, const float* w, const float* bias, float* dst,
                int64_t dld, int flags, const float* in_gate, const float* relu_src, int64_t rld, const float* out_gate, double* stats,
                void* ws, size_t ws_bytes, hipStream_t s, MfArgs* out, G16Plan* plan);

int mfma_conv_try(const n3d_conv_geom* g, bool data_grad, const float* src, int64_t sld, const float* w, const float* bias, float* dst,
                  int64_t dld, int flags, const float* in_gate, const float* relu_src, int64_t rld, const float* out_gate, double* stats,
                  void* ws, size_t ws_bytes, hipStream_t s) {
  if (flags & (N3D_SRC_BF16 | N3D_DST_BF16)) return 0;
  {
    Vs2Plan v2 = vs2_plan(g, data_grad);
    if (v2.ok) {
      if (in_gate || relu_src || out_gate || (flags & N3D_RELU_IN) || sld % 4 != 0 || dld % 4 != 0 || !aligned16(src) || !aligned16(dst)) {
        if (stats || (flags & N3D_PREPACKED)) { set_error("conv(vox_s2): gate / relu extras are not supported on this shape with statistics or pre-packed weights"); return N3D_ERR_UNSUPPORTED; }
        return 0;
      }
      const size_t need = (size_t)27 * v2.C * v2.C * 4;
      if (!ws || ws_bytes < need) { set_error("conv(vox_s2): workspace too small"); return N3D_ERR_WORKSPACE; }
      float* wq = (float*)ws;
      if (!(flags & N3D_PREPACKED))
        hipLaunchKernelGGL(pack_vox_kernel, dim3((unsigned)cdiv(27 * v2.C * v2.C, 256)), dim3(256), 0, s, w, wq, v2.C, 0);
      Vs2Args a;
      a.src = src; a.sld = sld; a.D = g->Di; a.H = g->Hi; a.W = g->Wi; a.dst = dst; a.dld = dld; a.oD = g->Do; a.oH = g->Ho; a.oW = g->Wo;
      a.wq = wq; a.bias = bias; a.flags = flags; a.stats = stats; a.rows_per_sample = v2.tiles; a.tiles = v2.tiles; a.zero_page = zero_page_ptr();
      a.fT = FastDiv((uint32_t)v2.tiles); a.fTw = FastDiv((uint32_t)(g->Wo / 16)); a.fTh = FastDiv((uint32_t)(g->Ho / 4));
      if (!a.zero_page) { set_error("conv(vox_s2): zero page symbol unavailable"); return N3D_ERR_HIP; }
      launch_vs2(a, v2, g->B, s);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { set_error("conv(vox_s2) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
      return 1;
    }
    VupPlan v3 = vup_plan(g, data_grad);
    if (v3.ok) {
      if (in_gate || relu_src || out_gate || (flags & N3D_RELU_IN) || sld % 4 != 0 || dld % 4 != 0 || !aligned16(src) || !aligned16(dst)) {
        if (stats || (flags & N3D_PREPACKED)) { set_error("conv(vox_up): gate / relu extras are not supported on this shape with statistics or pre-packed weights"); return N3D_ERR_UNSUPPORTED; }
        return 0;
      }
      const size_t need = (size_t)27 * v3.C * v3.C * 4;
      if (!ws || ws_bytes < need) { set_error("conv(vox_up): workspace too small"); return N3D_ERR_WORKSPACE; }
      float* wq = (float*)ws;
      if (!(flags & N3D_PREPACKED))
        hipLaunchKernelGGL(pack_vox_kernel, dim3((unsigned)cdiv(27 * v3.C * v3.C, 256)), dim3(256), 0, s, w, wq, v3.C, 2);
      VupArgs a;
      a.src = src; a.sld = sld; a.D = g->Do; a.H = g->Ho; a.W = g->Wo; a.dst = dst; a.dld = dld;
      a.wq = wq; a.bias = bias; a.flags = flags; a.stats = stats; a.rows_per_sample = v3.tiles; a.tiles = v3.tiles; a.zero_page = zero_page_ptr();
      a.fT = FastDiv((uint32_t)v3.tiles); a.fTw = FastDiv((uint32_t)(g->Wo / 16)); a.fTh = FastDiv((uint32_t)(g->Ho / 4));
      if (!a.zero_page) { set_error("conv(vox_up): zero page symbol unavailable"); return N3D_ERR_HIP; }
      if (v3.C == 4) { if (v3.dil == 1) hipLaunchKernelGGL((conv_vox_up_kernel<4, 1>), dim3(v3.tiles * g->B), dim3(64), v3.lds, s, a);
                       else hipLaunchKernelGGL((conv_vox_up_kernel<4, 2>), dim3(v3.tiles * g->B), dim3(64), v3.lds, s, a); }
      else { if (v3.dil == 1) hipLaunchKernelGGL((conv_vox_up_kernel<8, 1>), dim3(v3.tiles * g->B), dim3(64), v3.lds, s, a);
             else hipLaunchKernelGGL((conv_vox_up_kernel<8, 2>), dim3(v3.tiles * g->B), dim3(64), v3.lds, s, a); }
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { set_error("conv(vox_up) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
      return 1;
    }
    VxPlan v = vx_plan(g);
    if (v.ok) {
      if (in_gate || relu_src || out_gate || (flags & N3D_RELU_IN) || sld % 4 != 0 || dld % 4 != 0 || !aligned16(src) || !aligned16(dst)) {
        if (stats || (flags & N3D_PREPACKED)) { set_error("conv(vox64): gate / relu extras are not supported on this shape with statistics or pre-packed weights"); return N3D_ERR_UNSUPPORTED; }
        return 0;
      }
      const size_t need = (size_t)27 * v.C * v.C * 4;
      if (!ws || ws_bytes < need) { set_error("conv(vox64): workspace too small"); return N3D_ERR_WORKSPACE; }
      float* wq = (float*)ws;
      if (!(flags & N3D_PREPACKED))
        hipLaunchKernelGGL(pack_vox_kernel, dim3((unsigned)cdiv(27 * v.C * v.C, 256)), dim3(256), 0, s, w, wq, v.C, data_grad ? 1 : 0);
      VxArgs a;
      a.src = src; a.sld = sld; a.dst = dst; a.dld = dld; a.wq = wq; a.bias = bias; a.D = g->Di; a.H = g->Hi; a.W = g->Wi; a.flags = flags;
      a.stats = stats; a.rows_per_sample = v.tiles * v.nw;
      const int launched = v.C == 4 ? launch_vox_c<4>(a, v, g->B, s) : launch_vox_c<8>(a, v, g->B, s);
      if (!launched) { set_error("conv(vox64): zero page symbol unavailable"); return N3D_ERR_HIP; }
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { set_error("conv(vox64) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
      return 1;
    }
  }
  MfArgs a;
  G16Plan p;
  {
    const int r = g16_prepare(g, data_grad, src, sld, w, bias, dst, dld, flags, in_gate, relu_src, rld, out_gate, stats, ws, ws_bytes, s, &a, &p);
    if (r <= 0) return r;
  }
  const int64_t M = (int64_t)g->B * a.Dd * a.Hd * a.Wd;
  if (p.ksplit == 1) {
    const int tiles = tile16_up_tiles(g, data_grad);
    if (tiles) {
      const void* zp = zero_page_ptr();
      if (a.sld % 4 != 0 || !aligned16(a.src) || !aligned16(a.wp) || !zp) {
        if (stats) { set_error("conv(tile16_up): misaligned source with statistics requested"); return N3D_ERR_UNSUPPORTED; }
      } else {
        a.rows_per_sample = tiles;
        const int d = g->dil, lo = d == 2 ? 1 : 0;
        const int nv = (2 + lo) * (4 + lo + 1) * (16 + lo + 1);
        const size_t shm = (size_t)((nv * 4 + 255) / 256) * 256 * 16;
        const dim3 grid((unsigned)(tiles * g->B));
        const FastDiv fT((uint32_t)tiles), fTw((uint32_t)(g->Wo / 16)), fTh((uint32_t)(g->Ho / 4));      // (source grid = the o side)
        const bool bf = a.flags & N3D_MM_BF16;
        if (d == 1) { if (bf) hipLaunchKernelGGL((conv_tile16_up_kernel<1, true>), grid, dim3(256), shm, s, a, zp, tiles, fT, fTw, fTh);
                      else hipLaunchKernelGGL((conv_tile16_up_kernel<1, false>), grid, dim3(256), shm, s, a, zp, tiles, fT, fTw, fTh); }
        else { if (bf) hipLaunchKernelGGL((conv_tile16_up_kernel<2, true>), grid, dim3(256), shm, s, a, zp, tiles, fT, fTw, fTh);
               else hipLaunchKernelGGL((conv_tile16_up_kernel<2, false>), grid, dim3(256), shm, s, a, zp, tiles, fT, fTw, fTh); }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("conv(tile16_up) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
        return 1;
      }
    }
  }
  if (const int t32 = tile32_tiles(g)) {
    if (a.sn == 1 && a.den == 1 && a.Cs == 32 && a.Cd == 32) {
      if (stats) a.rows_per_sample = t32;
      if (launch_tile32(a, t32, s)) {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("conv(tile32) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
        return 1;
      }
      if (stats) { set_error("conv(tile32): misaligned source with statistics requested"); return N3D_ERR_UNSUPPORTED; }
    }
  }
  if (tile16_applies(a, p.ksplit) && launch_tile16(a, s)) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("conv(tile16) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
    return 1;
  }
  if (p.ksplit == 16) launch_g16<1, 1, 16>(a, M, s);
  else if (p.ksplit == 4) launch_g16<1, 1, 4>(a, M, s);
  else if (p.nt == 2) launch_g16<2, 2, 1>(a, M, s);
  else launch_g16<2, 1, 1>(a, M, s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("conv(mfma) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
  return 1;
}

// fills the arguments of the gemm16 family for one conv (packing the weights unless pre-packed); 1 = ready, 0 = shape not
// served by gemm16, < 0 error
int g16_prepare(const n3d_conv_geom* g, bool data_grad, const float* src, int64_t sld, const float* w, const float* bias, float* dst,
                int64_t dld, int flags, const float* in_gate, const float* relu_src, int64_t rld, const float* out_gate, double* stats,
                void* ws, size_t ws_bytes, hipStream_t s, MfArgs* out, G16Plan* plan) {
  if (vx_plan(g).ok) return 0;
  G16Plan p = g16_plan(g, data_grad);
  if (!p.ok) return 0;
  if (sld % 4 != 0 || !aligned16(src)) {
    if (flags & N3D_PREPACKED) { set_error("conv(gemm16): misaligned source with pre-packed weights"); return N3D_ERR_UNSUPPORTED; }
    return 0;
  }
  const int taps = g->k * g->k * g->k;
  MfArgs a;
  a.src = src; a.sld = sld; a.dst = dst; a.dld = dld; a.bias = bias; a.k = g->k; a.flags = flags; a.B = g->B;
  a.in_gate = in_gate; a.relu_src = relu_src; a.rld = rld; a.out_gate = out_gate; a.stats = stats;
  if (!data_grad) { a.Ds = g->Di; a.Hs = g->Hi; a.Ws = g->Wi; a.Cs = g->Ci; a.Dd = g->Do; a.Hd = g->Ho; a.Wd = g->Wo; a.Cd = g->Co;
    a.sn = g->stride; a.off = -g->pad; a.dt = g->dil; a.den = 1; }
  else { a.Ds = g->Do; a.Hs = g->Ho; a.Ws = g->Wo; a.Cs = g->Co; a.Dd = g->Di; a.Hd = g->Hi; a.Wd = g->Wi; a.Cd = g->Ci;
    a.sn = 1; a.off = g->pad; a.dt = -g->dil; a.den = g->stride; }
  const int64_t Nd = (int64_t)a.Dd * a.Hd * a.Wd;
  if ((int64_t)g->B * Nd >= (1ll << 31)) return 0;  // 32-bit voxel indexing
  {
    // 31-bit byte offsets into the source (its voxel pitch may be that of a wider buffer)
    const int64_t Ns = data_grad ? (int64_t)g->Do * g->Ho * g->Wo : (int64_t)g->Di * g->Hi * g->Wi;
    if (((int64_t)g->B * Ns + 8 * ((int64_t)(data_grad ? g->Ho : g->Hi) + 1) * ((data_grad ? g->Wo : g->Wi) + 1)) * sld * 4 >= (1ll << 30)) return 0;
  }
  a.fNd = FastDiv((uint32_t)Nd); a.fWd = FastDiv((uint32_t)a.Wd); a.fHd = FastDiv((uint32_t)a.Hd); a.fC16 = FastDiv((uint32_t)(a.Cs / 16));
  if (stats) {
    if (Nd * 2 == p.rows_per_block && p.ksplit > 1) a.rows_per_sample = 1;
    else {
      if (Nd % p.rows_per_block != 0) { set_error("conv(mfma): statistics requested for a shape whose n3d_conv_stats_rows() is -1"); return N3D_ERR_INVALID; }
      a.rows_per_sample = (int)(Nd / p.rows_per_block);
    }
  } else a.rows_per_sample = 0;
  const size_t need = (size_t)taps * a.Cs * a.Cd * 4;
  if (!ws || ws_bytes < need) { set_error("conv(mfma): workspace too small (%zu < %zu)", ws_bytes, need); return N3D_ERR_WORKSPACE; }
  float* wp = (float*)ws;
  a.wp = wp;
  const int total = taps * a.Cs * a.Cd;
  if (!(flags & N3D_PREPACKED))
    hipLaunchKernelGGL(pack16_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wp, g->Co, g->Ci, taps, data_grad ? 1 : 0);
  *out = a; *plan = p;
  return 1;
}

// two forward-type convs (forward, or transposed forward) of the K-split gemm16 family in one launch; 1 = launched,
// 0 = not applicable (nothing launched, nothing packed), < 0 error
int mfma_conv_pair_try(const n3d_conv_geom* g0, bool dg0, const float* src0, int64_t sld0, const float* w0, const float* bias0, float* dst0,
                       int64_t dld0, int flags0, const float* gate0, double* stats0, void* ws0, size_t wsb0, const n3d_conv_geom* g1, bool dg1,
                       const float* src1, int64_t sld1, const float* w1, const float* bias1, float* dst1, int64_t dld1, int flags1,
                       const float* gate1, double* stats1, void* ws1, size_t wsb1, hipStream_t s, const PairExtras* x0, const PairExtras* x1) {
  // decide before touching anything (g16_prepare packs weights)
  if ((flags0 | flags1) & (N3D_SRC_BF16 | N3D_DST_BF16)) return 0;
  if (vx_plan(g0).ok || vx_plan(g1).ok) return 0;
  if (tile32_tiles(g0) || tile32_tiles(g1)) return 0;      // the LDS-tile kernel takes these one at a time (two launches beat the K-split pair)
  const G16Plan p0 = g16_plan(g0, dg0), p1 = g16_plan(g1, dg1);
  if (!p0.ok || !p1.ok || p0.ksplit != p1.ksplit || p0.ksplit == 1) return 0;
  if (sld0 % 4 != 0 || !aligned16(src0) || sld1 % 4 != 0 || !aligned16(src1)) return 0;
  if (ws0 == ws1 && !((flags0 & flags1) & N3D_PREPACKED)) return 0;   // one kernel reads both packed copies: they must not share a workspace
  PairArgs q;
  G16Plan t0, t1;
  // x0 / x1: the data-gradient extras of a conv (ReLU mask source of the input, per-(b,c) output gate); NULL for a forward conv
  int r = g16_prepare(g0, dg0, src0, sld0, w0, bias0, dst0, dld0, flags0, gate0, x0 ? x0->relu_src : nullptr, x0 ? x0->rld : 0,
                      x0 ? x0->out_gate : nullptr, stats0, ws0, wsb0, s, &q.a0, &t0);
  if (r <= 0) return r < 0 ? r : N3D_ERR_INVALID;
  r = g16_prepare(g1, dg1, src1, sld1, w1, bias1, dst1, dld1, flags1, gate1, x1 ? x1->relu_src : nullptr, x1 ? x1->rld : 0,
                  x1 ? x1->out_gate : nullptr, stats1, ws1, wsb1, s, &q.a1, &t1);
  if (r <= 0) return r < 0 ? r : N3D_ERR_INVALID;
  const int64_t M0 = (int64_t)g0->B * q.a0.Dd * q.a0.Hd * q.a0.Wd, M1 = (int64_t)g1->B * q.a1.Dd * q.a1.Hd * q.a1.Wd;
  q.gx0 = (int)cdiv(M0, 16); q.gx1 = (int)cdiv(M1, 16);
  q.n0 = q.gx0 * (q.a0.Cd / 16);
  const int n1 = q.gx1 * (q.a1.Cd / 16);
  const size_t shm = (size_t)(p0.ksplit - 1) * 256 * sizeof(float) + (size_t)16 * 16 * 2 * sizeof(double);
  if (p0.ksplit == 16) hipLaunchKernelGGL(conv_gemm16_pair_kernel<16>, dim3((unsigned)(q.n0 + n1)), dim3(1024), shm, s, q);
  else hipLaunchKernelGGL(conv_gemm16_pair_kernel<4>, dim3((unsigned)(q.n0 + n1)), dim3(256), shm, s, q);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("conv(pair) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
  return 1;
}

// n = 3 or 4 forward-type convs (no data-gradient extras) in one launch; 1 = launched, 0 = not foldable (nothing touched)
int mfma_conv_multi_try(int n, const n3d_conv_geom* const* g, const bool* dg, const float* const* src, const int64_t* sld, const float* const* w,
                        const float* const* bias, float* const* dst, const int64_t* dld, const int* flags, const float* const* gate,
                        double* const* stats, void* const* ws, const size_t* wsb, hipStream_t s) {
  if (n < 3 || n > 4) return 0;
  int ksplit = 0;
  for (int i = 0; i < n; ++i) {
    if (vx_plan(g[i]).ok || tile32_tiles(g[i])) return 0;
    const G16Plan p = g16_plan(g[i], dg[i]);
    if (!p.ok || p.ksplit == 1 || (i > 0 && p.ksplit != ksplit)) return 0;
    ksplit = p.ksplit;
    if (sld[i] % 4 != 0 || !aligned16(src[i])) return 0;
    for (int j = 0; j < i; ++j) if (ws[j] == ws[i] && !((flags[j] & flags[i]) & N3D_PREPACKED)) return 0;   // (as mfma_conv_pair_try)
  }
  MultiArgs q;
  int total = 0;
  for (int i = 0; i < 4; ++i) {
    if (i < n) {
      G16Plan t;
      const int r = g16_prepare(g[i], dg[i], src[i], sld[i], w[i], bias[i], dst[i], dld[i], flags[i], gate[i], nullptr, 0, nullptr, stats[i],
                                ws[i], wsb[i], s, &q.a[i], &t);
      if (r <= 0) return r < 0 ? r : N3D_ERR_INVALID;
      const int64_t M = (int64_t)g[i]->B * q.a[i].Dd * q.a[i].Hd * q.a[i].Wd;
      q.gx[i] = (int)cdiv(M, 16);
      q.start[i] = total;
      total += q.gx[i] * (q.a[i].Cd / 16);
    } else {
      q.a[i] = q.a[0]; q.gx[i] = 1; q.start[i] = 0x7fffffff;   // never selected
    }
  }
  q.start[4] = total;
  const size_t shm = (size_t)(ksplit - 1) * 256 * sizeof(float) + (size_t)16 * 16 * 2 * sizeof(double);
  if (ksplit == 16) hipLaunchKernelGGL(conv_gemm16_multi_kernel<16>, dim3((unsigned)total), dim3(1024), shm, s, q);
  else hipLaunchKernelGGL(conv_gemm16_multi_kernel<4>, dim3((unsigned)total), dim3(256), shm, s, q);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("conv(multi) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
  return 1;
}

// returns 1 if handled
int mfma_wgrad_try(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, int flags, const float* in_gate,
                   float* partial, float* pbias, size_t avail_floats, int* nchunks_out, int* ntiles_out, hipStream_t s,
                   Wg16Args* prepared) {
  if (g->depthwise || g->Ci % 16 != 0 || g->Co % 16 != 0) return 0;
  const int taps = g->k * g->k * g->k;
  const int64_t No = (int64_t)g->Do * g->Ho * g->Wo;
  const int64_t total = (int64_t)g->B * No;
  Wg16Args a;
  a.x = x; a.xld = xld; a.Di = g->Di; a.Hi = g->Hi; a.Wi = g->Wi; a.Ci = g->Ci;
  a.dy = dy; a.dyld = dyld; a.Do = g->Do; a.Ho = g->Ho; a.Wo = g->Wo; a.Co = g->Co;
  a.B = g->B; a.k = g->k; a.stride = g->stride; a.dil = g->dil; a.pad = g->pad; a.flags = flags; a.in_gate = in_gate;
  a.tci = g->Ci / 16; a.tco = g->Co / 16;
  const int ntiles = taps * a.tci * a.tco;
  // aim for ~1024 workgroups; each needs at least 64 voxels to amortise the LDS reduction
  constexpr int wg_target = 1024;
  int64_t nch = cdiv(wg_target, ntiles);
  if (nch < 1) nch = 1;
  int64_t maxch = cdiv(total, 64);
  if (nch > maxch) nch = maxch;
  if (nch > 256) nch = 256;
  int64_t chunk = cdiv(cdiv(total, nch), 16) * 16;
  nch = cdiv(total, chunk);
  if ((size_t)nch * ntiles * 256 > avail_floats) return 0;
  a.chunk = chunk; a.partial = partial; a.pbias = pbias;
  if (total + 64 >= (1ll << 31)) return 0;
  a.fNo = FastDiv((uint32_t)No); a.fWo = FastDiv((uint32_t)g->Wo); a.fHo = FastDiv((uint32_t)g->Ho);
  a.fTco = FastDiv((uint32_t)a.tco); a.fTci = FastDiv((uint32_t)a.tci);
  *nchunks_out = (int)nch; *ntiles_out = ntiles;
  if (prepared) { *prepared = a; return 1; }
  hipLaunchKernelGGL(conv_wgrad16_kernel, dim3(ntiles, (unsigned)nch), dim3(256), 0, s, a);
  return 1;
}

// ------------------------------------------------------------------------------------------------
// wgrad_tile16: weight gradient of the tile16 shapes (3x3x3 stride-1 conv, 16 -> 16 channels, many voxels).
//   G[tap][ci][co] = sum_v X[v + tap][ci] * dY[v][co]   as v_mfma_f32_16x16x4_f32 with K = four W-consecutive voxels:
//   A[i = ci][k] = X[v_k + tap][ci]: ONE ds_read_b32 per lane from the LDS halo tile (lane (m, kk) -> channel m of voxel kk of the
//   group: 256 contiguous bytes per wave, conflict-free); B[k][j = co] = dY[v_k][co]: tap-independent, loaded once per tile
//   into 32 registers (one per voxel group).  The 27 taps are split over the four waves (7/7/7/6): no cross-wave reduction, a
//   wave keeps 7 x 2 accumulator tiles (two chains per tap); a workgroup walks tiles_per_wg tiles and leaves one partial slab
//   [27][16][16] + [16] for the common fixed-order finalize.  conv_wgrad16_kernel gathers both operands from global memory
//   4 bytes per lane and voxel, one tap per workgroup: 80 us at (2,16,32^3); this form: see DESIGN.md.
// ------------------------------------------------------------------------------------------------
struct WgT16Args {
  const float* x; int64_t xld; const float* dy; int64_t dyld;
  int D, H, W, B, flags;
  const float* in_gate;
  float* partial;   // [chunks][27 * tci * tco][256]   (tile index (tap * tci + cit) * tco + cot, as conv_wgrad16_kernel)
  float* pbias;     // [chunks][tco][16]
  int tiles_per_sample, tiles_total, tiles_per_wg;
  int Ci, Co;       // multiples of 16: blockIdx.y = cit * tco + cot selects the 16 x 16 channel tile of this workgroup
  const void* zero_page;
};

// TW: tile width -- 16 (a 2 x 4 x 16 tile) or 8 (4 x 4 x 8: the 8^3 level of 128^3 patches, 64 channels, where a 16-wide tile does not
// fit and the gather kernel took 25 us for 226 MFLOP); either way 32 groups of four W-consecutive voxels
template <int DIL, int TW = 16, bool BF = false>
__global__ __launch_bounds__(256, 2) void wgrad_tile16_kernel(WgT16Args a) {
  constexpr int TH = 4, GPR = TW / 4, TD = 32 / GPR / TH;      // groups per row; tile depth
  static_assert(TD * TH * GPR == 32, "32 voxel groups per tile");
  constexpr int LD = TD + 2 * DIL, LH = TH + 2 * DIL, LW = TW + 2 * DIL, NV = LD * LH * LW;
  constexpr int NP = NV * 4, NIT = (NP + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float4 wt16[];   // X halo tile: [NIT * 256] float4, voxel-major 64-byte records
  const float* xl = reinterpret_cast<const float*>(wt16);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, kk = lane >> 4;
  const int tap0 = wave * 7, ntap = wave < 3 ? 7 : 6;
  const int tci = a.Ci >> 4, tco = a.Co >> 4;
  const int cit = (int)blockIdx.y / tco, cot = (int)blockIdx.y - cit * tco;
  const int D = a.D, H = a.H, W = a.W;
  const int tw_n = W / TW, th_n = H / TH;
  const int64_t N = (int64_t)D * H * W;
  const float floor_ = (a.flags & N3D_RELU_IN) ? 0.f : -INFINITY;
  constexpr bool bfmm = BF;     // N3D_MM_BF16: a kernel of its own (the fp32 form keeps its registers)
  f32x4 acc[7][2];
#pragma unroll
  for (int t = 0; t < 7; ++t) { acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  float bsum = 0.f;
  // LDS float index of this lane's A element for group 0, tap offset 0
  const int abase = (kk * 16) + m;
  for (int it = 0; it < a.tiles_per_wg; ++it) {
    const int tg = (int)blockIdx.x * a.tiles_per_wg + it;
    if (tg >= a.tiles_total) break;
    const int b = tg / a.tiles_per_sample, tid = tg - b * a.tiles_per_sample;
    const int w0 = (tid % tw_n) * TW, h0 = ((tid / tw_n) % th_n) * TH, d0 = (tid / (tw_n * th_n)) * TD;
    // B operand: channel m of voxel 4*xq + kk of every row of the tile (32 groups), straight from global memory
    float bv[32];
    {
      const float* dyb = a.dy + ((int64_t)b * N + ((int64_t)d0 * H + h0) * W + w0 + kk) * a.dyld + cot * 16 + m;
#pragma unroll
      for (int gi = 0; gi < 32; ++gi) {
        const int row = gi / GPR, xq = gi % GPR, z = row >> 2, y = row & 3;
        bv[gi] = dyb[(((int64_t)z * H + y) * W + xq * 4) * a.dyld];
      }
    }
    const float gq = a.in_gate ? a.in_gate[(int64_t)b * a.Ci + cit * 16 + m] : 1.f;
    {  // X halo tile (the 16 input channels of this workgroup's tile) by LDS-DMA, as conv_tile16_kernel
      typedef const __attribute__((address_space(1))) void* gptr_t;
      typedef __attribute__((address_space(3))) void* lptr_t;
      const float4* zp = reinterpret_cast<const float4*>(a.zero_page);
      const float* srcb = a.x + (int64_t)b * N * a.xld;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int pc = (i * 4 + wave) * 64 + lane;
        const int v = pc >> 2, q = pc & 3;
        const int x = v % LW, y = (v / LW) % LH, z = v / (LW * LH);
        const int gd = d0 - DIL + z, gh = h0 - DIL + y, gw = w0 - DIL + x;
        const bool ok = v < NV && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        const float* sp = srcb + (((int64_t)gd * H + gh) * W + gw) * a.xld + cit * 16 + q * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)(ok ? reinterpret_cast<const float4*>(sp) : zp), (lptr_t)(wt16 + (i * 4 + wave) * 64), 16, 0, N3D_WGRAD_AUX);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int gi = 0; gi < 32; ++gi) bsum += bv[gi];
    }
    mm_bf16x4 bb[8];
    if constexpr (bfmm) {
#pragma unroll
      for (int g4 = 0; g4 < 8; ++g4) bb[g4] = mm_cvt4(bv[g4 * 4], bv[g4 * 4 + 1], bv[g4 * 4 + 2], bv[g4 * 4 + 3]);
    }
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      if (t < ntap) {
        const int tap = tap0 + t;
        const int kd = tap / 9, kh = (tap - kd * 9) / 3, kw = tap - kd * 9 - kh * 3;
        const float* ap = xl + abase + (((kd * DIL) * LH + kh * DIL) * LW + kw * DIL) * 16;
        if constexpr (bfmm) {
          // N3D_MM_BF16: four voxel groups per 16-deep bf16 MFMA (K index 4 kk + jj = voxel kk of group 4 g4 + jj; dY rounded once per tile)
#pragma unroll
          for (int g4 = 0; g4 < 8; ++g4) {
            float a4[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const int gi = g4 * 4 + jj;
              const int row = gi / GPR, xq = gi % GPR, z = row >> 2, y = row & 3;
              a4[jj] = fmaxf(ap[((z * LH + y) * LW + xq * 4) * 16], floor_) * gq;
            }
            acc[t][g4 & 1] = mm_bf16(mm_cvt4(a4[0], a4[1], a4[2], a4[3]), bb[g4], acc[t][g4 & 1]);
          }
        } else {
#pragma unroll
        for (int gi = 0; gi < 32; ++gi) {
          const int row = gi / GPR, xq = gi % GPR, z = row >> 2, y = row & 3;
          float av = ap[((z * LH + y) * LW + xq * 4) * 16];
          av = fmaxf(av, floor_) * gq;
          acc[t][gi & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[gi], acc[t][gi & 1], 0, 0, 0);
        }
        }
      }
    }
    __syncthreads();   // the next tile's fill overwrites the LDS image
  }
  // D: rows (ci) 4*kk + r, column (co) m  ->  slab position ci*16 + co
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    if (t < ntap) {
      float* p = a.partial + (((int64_t)blockIdx.x * 27 + tap0 + t) * tci * tco + (int64_t)cit * tco + cot) * 256;
#pragma unroll
      for (int r = 0; r < 4; ++r) p[(kk * 4 + r) * 16 + m] = acc[t][0][r] + acc[t][1][r];
    }
  }
  if (wave == 0 && cit == 0) {
    bsum = xsum32_f(xsum16_f(bsum));
    if (kk == 0) a.pbias[((int64_t)blockIdx.x * tco + cot) * 16 + m] = bsum;
  }
}

#define N3D_WGT16_MAX_WG 256
// plan of the LDS-tile weight gradient: chunks (grid.x) and tiles per workgroup; false = shape not served.
// Served: 3x3x3 stride-1 convs (dilation 1 / 2) with channel counts multiples of 16 (<= 64) on tileable volumes with enough tiles x
// channel tiles to fill a good part of the chip -- below that the K-split kernels (one launch for data + weight gradient) win.
bool wgrad_tile16_plan(const n3d_conv_geom* g, int* chunks, int* tiles_per_wg, int* tw_out) {
  constexpr bool off = false;
  constexpr int min_units = 64;   // (256 -> 64: step 1.89 -> 1.86 ms)
  if (off || g->depthwise || g->k != 3 || g->stride != 1 || g->Ci % 16 != 0 || g->Co % 16 != 0 || g->Ci > 64 || g->Co > 64) return false;
  if ((g->dil != 1 && g->dil != 2) || g->pad != g->dil) return false;
  // tile: 2 x 4 x 16 voxels, or 4 x 4 x 8 on volumes 8 wide
  const int tw = g->Wi % 16 == 0 ? 16 : 8, td = tw == 16 ? 2 : 4;
  if (g->Wi % tw != 0 || g->Hi % 4 != 0 || g->Di % td != 0) return false;
  if (tw_out) *tw_out = tw;
  const int64_t tiles_total = (int64_t)(g->Wi / tw) * (g->Hi / 4) * (g->Di / td) * g->B;
  const int combos = (g->Ci / 16) * (g->Co / 16);
  if (tiles_total * combos < min_units || tiles_total >= (1 << 30)) return false;
  int64_t nx = N3D_WGT16_MAX_WG / combos;
  if (nx < 1) nx = 1;
  if (nx > tiles_total) nx = tiles_total;
  const int64_t tpw = cdiv(tiles_total, nx);
  *tiles_per_wg = (int)tpw;
  *chunks = (int)cdiv(tiles_total, tpw);
  return true;
}

// 1 = launched (partial slabs [chunks][27 * tci * tco][256] at `partial`, bias rows [chunks][tco][16] behind them), 0 = shape not served
int wgrad_tile16_try(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, int flags, const float* in_gate,
                     float* partial, size_t avail_floats, int* nchunks_out, float** pbias_out, hipStream_t s) {
  int chunks = 0, tpw = 0, tw = 16;
  if (!wgrad_tile16_plan(g, &chunks, &tpw, &tw)) return 0;
  const int td = tw == 16 ? 2 : 4;
  if (xld % 4 != 0 || !aligned16(x) || (flags & (N3D_SRC_BF16 | N3D_DST_BF16 | N3D_NO_MFMA))) return 0;
  const int tci = g->Ci / 16, tco = g->Co / 16;
  const size_t slabs = (size_t)chunks * 27 * tci * tco;
  if (slabs * 256 + (size_t)chunks * tco * 16 > avail_floats) return 0;
  WgT16Args a;
  a.x = x; a.xld = xld; a.dy = dy; a.dyld = dyld; a.D = g->Di; a.H = g->Hi; a.W = g->Wi; a.B = g->B; a.flags = flags; a.in_gate = in_gate;
  a.partial = partial; a.pbias = partial + slabs * 256;
  a.tiles_per_sample = (g->Wi / tw) * (g->Hi / 4) * (g->Di / td); a.tiles_total = a.tiles_per_sample * g->B; a.tiles_per_wg = tpw;
  a.Ci = g->Ci; a.Co = g->Co;
  a.zero_page = zero_page_ptr();
  if (!a.zero_page) return 0;
  const int d = g->dil;
  const int nv = (td + 2 * d) * (4 + 2 * d) * (tw + 2 * d);
  const size_t shm = (size_t)((nv * 4 + 255) / 256) * 256 * 16;
  const dim3 grid((unsigned)chunks, (unsigned)(tci * tco));
  const bool bf = flags & N3D_MM_BF16;
#define N3D_WT16_LAUNCH(DIL_, TW_) do { if (bf) hipLaunchKernelGGL((wgrad_tile16_kernel<DIL_, TW_, true>), grid, dim3(256), shm, s, a); \
    else hipLaunchKernelGGL((wgrad_tile16_kernel<DIL_, TW_, false>), grid, dim3(256), shm, s, a); } while (0)
  if (tw == 16) {
    if (d == 1) N3D_WT16_LAUNCH(1, 16); else N3D_WT16_LAUNCH(2, 16);
  } else {
    if (d == 1) N3D_WT16_LAUNCH(1, 8); else N3D_WT16_LAUNCH(2, 8);
  }
#undef N3D_WT16_LAUNCH
  *nchunks_out = chunks; *pbias_out = a.pbias;
  return 1;
}

// Data gradient + weight gradient of a non-transposed conv whose channel counts are multiples of 16 and whose data
// gradient is a tiny GEMM (K-split plan), in one launch.  Returns 1 if launched, 0 if the shape does not qualify.
int mfma_bwd_dual_try(const n3d_conv_geom* g, bool transposed, const float* dy, int64_t dyld, const float* wp_packed, float* dx,
                      int64_t dxld, int flags_d, const float* relu_src, int64_t rld, const float* out_gate, const float* x, int64_t xld,
                      int flags_w, const float* in_gate, float* partial, float* pbias, size_t avail_floats, int* nchunks_out,
                      int* ntiles_out, hipStream_t s, DualArgs* prepared, int* ksplit_out) {
  // transposed conv: its data gradient is the FORWARD gather of the geometry (dy lives on the i side), and the weight
  // gradient kernel sees dy as its i-side operand and x as its o-side operand (roles swapped by the caller's convention)
  const bool dgrad_is_data_grad = !transposed;
  if (vx_plan(g).ok) return 0;
  const G16Plan p = g16_plan(g, dgrad_is_data_grad);
  if (!p.ok || p.ksplit == 1) return 0;
  if (dyld % 4 != 0 || !aligned16(dy)) return 0;
  MfArgs a;
  a.src = dy; a.sld = dyld; a.dst = dx; a.dld = dxld; a.bias = nullptr; a.k = g->k; a.flags = flags_d; a.B = g->B;
  a.in_gate = nullptr; a.relu_src = relu_src; a.rld = rld; a.out_gate = out_gate; a.stats = nullptr; a.rows_per_sample = 0;
  if (!dgrad_is_data_grad) { a.Ds = g->Di; a.Hs = g->Hi; a.Ws = g->Wi; a.Cs = g->Ci; a.Dd = g->Do; a.Hd = g->Ho; a.Wd = g->Wo; a.Cd = g->Co;
    a.sn = g->stride; a.off = -g->pad; a.dt = g->dil; a.den = 1; }
  else { a.Ds = g->Do; a.Hs = g->Ho; a.Ws = g->Wo; a.Cs = g->Co; a.Dd = g->Di; a.Hd = g->Hi; a.Wd = g->Wi; a.Cd = g->Ci;
    a.sn = 1; a.off = g->pad; a.dt = -g->dil; a.den = g->stride; }
  const int64_t Nd = (int64_t)a.Dd * a.Hd * a.Wd;
  if ((int64_t)g->B * Nd >= (1ll << 31)) return 0;
  a.fNd = FastDiv((uint32_t)Nd); a.fWd = FastDiv((uint32_t)a.Wd); a.fHd = FastDiv((uint32_t)a.Hd); a.fC16 = FastDiv((uint32_t)(a.Cs / 16));
  a.wp = wp_packed;
  DualArgs q;
  q.d = a;
  const int wok = transposed ? mfma_wgrad_try(g, dy, dyld, x, xld, flags_w, in_gate, partial, pbias, avail_floats, nchunks_out, ntiles_out, s, &q.w)
                             : mfma_wgrad_try(g, x, xld, dy, dyld, flags_w, in_gate, partial, pbias, avail_floats, nchunks_out, ntiles_out, s, &q.w);
  if (!wok) return 0;
  const int64_t M = (int64_t)g->B * Nd;
  q.gxA = (int)cdiv(M, 16);
  q.nA = q.gxA * (a.Cd / 16);
  q.ntilesB = *ntiles_out;
  q.nB = *ntiles_out * *nchunks_out;
  if (prepared) { *prepared = q; *ksplit_out = p.ksplit; return 1; }
  const int units = p.ksplit == 16 ? 4 : 1;
  const size_t shm_a = (size_t)(p.ksplit - 1) * 256 * sizeof(float) + (size_t)16 * 16 * 2 * sizeof(double);
  const size_t shm_b = (size_t)units * (3 * 64 * 16 + 4 * 16 * 4);
  const size_t shm = shm_a > shm_b ? shm_a : shm_b;
  const dim3 grid((unsigned)(q.nA + cdiv(q.nB, units)));
  if (p.ksplit == 16) hipLaunchKernelGGL(conv_bwd16_dual_kernel<16>, grid, dim3(1024), shm, s, q);
  else hipLaunchKernelGGL(conv_bwd16_dual_kernel<4>, grid, dim3(256), shm, s, q);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("conv(bwd dual) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
  return 1;
}

// can the backward of these two convs share one launch?  (pure function of the geometries)
int mfma_bwd_quad_ok(const n3d_conv_geom* g0, bool t0, const n3d_conv_geom* g1, bool t1) {
  if (vx_plan(g0).ok || vx_plan(g1).ok) return 0;
  if (g0->depthwise || g1->depthwise || g0->Ci % 16 || g0->Co % 16 || g1->Ci % 16 || g1->Co % 16) return 0;
  // a conv whose weight gradient the LDS-tile kernel serves keeps it in every schedule (the same arithmetic whether the weight
  // gradients are deferred to the side stream or launched in place)
  int ch = 0, tpw = 0;
  if ((!t0 && wgrad_tile16_plan(g0, &ch, &tpw, nullptr)) || (!t1 && wgrad_tile16_plan(g1, &ch, &tpw, nullptr))) return 0;
  const G16Plan p0 = g16_plan(g0, !t0), p1 = g16_plan(g1, !t1);
  return p0.ok && p1.ok && p0.ksplit == p1.ksplit && p0.ksplit != 1;
}

struct BwdOne {  // one conv's backward operands for the quad launch
  const n3d_conv_geom* g; bool transposed; const float* dy; int64_t dyld; const float* wp; float* dx; int64_t dxld; int flags_d;
  const float* relu_src; int64_t rld; const float* out_gate; const float* x; int64_t xld; int flags_w; const float* in_gate;
  float* partial; float* pbias; size_t avail; int nch, ntl;
};

int mfma_bwd_quad_try(BwdOne* c0, BwdOne* c1, hipStream_t s) {
  QuadArgs z;
  int k0 = 0, k1 = 0;
  BwdOne* cs[2] = {c0, c1};
  DualArgs* qs[2] = {&z.q0, &z.q1};
  int* ks[2] = {&k0, &k1};
  for (int i = 0; i < 2; ++i) {
    BwdOne* c = cs[i];
    const int r = mfma_bwd_dual_try(c->g, c->transposed, c->dy, c->dyld, c->wp, c->dx, c->dxld, c->flags_d, c->relu_src, c->rld, c->out_gate,
                                    c->x, c->xld, c->flags_w, c->in_gate, c->partial, c->pbias, c->avail, &c->nch, &c->ntl, s, qs[i], ks[i]);
    if (r != 1) return r;
  }
  if (k0 != k1) return 0;
  const int units = k0 == 16 ? 4 : 1;
  z.n0 = z.q0.nA + (int)cdiv(z.q0.nB, units);
  const int n1 = z.q1.nA + (int)cdiv(z.q1.nB, units);
  const size_t shm_a = (size_t)(k0 - 1) * 256 * sizeof(float) + (size_t)16 * 16 * 2 * sizeof(double);
  const size_t shm_b = (size_t)units * (3 * 64 * 16 + 4 * 16 * 4);
  const size_t shm = shm_a > shm_b ? shm_a : shm_b;
  if (k0 == 16) hipLaunchKernelGGL(conv_bwd16_quad_kernel<16>, dim3((unsigned)(z.n0 + n1)), dim3(1024), shm, s, z);
  else hipLaunchKernelGGL(conv_bwd16_quad_kernel<4>, dim3((unsigned)(z.n0 + n1)), dim3(256), shm, s, z);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("conv(bwd quad) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
  return 1;
}


}  // namespace n3d
