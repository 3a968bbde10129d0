// bf16-storage forms of the vox64 family (BASELINE configs[4]: 4x128^3 patches, where every C <= 8 level is HBM-bound):
// the 3x3x3 stride-1 (dilation 1 / 2) convolution with C = 4 or 8 channels, forward and data gradient, on tensors whose
// activations are stored as bfloat16.
//   * Same tile geometry as conv_vox64_kernel (conv_mfma.hip): one 16-byte LDS slot per voxel, halo tile filled by LDS-DMA
//     (global_load_lds_dwordx4, zero padding from a zero page, XCD-aware tile order).  A C = 8 voxel IS 16 bytes; a C = 4 voxel is
//     8 bytes, the DMA simply brings the 8 bytes behind it along (the next voxel, or the neighbouring channels of a wider buffer:
//     activation buffers are allocated with 16 bytes of slack, kernels.empty_ndhwc) and the kernel reads the low half of the slot.
//   * The matrix cores take the bf16 operands as they are: v_mfma_f32_4x4x4_16b_bf16 -- 16 blocks of (4 output channels x 4 input
//     channels) x (4 input channels x 4 voxels) -- so ONE instruction does what four f32 4x4x1 instructions do in the fp32 kernel.
//     A operand = the weights (rounded to bf16 by the pack kernel, [tap][cd][cs]), B operand = the 8 bytes of a voxel's channel
//     quad straight from LDS (no conversion instruction anywhere); fp32 accumulators, fp32 bias, fp32 GroupNorm statistics.
//   * Output rounded to bf16 on store (v_cvt_pk_bf16_f32, round-to-nearest-even).
// The MFMA phase shrinks 4x against the fp32 kernel, the fill moves half the bytes: the kernel is bound by the LDS-DMA fill and
// HBM, which is where this configuration lives.
#include "n3d_common.h"

namespace n3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4v __attribute__((ext_vector_type(4)));

__device__ float4 n3d_zero_page16[1];  // zero-initialised; source of the zero padding for LDS-DMA fills
static const void* zero_page16_ptr() {
  static thread_local const void* p = nullptr;
  static thread_local int dev = -1;
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) return nullptr;
  if (!p || d != dev) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(n3d_zero_page16)) != hipSuccess) return nullptr;
    p = q; dev = d;
  }
  return p;
}

// Wq[tap][cd][cs] in bf16; forward: cd = co, cs = ci; data gradient: cd = ci, cs = co, taps flipped (26 - tap)
__global__ void pack_vox16_kernel(const float* __restrict__ w, bf16_t* __restrict__ wq, int C, int data_grad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 27 * C * C) return;
  const int cs = i % C, cd = (i / C) % C, tap = i / (C * C);
  const int co = data_grad ? cs : cd, ci = data_grad ? cd : cs, t2 = data_grad ? 26 - tap : tap;
  st1(wq + i, w[((int64_t)co * C + ci) * 27 + t2]);
}

struct Vx16Args {
  const bf16_t* src; int64_t sld;
  bf16_t* dst; int64_t dld;
  const bf16_t* wq;     // packed [27][C (cd)][C (cs)] bf16
  const float* bias;
  int D, H, W, flags;
  double* stats; int rows_per_sample;
  int tiles;
  const void* zero_page;
};

__device__ __forceinline__ f32x4 mfma_bf16_4x4x4(const uint2 a, const uint2 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(bf16x4v, a), __builtin_bit_cast(bf16x4v, b), c, 0, 0, 0);
}

template <int C, int TD, int DIL, int NW>
__global__ __launch_bounds__(64 * NW, 2) void conv_vox64b_kernel(Vx16Args a) {
  constexpr int HF = C / 4, GH = 4 * NW, GW = 16;
  constexpr int LD = TD + 2 * DIL, LH = GH + 2 * DIL, LW = GW + 2 * DIL;
  constexpr int PLANE = LH * LW, NPOS = (PLANE + 63) / 64, PSTRIDE = NPOS * 64;
  constexpr int NW4 = (27 * C * C * 2 + 15) / 16, NWI = (NW4 + 63) / 64;   // 16-byte pieces of the packed weights
  extern __shared__ __attribute__((aligned(16))) uint4 vlds16[];  // tile [LD][PSTRIDE] slots, then the weights
  uint4* tile = vlds16;
  uint4* wl = vlds16 + LD * PSTRIDE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int wg = blockIdx.x;
  {  // XCD-aware placement (see conv_vox64_kernel)
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  const int b = wg / a.tiles;
  const int tile_id = wg - b * a.tiles;
  const int tw_n = a.W / GW, th_n = a.H / GH;
  int bx = tile_id;
  const int w0 = (bx % tw_n) * GW; bx /= tw_n;
  const int h0 = (bx % th_n) * GH;
  const int d0 = (bx / th_n) * TD;
  const int64_t N = (int64_t)a.D * a.H * a.W;
  const bf16_t* srcb = a.src + (int64_t)b * N * a.sld;
  const int j = lane & 3;
  // lane -> voxel as in conv_vox64_kernel: odd rows rotated so that the 16-lane groups of ds_read_b128 (C = 8) hit distinct banks
  const int hh = lane >> 4, ww = ((lane & 15) - (hh & 1) * (LW % 16)) & 15;
  const int hrow = 4 * wave + hh;
  bf16_t* dstb = a.dst + (int64_t)b * N * a.dld;
  const bool accum = a.flags & N3D_ACCUMULATE;
  const int64_t vox_off = ((int64_t)(h0 + 4 * wave + hh) * a.W + w0 + ww);
  float4 biasv[HF], prevv[TD][HF];
#pragma unroll
  for (int hf = 0; hf < HF; ++hf) {
    biasv[hf] = a.bias ? *reinterpret_cast<const float4*>(a.bias + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < TD; ++g)
      prevv[g][hf] = accum ? ld4(dstb + (((int64_t)(d0 + g) * a.H * a.W) + vox_off) * a.dld + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const uint4* __restrict__ wq4 = reinterpret_cast<const uint4*>(a.wq);
    const uint4* zp = reinterpret_cast<const uint4*>(a.zero_page);
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
      const bf16_t* prow = srcb + ((int64_t)gh * a.W + gw) * a.sld;
      static_assert(LD % NW == 0, "tile depth must split evenly over the waves");
#pragma unroll
      for (int m = 0; m < LD / NW; ++m) {
        const int dz = m * NW + (NW > 1 ? wave : 0);
        const int gd = d0 - DIL + dz;
        const bool inb = okp && gd >= 0 && gd < a.D;
        const bf16_t* p = prow + gd * pstride;
        // 16 bytes from the voxel's address: the whole voxel (C = 8) or the voxel and the 8 bytes behind it (C = 4)
        __builtin_amdgcn_global_load_lds((gptr_t)(inb ? reinterpret_cast<const void*>(p) : reinterpret_cast<const void*>(zp)),
                                         (lptr_t)(tile + dz * PSTRIDE + i * 64), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (NW > 1) __syncthreads();

  f32x4 acc[TD][HF];
#pragma unroll
  for (int hf = 0; hf < HF; ++hf) {
    const f32x4 bv = {biasv[hf].x, biasv[hf].y, biasv[hf].z, biasv[hf].w};
#pragma unroll
    for (int g = 0; g < TD; ++g) acc[g][hf] = bv;
  }
  float cs[HF][4], cq[HF][4];
#pragma unroll
  for (int hf = 0; hf < HF; ++hf)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[hf][r] = cq[hf][r] = 0.f;
  bf16_t* const o_plane0 = dstb + ((int64_t)d0 * a.H * a.W + vox_off) * a.dld;
  const int64_t o_pstride = (int64_t)a.H * a.W * a.dld;
  auto emit_plane = [&](int g) {
    bf16_t* o = o_plane0 + g * o_pstride;
#pragma unroll
    for (int hf = 0; hf < HF; ++hf) {
      const f32x4 v = acc[g][hf];
#pragma unroll
      for (int r = 0; r < 4; ++r) { cs[hf][r] += v[r]; cq[hf][r] = fmaf(v[r], v[r], cq[hf][r]); }
      float4 w4 = make_float4(v[0], v[1], v[2], v[3]);
      { const float4 pv = prevv[g][hf]; w4.x += pv.x; w4.y += pv.y; w4.z += pv.z; w4.w += pv.w; }
      st4(o + hf * 4, w4);
    }
  };
  const uint2* tile2 = reinterpret_cast<const uint2*>(tile);   // slot s: low half at [2*s], high half at [2*s + 1]
  const uint2* wl2 = reinterpret_cast<const uint2*>(wl);
  if constexpr (C == 4) {
    // all 27 weight quads (one 8-byte A operand each) in registers; INPUT-PLANE major: plane dz feeds the output planes
    // dz, dz - DIL, dz - 2*DIL and an output plane is stored as soon as its last input plane is done
    uint2 wr[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wr[t] = wl2[t * 4 + j];
    f32x4 acc2[TD];
#pragma unroll
    for (int g = 0; g < TD; ++g) acc2[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    uint2 avb[2][9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) avb[0][t9] = tile2[2 * ((hrow + (t9 / 3) * DIL) * LW + (ww + (t9 % 3) * DIL))];
#pragma unroll
    for (int dz = 0; dz < LD; ++dz) {
      if (dz + 1 < LD) {
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
          avb[(dz + 1) & 1][t9] = tile2[2 * ((dz + 1) * PSTRIDE + (hrow + (t9 / 3) * DIL) * LW + (ww + (t9 % 3) * DIL))];
      }
      const uint2* av = avb[dz & 1];
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
          const int g = dz - kd * DIL;
          if (g >= 0 && g < TD) {
            // two accumulator chains per output plane (even / odd tap)
            if (t9 & 1) acc2[g] = mfma_bf16_4x4x4(wr[kd * 9 + t9], av[t9], acc2[g]);
            else acc[g][0] = mfma_bf16_4x4x4(wr[kd * 9 + t9], av[t9], acc[g][0]);
          }
        }
      }
      if (dz - 2 * DIL >= 0) {
        acc[dz - 2 * DIL][0] += acc2[dz - 2 * DIL];
        emit_plane(dz - 2 * DIL);
      }
    }
  } else {
    // C = 8: (kh, kw) outer with the three kd weight sets in registers, input plane inner; a voxel's 16 bytes are two K = 4
    // halves, a weight row (tap, cd) likewise
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int kh = t9 / 3, kw = t9 % 3;
      uint4 wr[3][HF];
#pragma unroll
      for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int hf = 0; hf < HF; ++hf) wr[kd][hf] = wl[(kd * 9 + t9) * C + hf * 4 + j];
      const int base = (hrow + kh * DIL) * LW + (ww + kw * DIL);
      uint4 av[LD];
#pragma unroll
      for (int dz = 0; dz < LD; ++dz) av[dz] = tile[dz * PSTRIDE + base];
#pragma unroll
      for (int dz = 0; dz < LD; ++dz) {
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
          const int g = dz - kd * DIL;
          if (g >= 0 && g < TD) {
#pragma unroll
            for (int hf = 0; hf < HF; ++hf) {
              acc[g][hf] = mfma_bf16_4x4x4(make_uint2(wr[kd][hf].x, wr[kd][hf].y), make_uint2(av[dz].x, av[dz].y), acc[g][hf]);
              acc[g][hf] = mfma_bf16_4x4x4(make_uint2(wr[kd][hf].z, wr[kd][hf].w), make_uint2(av[dz].z, av[dz].w), acc[g][hf]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int g = 0; g < TD; ++g) emit_plane(g);
  }
  // GroupNorm partial row of this tile (as conv_vox64_kernel)
  if (a.stats) {
    const bool odd = lane & 1, hi = lane & 2;
#pragma unroll
    for (int hf = 0; hf < HF; ++hf) {
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

struct Vx16Plan { bool ok; int C, td, dil, tiles, nw; size_t lds; };

static Vx16Plan vx16_plan(const n3d_conv_geom* g) {
  Vx16Plan p; p.ok = false;
  if (g->depthwise || g->k != 3 || g->stride != 1 || g->Ci != g->Co || (g->Ci != 4 && g->Ci != 8)) return p;
  if (!(g->dil == 1 || g->dil == 2) || g->pad != g->dil) return p;
  const int W = g->Wi, H = g->Hi, D = g->Di;
  if (W % 16 != 0 || H % 4 != 0) return p;
  const int64_t groups = (int64_t)g->B * D * (H / 4) * (W / 16);
  int td = 1;
  if (D % 4 == 0 && groups / 4 >= 2048 && g->Ci == 4) td = 4;
  else if (D % 2 == 0 && groups / 2 >= 2048) td = 2;
  p.ok = true; p.C = g->Ci; p.td = td; p.dil = g->dil;
  p.nw = (g->Ci == 4 && td == 4 && H % 8 == 0 && g->dil == 2) ? 2 : 1;
  p.tiles = (W / 16) * (H / (4 * p.nw)) * (D / td);
  const size_t pstride = ((size_t)(4 * p.nw + 2 * g->dil) * (16 + 2 * g->dil) + 63) / 64 * 64;
  const size_t wslots = (((size_t)27 * g->Ci * g->Ci * 2 + 15) / 16 + 63) / 64 * 64;
  p.lds = ((size_t)(td + 2 * g->dil) * pstride + wslots) * 16;
  return p;
}

template <int C, int TD, int DIL>
static void launch_vox16_t(const Vx16Args& a, const Vx16Plan& p, int B, hipStream_t s) {
  if constexpr (C == 4 && TD == 4) {
    if (p.nw == 2) { hipLaunchKernelGGL((conv_vox64b_kernel<C, TD, DIL, 2>), dim3(p.tiles * B), dim3(128), p.lds, s, a); return; }
  }
  hipLaunchKernelGGL((conv_vox64b_kernel<C, TD, DIL, 1>), dim3(p.tiles * B), dim3(64), p.lds, s, a);
}

template <int C>
static void launch_vox16_c(const Vx16Args& a, const Vx16Plan& p, int B, hipStream_t s) {
  if (p.dil == 1) {
    if (p.td == 4) return launch_vox16_t<C, 4, 1>(a, p, B, s);
    if (p.td == 2) return launch_vox16_t<C, 2, 1>(a, p, B, s);
    return launch_vox16_t<C, 1, 1>(a, p, B, s);
  }
  if (p.td == 4) return launch_vox16_t<C, 4, 2>(a, p, B, s);
  if (p.td == 2) return launch_vox16_t<C, 2, 2>(a, p, B, s);
  return launch_vox16_t<C, 1, 2>(a, p, B, s);
}

// ---- interface to conv_generic.hip -------------------------------------------------------------------------------
// weights of this conv are packed as layout 4 (bf16 [27][cd][cs]) when both tensors are bf16 and the shape is served here
bool vox16_serves(const n3d_conv_geom* g, int flags) {
  return (flags & N3D_SRC_BF16) && (flags & N3D_DST_BF16) && !(flags & N3D_NO_MFMA) && vx16_plan(g).ok;
}

int vox16_stats_rows(const n3d_conv_geom* g, int flags) {
  if (!vox16_serves(g, flags)) return 0;
  const Vx16Plan v = vx16_plan(g);
  return v.tiles * v.nw;
}

void vox16_pack(const float* w, void* wq, int C, int data_grad, hipStream_t s) {
  hipLaunchKernelGGL(pack_vox16_kernel, dim3((unsigned)cdiv(27 * C * C, 256)), dim3(256), 0, s, w, (bf16_t*)wq, C, data_grad);
}

// 1 = launched, 0 = not applicable, < 0 error
int vox16_conv_try(const n3d_conv_geom* g, bool data_grad, const void* src, int64_t sld, const float* w, const float* bias, void* dst,
                   int64_t dld, int flags, const float* in_gate, const void* relu_src, const float* out_gate, double* stats, void* ws,
                   size_t ws_bytes, hipStream_t s) {
  if (!vox16_serves(g, flags)) return 0;
  const Vx16Plan v = vx16_plan(g);
  const bool extras = in_gate || relu_src || out_gate || (flags & N3D_RELU_IN);
  const bool aligned = sld % 4 == 0 && dld % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 7) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0 &&
                       (v.C == 4 || (sld % 8 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0));
  if (extras || !aligned) {
    if (stats || (flags & N3D_PREPACKED)) { set_error("conv(vox64b): gate / relu extras or an unaligned tensor on a shape whose statistics rows / packed weights assume this kernel"); return N3D_ERR_UNSUPPORTED; }
    return 0;
  }
  const size_t need = (size_t)27 * v.C * v.C * 2;
  if (!ws || ws_bytes < need) { set_error("conv(vox64b): workspace too small"); return N3D_ERR_WORKSPACE; }
  if (!(flags & N3D_PREPACKED)) vox16_pack(w, ws, v.C, data_grad ? 1 : 0, s);
  Vx16Args a;
  a.src = (const bf16_t*)src; a.sld = sld; a.dst = (bf16_t*)dst; a.dld = dld; a.wq = (const bf16_t*)ws; a.bias = bias;
  a.D = g->Di; a.H = g->Hi; a.W = g->Wi; a.flags = flags; a.stats = stats; a.rows_per_sample = v.tiles * v.nw; a.tiles = v.tiles;
  a.zero_page = zero_page16_ptr();
  if (!a.zero_page) { set_error("conv(vox64b): zero page symbol unavailable"); return N3D_ERR_HIP; }
  if (v.C == 4) launch_vox16_c<4>(a, v, g->B, s); else launch_vox16_c<8>(a, v, g->B, s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("conv(vox64b) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
  return 1;
}

}  // namespace n3d
