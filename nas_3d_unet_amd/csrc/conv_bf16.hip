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
#include <stdlib.h>
#include "n3d_common.h"

namespace n3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
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

// Wq[tap][cd][cs] in bf16; forward: cd = co, cs = ci; data_grad 1 (stride-1 data gradient): cd = ci, cs = co, taps flipped
// (26 - tap); data_grad 2 (gather form of the stride-2 data gradient / transposed forward): channels transposed, taps as they are
__global__ void pack_vox16_kernel(const float* __restrict__ w, bf16_t* __restrict__ wq, int C, int data_grad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 27 * C * C) return;
  const int cs = i % C, cd = (i / C) % C, tap = i / (C * C);
  const int co = data_grad ? cs : cd, ci = data_grad ? cd : cs, t2 = data_grad == 1 ? 26 - tap : tap;
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
  FastDiv fT, fTw, fTh;      // workgroup -> (sample, tile column, tile row) without run-time divisions: see VxArgs (conv_mfma.hip)
};

__device__ __forceinline__ f32x4 mfma_bf16_4x4x4(const uint2 a, const uint2 b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(bf16x4v, a), __builtin_bit_cast(bf16x4v, b), c, 0, 0, 0);
}

// P2 (round 3; C = 4, DENSE source tensors only: voxel pitch 4 elements, 16-byte aligned): the LDS image is dense -- 8 bytes per
// voxel, rows of 20 voxels starting at the even column w0 - 2 -- and one DMA lane fetches a PAIR of voxels (16 contiguous bytes).
// The one-slot-per-voxel image moves 16 bytes per 8-byte voxel (half of every fetch is the neighbour) and reads the low half of
// every slot (2-way bank conflict on each read); the dense image halves the fill (60 DMA lanes per plane instead of 108: one
// instruction instead of two) and reads consecutive 8-byte voxels.  Slices of a wider buffer (the node slices of a cell output,
// voxel pitch 24 bytes) cannot be fetched in pairs and keep the slot image.
// ACCM: -1 = N3D_ACCUMULATE is read from the flags; 0 / 1 = compiled in (the 8-plane form: the forward launch carries no operand
// fetch of a previous value and none of its address arithmetic)
template <int C, int TD, int DIL, int NW, bool P2 = false, int ACCM = -1>
__global__ __launch_bounds__(64 * NW, 2) void conv_vox64b_kernel(Vx16Args a) {
  N3D_CHAIN_PRIO();
  static_assert(!P2 || C == 4, "the dense two-voxels-per-slot image is the C = 4 form");
  constexpr int HF = C / 4, GH = 4 * NW, GW = 16;
  constexpr int LD = TD + 2 * DIL, LH = GH + 2 * DIL, LW = GW + 2 * DIL;
  constexpr int LWS = P2 ? 10 : LW;                // 16-byte slots per tile row
  constexpr int LW8 = P2 ? 20 : 2 * LW;            // row pitch in 8-byte units
  constexpr int WOFF = P2 ? (2 - DIL) : 0;         // column of the halo's first voxel inside a row of the dense image
  constexpr int PLANE = LH * LWS, NPOS = (PLANE + 63) / 64, PSTRIDE = NPOS * 64;
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
  uint32_t ub, utile, ubx, uw, ud, uh;
  a.fT.divmod((uint32_t)wg, ub, utile);
  a.fTw.divmod(utile, ubx, uw);
  a.fTh.divmod(ubx, ud, uh);
  const int b = (int)ub, tile_id = (int)utile;
  const int w0 = (int)uw * GW, h0 = (int)uh * GH, d0 = (int)ud * TD;
  const int64_t N = (int64_t)a.D * a.H * a.W;
  const bf16_t* srcb = a.src + (int64_t)b * N * a.sld;
  const int j = lane & 3;
  // lane -> voxel as in conv_vox64_kernel: odd rows rotated so that the 16-lane groups of ds_read_b128 (C = 8) hit distinct banks
  const int hh = lane >> 4, ww = ((lane & 15) - (hh & 1) * (LW % 16)) & 15;
  const int hrow = 4 * wave + hh;
  bf16_t* dstb = a.dst + (int64_t)b * N * a.dld;
  const bool accum = ACCM < 0 ? bool(a.flags & N3D_ACCUMULATE) : bool(ACCM);
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
    // Fill by LDS-DMA through BUFFER loads (round 5; it was global_load_lds with a per-lane 64-bit address and a zero page).  A raw
    // buffer load returns 0 for an offset >= num_records, which is the conv's zero padding for free: the lane's byte offset inside
    // its plane -- or OOB for a halo position outside H x W -- is computed ONCE per tile position, the plane's offset is a scalar
    // (soffset), and a plane outside D takes a resource with num_records = 0 (a scalar select).  Per DMA instruction: no vector ALU
    // work at all (the pointer form cost ~10 VALU instructions each: 64-bit multiply-add, two selects against the zero page, the
    // bounds tests -- PMC before: 2.7 VALU per MFMA, profiles/r03_pmc_conv_vox64b_bf16_2x4x128.json).
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr uint32_t OOB = 0x80000000u;         // >= num_records below, with or without the scalar offset
    constexpr int RSRC_FLAGS = 0x00020000;        // gfx9 raw buffer: 32-bit data format, no swizzle
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.wq), 0, NW4 * 16, RSRC_FLAGS);
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      if (NW == 1 || (i % NW) == wave)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(wl + i * 64), 16, (lane + i * 64) * 16, 0, 0, 0);
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(srcb), 0, 0x7fffffff, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(srcb), 0, 0, RSRC_FLAGS);
    const int pstride_b = a.H * a.W * (int)a.sld * 2;      // bytes per plane (a sample stays below 2 GB: checked by the launcher)
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
      const int pos = lane + i * 64;
      const int wx = pos % LWS, hy = pos / LWS;
      const int gh = h0 - DIL + hy, gw = P2 ? w0 - 2 + 2 * wx : w0 - DIL + wx;   // P2: an even column, the pair never straddles the volume's edge
      const bool okp = pos < PLANE && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
      const uint32_t voff = okp ? (uint32_t)((gh * a.W + gw) * (int)a.sld * 2) : OOB;
      static_assert(LD % NW == 0, "tile depth must split evenly over the waves");
#pragma unroll
      for (int m = 0; m < LD / NW; ++m) {
        const int dz = m * NW + (NW > 1 ? wave : 0);
        const int gd = d0 - DIL + dz;
        const bool ind = gd >= 0 && gd < a.D;          // wave-uniform
        // 16 bytes from the voxel's address: the whole voxel (C = 8) or the voxel and the 8 bytes behind it (C = 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ind ? rs : rz, (lptr_t)(tile + dz * PSTRIDE + i * 64), 16, (int)voff, ind ? gd * pstride_b : 0, 0, 0);
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
  // statistics rows: one per 4 output planes, so that a TD = 8 tile writes exactly the two rows (in the same summation order) that
  // the two TD = 4 tiles it replaces would write -- n3d_conv_stats_rows does not depend on which form runs
  constexpr int SR = TD == 8 ? 2 : 1;
  // (packed pairs: v_pk_add_f32 / v_pk_fma_f32 -- four instead of eight vector instructions per plane and channel quad)
  f32x2 cs2[SR][HF][2], cq2[SR][HF][2];
#pragma unroll
  for (int sr = 0; sr < SR; ++sr)
#pragma unroll
    for (int hf = 0; hf < HF; ++hf)
#pragma unroll
      for (int r = 0; r < 2; ++r) cs2[sr][hf][r] = cq2[sr][hf][r] = (f32x2){0.f, 0.f};
  bf16_t* const o_plane0 = dstb + ((int64_t)d0 * a.H * a.W + vox_off) * a.dld;
  const int64_t o_pstride = (int64_t)a.H * a.W * a.dld;
  auto emit_plane = [&](int g) {
    bf16_t* o = o_plane0 + g * o_pstride;
#pragma unroll
    for (int hf = 0; hf < HF; ++hf) {
      const f32x4 v = acc[g][hf];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const f32x2 v2 = {v[2 * r], v[2 * r + 1]};
        cs2[SR == 2 ? g / 4 : 0][hf][r] += v2;
        cq2[SR == 2 ? g / 4 : 0][hf][r] = __builtin_elementwise_fma(v2, v2, cq2[SR == 2 ? g / 4 : 0][hf][r]);
      }
      float4 w4 = make_float4(v[0], v[1], v[2], v[3]);
      if (accum) { const float4 pv = prevv[g][hf]; w4.x += pv.x; w4.y += pv.y; w4.z += pv.z; w4.w += pv.w; }   // (uniform branch)
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
    // 8-byte index of voxel (row, col) of plane dz: slot image = low half of slot row * LW + col; dense image = row * 20 + col + WOFF
    auto vidx = [&](int dz, int row, int col) { return 2 * dz * PSTRIDE + row * LW8 + (P2 ? col + WOFF : 2 * col); };
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) avb[0][t9] = tile2[vidx(0, hrow + (t9 / 3) * DIL, ww + (t9 % 3) * DIL)];
#pragma unroll
    for (int dz = 0; dz < LD; ++dz) {
      if (dz + 1 < LD) {
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9)
          avb[(dz + 1) & 1][t9] = tile2[vidx(dz + 1, hrow + (t9 / 3) * DIL, ww + (t9 % 3) * DIL)];
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
    float cs[SR][HF][4], cq[SR][HF][4];
#pragma unroll
    for (int sr = 0; sr < SR; ++sr)
#pragma unroll
      for (int hf = 0; hf < HF; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) { cs[sr][hf][r] = cs2[sr][hf][r >> 1][r & 1]; cq[sr][hf][r] = cq2[sr][hf][r >> 1][r & 1]; }
    const bool odd = lane & 1, hi = lane & 2;
#pragma unroll
    for (int sr = 0; sr < SR; ++sr)
#pragma unroll
    for (int hf = 0; hf < HF; ++hf) {
      float u[2], uq[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float keep = odd ? cs[sr][hf][2 + k] : cs[sr][hf][k], send = odd ? cs[sr][hf][k] : cs[sr][hf][2 + k];
        u[k] = keep + dpp_f<0xB1>(send);
        const float keepq = odd ? cq[sr][hf][2 + k] : cq[sr][hf][k], sendq = odd ? cq[sr][hf][k] : cq[sr][hf][2 + k];
        uq[k] = keepq + dpp_f<0xB1>(sendq);
      }
      float v1 = (hi ? u[1] : u[0]) + dpp_f<0x4E>(hi ? u[0] : u[1]);
      float v2 = (hi ? uq[1] : uq[0]) + dpp_f<0x4E>(hi ? uq[0] : uq[1]);
      v1 = wave_classsum_f(v1, 4); v2 = wave_classsum_f(v2, 4);
      if (lane < 4) {
        const int ch = (lane & 1) * 2 + (lane >> 1);
        // row of the TD = 4 tile this half stands for: its D block is 2 * (d0 / 8) + sr, i.e. tile id + sr * (tiles per D block) + ...
        int row = tile_id * NW + wave;
        if (SR == 2) row = (((d0 / 4) + sr) * (int)a.fTh.d + (int)uh) * (int)a.fTw.d + (int)uw;     // (fTh.d / fTw.d: tile rows / columns)
        double* o = a.stats + (((int64_t)b * a.rows_per_sample + row) * C + hf * 4 + ch) * 2;
        reinterpret_cast<double2*>(o)[0] = make_double2((double)v1, (double)v2);
      }
    }
  }
}


// ---- operand helpers shared by the stride-2 / transposed forms ----------------------------------------------------
// weights of one tap for this lane's output-channel row(s): [hf = output quad][kh = K half] 8-byte A operands
template <int C>
struct W16 { uint2 k[C / 4][C / 4]; };
template <int C>
__device__ __forceinline__ W16<C> ldw16(const uint4* wl, int tap, int j) {
  W16<C> w;
  if constexpr (C == 4) {
    w.k[0][0] = reinterpret_cast<const uint2*>(wl)[tap * 4 + j];
  } else {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const uint4 q = wl[tap * 8 + hf * 4 + j];
      w.k[hf][0] = make_uint2(q.x, q.y); w.k[hf][1] = make_uint2(q.z, q.w);
    }
  }
  return w;
}
// one voxel's channels from its LDS slot: 8 bytes (C = 4, low half of the slot) or 16 bytes
template <int C>
__device__ __forceinline__ uint4 rd_slot(const uint4* tile, int idx) {
  if constexpr (C == 4) {
    const uint2 v = reinterpret_cast<const uint2*>(tile)[2 * idx];
    return make_uint4(v.x, v.y, 0u, 0u);
  } else {
    return tile[idx];
  }
}
template <int C>
__device__ __forceinline__ void mac16(f32x4 (&acc)[C / 4], const W16<C>& w, const uint4 x) {
#pragma unroll
  for (int hf = 0; hf < C / 4; ++hf) {
    acc[hf] = mfma_bf16_4x4x4(w.k[hf][0], make_uint2(x.x, x.y), acc[hf]);
    if constexpr (C == 8) acc[hf] = mfma_bf16_4x4x4(w.k[hf][1], make_uint2(x.z, x.w), acc[hf]);
  }
}
// GroupNorm partial row of a tile from the per-lane sums (as conv_vox64_kernel)
template <int C>
__device__ __forceinline__ void tile_stats_row(const float (&cs)[C / 4][4], const float (&cq)[C / 4][4], const int lane, double* row /* [C][2] */) {
  const bool odd = lane & 1, hi = lane & 2;
#pragma unroll
  for (int hf = 0; hf < C / 4; ++hf) {
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
      reinterpret_cast<double2*>(row + (hf * 4 + ch) * 2)[0] = make_double2((double)v1, (double)v2);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// vox_s2 on bf16 tensors (conv_vox_s2_kernel, conv_mfma.hip): stride-2 3x3x3 conv forward / data gradient of the stride-2
// transposed conv, C = 4 / 8.  Same tile (rows de-interleaved by W parity), one 16-byte slot per voxel, bf16 MFMA.
// ------------------------------------------------------------------------------------------------
struct Vs2bArgs {
  const bf16_t* src; int64_t sld; int D, H, W;
  bf16_t* dst; int64_t dld; int oD, oH, oW;
  const bf16_t* wq; const float* bias; int flags;
  double* stats; int rows_per_sample; int tiles; const void* zero_page;
  FastDiv fT, fTw, fTh;
};

template <int C, int TD, int DIL>
__global__ __launch_bounds__(64, 2) void conv_vox_s2b_kernel(Vs2bArgs a) {
  N3D_CHAIN_PRIO();
  constexpr int HF = C / 4;
  constexpr int LD = 2 * (TD - 1) + 2 * DIL + 1, LH = 7 + 2 * DIL, LW = 31 + 2 * DIL;
  constexpr int HW = (LW + 1) / 2, RW = 2 * HW;
  constexpr int PLANE = LH * RW, NPOS = (PLANE + 63) / 64, PSTRIDE = NPOS * 64;
  constexpr int NW4 = (27 * C * C * 2 + 15) / 16, NWI = (NW4 + 63) / 64;
  constexpr int ROT = ((2 * RW * 4) % 64) / 4;
  extern __shared__ __attribute__((aligned(16))) uint4 vlds16[];
  uint4* tile = vlds16;
  uint4* wl = vlds16 + LD * PSTRIDE;
  const int lane = threadIdx.x;
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  uint32_t ub, utile, ubx, uw, ud, uh;
  a.fT.divmod((uint32_t)wg, ub, utile);
  a.fTw.divmod(utile, ubx, uw);
  a.fTh.divmod(ubx, ud, uh);
  const int b = (int)ub, tile_id = (int)utile;
  const int w0 = (int)uw * 16, h0 = (int)uh * 4, d0 = (int)ud * TD;
  const int64_t Ns = (int64_t)a.D * a.H * a.W, Nd = (int64_t)a.oD * a.oH * a.oW;
  const bf16_t* srcb = a.src + (int64_t)b * Ns * a.sld;
  bf16_t* dstb = a.dst + (int64_t)b * Nd * a.dld;
  const int j = lane & 3;
  const int hh = lane >> 4, ww = ((lane & 15) - (hh & 1) * ROT) & 15;
  const bool accum = a.flags & N3D_ACCUMULATE;
  const int64_t vox_off = ((int64_t)(h0 + hh) * a.oW + w0 + ww);
  float4 biasv[HF], prevv[TD][HF];
#pragma unroll
  for (int hf = 0; hf < HF; ++hf) {
    biasv[hf] = a.bias ? *reinterpret_cast<const float4*>(a.bias + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int g = 0; g < TD; ++g)
      prevv[g][hf] = accum ? ld4(dstb + (((int64_t)(d0 + g) * a.oH * a.oW) + vox_off) * a.dld + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const uint4* __restrict__ wq4 = reinterpret_cast<const uint4*>(a.wq);
    const uint4* zp = reinterpret_cast<const uint4*>(a.zero_page);
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int idx = lane + i * 64;
      __builtin_amdgcn_global_load_lds((gptr_t)(idx < NW4 ? wq4 + idx : zp), (lptr_t)(wl + i * 64), 16, 0, 0);
    }
    const int64_t pstride = (int64_t)a.H * a.W * a.sld;
    const int id0 = 2 * d0 - DIL, ih0 = 2 * h0 - DIL, iw0 = 2 * w0 - DIL;
#pragma unroll
    for (int i = 0; i < NPOS; ++i) {
      const int pos = lane + i * 64;
      const int row = pos / RW, rem = pos - row * RW;
      const int par = rem / HW, half = rem - par * HW;
      const int wx = 2 * half + par;
      const int gh = ih0 + row, gw = iw0 + wx;
      const bool okp = pos < PLANE && wx < LW && gh >= 0 && gh < a.H && gw >= 0 && gw < a.W;
      const bf16_t* prow = srcb + ((int64_t)gh * a.W + gw) * a.sld;
#pragma unroll
      for (int dz = 0; dz < LD; ++dz) {
        const int gd = id0 + dz;
        const bool inb = okp && gd >= 0 && gd < a.D;
        const bf16_t* p = prow + gd * pstride;
        __builtin_amdgcn_global_load_lds((gptr_t)(inb ? reinterpret_cast<const void*>(p) : reinterpret_cast<const void*>(zp)),
                                         (lptr_t)(tile + dz * PSTRIDE + i * 64), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 acc[TD][HF];
#pragma unroll
  for (int hf = 0; hf < HF; ++hf) {
    const f32x4 bv = {biasv[hf].x, biasv[hf].y, biasv[hf].z, biasv[hf].w};
#pragma unroll
    for (int g = 0; g < TD; ++g) acc[g][hf] = bv;
  }
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9) {
    const int kh = t9 / 3, kw = t9 % 3;
    W16<C> wr[3];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd) wr[kd] = ldw16<C>(wl, kd * 9 + t9, j);
    const int base = (2 * hh + kh * DIL) * RW + ((kw * DIL) & 1) * HW + ww + ((kw * DIL) >> 1);
    uint4 av[LD];
#pragma unroll
    for (int dz = 0; dz < LD; ++dz) av[dz] = rd_slot<C>(tile, dz * PSTRIDE + base);
#pragma unroll
    for (int dz = 0; dz < LD; ++dz) {
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int g2 = dz - kd * DIL;                 // = 2 * output plane
        if (g2 >= 0 && (g2 & 1) == 0 && (g2 >> 1) < TD) mac16<C>(acc[g2 >> 1], wr[kd], av[dz]);
      }
    }
  }
  float cs[HF][4], cq[HF][4];
#pragma unroll
  for (int hf = 0; hf < HF; ++hf)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[hf][r] = cq[hf][r] = 0.f;
#pragma unroll
  for (int g = 0; g < TD; ++g) {
    bf16_t* o = dstb + (((int64_t)(d0 + g) * a.oH * a.oW) + vox_off) * a.dld;
#pragma unroll
    for (int hf = 0; hf < HF; ++hf) {
      const f32x4 v = acc[g][hf];
#pragma unroll
      for (int r = 0; r < 4; ++r) { cs[hf][r] += v[r]; cq[hf][r] = fmaf(v[r], v[r], cq[hf][r]); }
      const float4 pv = prevv[g][hf];
      st4(o + hf * 4, make_float4(v[0] + pv.x, v[1] + pv.y, v[2] + pv.z, v[3] + pv.w));
    }
  }
  if (a.stats) tile_stats_row<C>(cs, cq, lane, a.stats + ((int64_t)b * a.rows_per_sample + tile_id) * C * 2);
}

// ------------------------------------------------------------------------------------------------
// vox_up on bf16 tensors (conv_vox_up_kernel, conv_mfma.hip): stride-2 transposed 3x3x3 conv forward / data gradient of the
// stride-2 convs, C = 4 / 8: one wave = 4 x 16 source voxels -> all 8 output parity classes.
// ------------------------------------------------------------------------------------------------
struct VupbArgs {
  const bf16_t* src; int64_t sld; int D, H, W;
  bf16_t* dst; int64_t dld;
  const bf16_t* wq; const float* bias; int flags;
  double* stats; int rows_per_sample; int tiles; const void* zero_page;
  FastDiv fT, fTw, fTh;
};
__host__ __device__ constexpr int vupb_count(int dil, int s) { return dil == 2 ? 1 : (s == 0 ? 2 : (s == 1 ? 1 : 0)); }
__host__ __device__ constexpr int vupb_p(int dil, int s, int i) { return dil == 2 ? 0 : (s == 0 ? i : 1); }
__host__ __device__ constexpr int vupb_k(int dil, int s, int i) { return dil == 2 ? 1 - s : (s == 0 ? 1 + i : 0); }

template <int C, int DIL>
__global__ __launch_bounds__(64, 2) void conv_vox_upb_kernel(VupbArgs a) {
  N3D_CHAIN_PRIO();
  constexpr int HF = C / 4;
  constexpr int LD = 3, LH = 6, LW = 18;
  constexpr int PLANE = LH * LW, NPOS = (PLANE + 63) / 64, PSTRIDE = NPOS * 64;
  constexpr int NW4 = (27 * C * C * 2 + 15) / 16, NWI = (NW4 + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) uint4 vlds16[];
  uint4* tile = vlds16;
  uint4* wl = vlds16 + LD * PSTRIDE;
  const int lane = threadIdx.x;
  int wg = blockIdx.x;
  {
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
  const bf16_t* srcb = a.src + (int64_t)b * Ns * a.sld;
  bf16_t* dstb = a.dst + (int64_t)b * 8 * Ns * a.dld;
  const int j = lane & 3;
  const int hh = lane >> 4, ww = ((lane & 15) - (hh & 1) * (LW % 16)) & 15;
  const bool accum = a.flags & N3D_ACCUMULATE;
  const int64_t obase = (((int64_t)(2 * d0) * oH + 2 * (h0 + hh)) * oW + 2 * (w0 + ww)) * a.dld;
  auto ooff = [&](int cls) { return (((int64_t)(cls >> 2) * oH + ((cls >> 1) & 1)) * oW + (cls & 1)) * a.dld; };
  float4 biasv[HF], prevv[8][HF];
#pragma unroll
  for (int hf = 0; hf < HF; ++hf) {
    biasv[hf] = a.bias ? *reinterpret_cast<const float4*>(a.bias + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int cls = 0; cls < 8; ++cls)
      prevv[cls][hf] = accum ? ld4(dstb + obase + ooff(cls) + hf * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const uint4* __restrict__ wq4 = reinterpret_cast<const uint4*>(a.wq);
    const uint4* zp = reinterpret_cast<const uint4*>(a.zero_page);
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
      const bf16_t* prow = srcb + ((int64_t)gh * a.W + gw) * a.sld;
#pragma unroll
      for (int dz = 0; dz < LD; ++dz) {
        const int gd = d0 - 1 + dz;
        const bool inb = okp && gd >= 0 && gd < a.D;
        const bf16_t* p = prow + gd * pstride;
        __builtin_amdgcn_global_load_lds((gptr_t)(inb ? reinterpret_cast<const void*>(p) : reinterpret_cast<const void*>(zp)),
                                         (lptr_t)(tile + dz * PSTRIDE + i * 64), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 acc[8][HF];
#pragma unroll
  for (int hf = 0; hf < HF; ++hf) {
    const f32x4 bv = {biasv[hf].x, biasv[hf].y, biasv[hf].z, biasv[hf].w};
#pragma unroll
    for (int cls = 0; cls < 8; ++cls) acc[cls][hf] = bv;
  }
#pragma unroll
  for (int sd = -1; sd <= 1; ++sd) {
    if (vupb_count(DIL, sd) == 0) continue;
    uint4 av[3][3];
#pragma unroll
    for (int sh = -1; sh <= 1; ++sh)
#pragma unroll
      for (int sw = -1; sw <= 1; ++sw)
        if (vupb_count(DIL, sh) > 0 && vupb_count(DIL, sw) > 0)
          av[sh + 1][sw + 1] = rd_slot<C>(tile, (1 + sd) * PSTRIDE + (hh + 1 + sh) * LW + (ww + 1 + sw));
#pragma unroll
    for (int sh = -1; sh <= 1; ++sh)
#pragma unroll
      for (int sw = -1; sw <= 1; ++sw) {
        if (vupb_count(DIL, sh) == 0 || vupb_count(DIL, sw) == 0) continue;
#pragma unroll
        for (int id = 0; id < vupb_count(DIL, sd); ++id)
#pragma unroll
          for (int ih = 0; ih < vupb_count(DIL, sh); ++ih)
#pragma unroll
            for (int iw = 0; iw < vupb_count(DIL, sw); ++iw) {
              const int cls = vupb_p(DIL, sd, id) * 4 + vupb_p(DIL, sh, ih) * 2 + vupb_p(DIL, sw, iw);
              const int tap = vupb_k(DIL, sd, id) * 9 + vupb_k(DIL, sh, ih) * 3 + vupb_k(DIL, sw, iw);
              const W16<C> wr = ldw16<C>(wl, tap, j);
              mac16<C>(acc[cls], wr, av[sh + 1][sw + 1]);
            }
      }
  }
  float cs[HF][4], cq[HF][4];
#pragma unroll
  for (int hf = 0; hf < HF; ++hf)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[hf][r] = cq[hf][r] = 0.f;
#pragma unroll
  for (int cls = 0; cls < 8; ++cls) {
    bf16_t* o = dstb + obase + ooff(cls);
#pragma unroll
    for (int hf = 0; hf < HF; ++hf) {
      const f32x4 v = acc[cls][hf];
#pragma unroll
      for (int r = 0; r < 4; ++r) { cs[hf][r] += v[r]; cq[hf][r] = fmaf(v[r], v[r], cq[hf][r]); }
      const float4 pv = prevv[cls][hf];
      st4(o + hf * 4, make_float4(v[0] + pv.x, v[1] + pv.y, v[2] + pv.z, v[3] + pv.w));
    }
  }
  if (a.stats) tile_stats_row<C>(cs, cq, lane, a.stats + ((int64_t)b * a.rows_per_sample + tile_id) * C * 2);
}

struct Vs2bPlan { bool ok; int C, td, dil, tiles; size_t lds; };
static Vs2bPlan vs2b_plan(const n3d_conv_geom* g, bool data_grad) {
  Vs2bPlan p; p.ok = false;
  if (data_grad || g->depthwise || g->k != 3 || g->stride != 2 || g->Ci != g->Co || (g->Ci != 4 && g->Ci != 8)) return p;
  if (!(g->dil == 1 || g->dil == 2) || g->pad != g->dil) return p;
  if (g->Wo % 16 != 0 || g->Ho % 4 != 0) return p;
  p.C = g->Ci; p.dil = g->dil;
  p.td = (g->Ci == 4 && g->Do % 2 == 0) ? 2 : 1;
  p.tiles = (g->Wo / 16) * (g->Ho / 4) * (g->Do / p.td);
  const int LD = 2 * (p.td - 1) + 2 * g->dil + 1, LH = 7 + 2 * g->dil, LW = 31 + 2 * g->dil;
  const size_t pstride = ((size_t)LH * 2 * ((LW + 1) / 2) + 63) / 64 * 64;
  const size_t wslots = (((size_t)27 * g->Ci * g->Ci * 2 + 15) / 16 + 63) / 64 * 64;
  p.lds = ((size_t)LD * pstride + wslots) * 16;
  p.ok = p.lds <= 64 * 1024;
  return p;
}
struct VupbPlan { bool ok; int C, dil, tiles; size_t lds; };
static VupbPlan vupb_plan(const n3d_conv_geom* g, bool data_grad) {
  VupbPlan p; p.ok = false;
  if (!data_grad || g->depthwise || g->k != 3 || g->stride != 2 || g->Ci != g->Co || (g->Ci != 4 && g->Ci != 8)) return p;
  if (!(g->dil == 1 || g->dil == 2) || g->pad != g->dil) return p;
  if (g->Di != 2 * g->Do || g->Hi != 2 * g->Ho || g->Wi != 2 * g->Wo) return p;
  if (g->Wo % 16 != 0 || g->Ho % 4 != 0) return p;
  p.C = g->Ci; p.dil = g->dil;
  p.tiles = (g->Wo / 16) * (g->Ho / 4) * g->Do;
  const size_t wslots = (((size_t)27 * g->Ci * g->Ci * 2 + 15) / 16 + 63) / 64 * 64;
  p.lds = ((size_t)3 * 128 + wslots) * 16;
  p.ok = true;
  return p;
}

struct Vx16Plan { bool ok; int C, td, dil, tiles, nw; size_t lds, lds_p2; };

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
  const size_t pstride2 = ((size_t)(4 * p.nw + 2 * g->dil) * 10 + 63) / 64 * 64;     // dense image: 10 slots (20 voxels) per row
  p.lds_p2 = ((size_t)(td + 2 * g->dil) * pstride2 + wslots) * 16;
  return p;
}

// p2: the source is a DENSE 4-channel tensor (pitch 4, 16-byte aligned): the two-voxels-per-slot image
template <int C, int TD, int DIL>
static void launch_vox16_t(const Vx16Args& a, const Vx16Plan& p, int B, hipStream_t s, bool p2) {
  if constexpr (C == 4) {
    if (p2) {
      if constexpr (TD == 4) {
        if (p.nw == 2) { hipLaunchKernelGGL((conv_vox64b_kernel<C, TD, DIL, 2, true>), dim3(p.tiles * B), dim3(128), p.lds_p2, s, a); return; }
      }
      hipLaunchKernelGGL((conv_vox64b_kernel<C, TD, DIL, 1, true>), dim3(p.tiles * B), dim3(64), p.lds_p2, s, a);
      return;
    }
  }
  if constexpr (C == 4 && TD == 4) {
    if (p.nw == 2) { hipLaunchKernelGGL((conv_vox64b_kernel<C, TD, DIL, 2>), dim3(p.tiles * B), dim3(128), p.lds, s, a); return; }
  }
  hipLaunchKernelGGL((conv_vox64b_kernel<C, TD, DIL, 1>), dim3(p.tiles * B), dim3(64), p.lds, s, a);
}

template <int C>
static void launch_vox16_c(const Vx16Args& a, const Vx16Plan& p, int B, hipStream_t s, bool p2) {
  if (p.dil == 1) {
    if (p.td == 4) return launch_vox16_t<C, 4, 1>(a, p, B, s, p2);
    if (p.td == 2) return launch_vox16_t<C, 2, 1>(a, p, B, s, p2);
    return launch_vox16_t<C, 1, 1>(a, p, B, s, p2);
  }
  if (p.td == 4) return launch_vox16_t<C, 4, 2>(a, p, B, s, p2);
  if (p.td == 2) return launch_vox16_t<C, 2, 2>(a, p, B, s, p2);
  return launch_vox16_t<C, 1, 2>(a, p, B, s, p2);
}

// ---- interface to conv_generic.hip -------------------------------------------------------------------------------
// Which bf16 kernel serves (geometry, gather direction)?  0 none, 1 vox64b, 2 vox_s2b, 3 vox_upb.  Only with both tensors in bf16.
static int vox16_kind(const n3d_conv_geom* g, bool data_grad, int flags) {
  if (!((flags & N3D_SRC_BF16) && (flags & N3D_DST_BF16)) || (flags & N3D_NO_MFMA)) return 0;
  if (vs2b_plan(g, data_grad).ok) return 2;
  if (vupb_plan(g, data_grad).ok) return 3;
  if (vx16_plan(g).ok) return 1;
  return 0;
}
// packed-weight layout of n3d_pack_batch: 4 = bf16 [27][cd][cs], data gradient transposed + taps flipped (vox64b, vox_s2b);
// 5 = bf16, data gradient transposed only (vox_upb); 0 = not served here
int vox16_layout(const n3d_conv_geom* g, bool data_grad, int flags) {
  const int k = vox16_kind(g, data_grad, flags);
  return k == 0 ? 0 : (k == 3 ? 5 : 4);
}

int vox16_stats_rows(const n3d_conv_geom* g, bool data_grad, int flags) {
  switch (vox16_kind(g, data_grad, flags)) {
    case 1: { const Vx16Plan v = vx16_plan(g); return v.tiles * v.nw; }
    case 2: return vs2b_plan(g, data_grad).tiles;
    case 3: return vupb_plan(g, data_grad).tiles;
    default: return 0;
  }
}

// 1 = launched, 0 = not applicable, < 0 error
int vox16_conv_try(const n3d_conv_geom* g, bool data_grad, const void* src, int64_t sld, const float* w, const float* bias, void* dst,
                   int64_t dld, int flags, const float* in_gate, const void* relu_src, const float* out_gate, double* stats, void* ws,
                   size_t ws_bytes, hipStream_t s) {
  const int kind = vox16_kind(g, data_grad, flags);
  if (!kind) return 0;
  const int C = g->Ci;
  const bool extras = in_gate || relu_src || out_gate || (flags & N3D_RELU_IN);
  const bool aligned = sld % 4 == 0 && dld % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 7) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0 &&
                       (C == 4 || (sld % 8 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0));
  if (extras || !aligned) {
    if (stats || (flags & N3D_PREPACKED)) { set_error("conv(bf16 mfma): gate / relu extras or an unaligned tensor on a shape whose statistics rows / packed weights assume this kernel"); return N3D_ERR_UNSUPPORTED; }
    return 0;
  }
  const size_t need = (size_t)27 * C * C * 2;
  if (!ws || ws_bytes < need) { set_error("conv(bf16 mfma): workspace too small"); return N3D_ERR_WORKSPACE; }
  // pack modes: vox64b 0 / 1 (data gradient: transposed + flipped), vox_s2b 0 (forward-type gathers only), vox_upb 2 (transposed)
  if (!(flags & N3D_PREPACKED))
    hipLaunchKernelGGL(pack_vox16_kernel, dim3((unsigned)cdiv(27 * C * C, 256)), dim3(256), 0, s, w, (bf16_t*)ws, C,
                       kind == 3 ? 2 : (kind == 1 && data_grad ? 1 : 0));
  const void* zp = zero_page16_ptr();
  if (!zp) { set_error("conv(bf16 mfma): zero page symbol unavailable"); return N3D_ERR_HIP; }
  if (kind == 1) {
    // (the fill addresses a sample through a raw buffer resource: 32-bit byte offsets)
    if ((int64_t)g->Di * g->Hi * g->Wi * sld * 2 >= 0x7fffffffLL) return 0;
    const Vx16Plan v = vx16_plan(g);
    Vx16Args a;
    a.src = (const bf16_t*)src; a.sld = sld; a.dst = (bf16_t*)dst; a.dld = dld; a.wq = (const bf16_t*)ws; a.bias = bias;
    a.D = g->Di; a.H = g->Hi; a.W = g->Wi; a.flags = flags; a.stats = stats; a.rows_per_sample = v.tiles * v.nw; a.tiles = v.tiles; a.zero_page = zp;
    a.fT = FastDiv((uint32_t)v.tiles); a.fTw = FastDiv((uint32_t)(g->Wi / 16)); a.fTh = FastDiv((uint32_t)(g->Hi / (4 * v.nw)));
    const bool p2 = v.C == 4 && sld == 4 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    if (p2 && v.nw == 1 && v.td == 4 && g->Di % 8 == 0 && (int64_t)v.tiles * g->B / 2 >= 2048) {
      const bool acc = flags & N3D_ACCUMULATE;
      // dense image, many tiles: 8 output planes per tile (the D halo is re-fetched every 8 planes instead of every 4: 24.4 -> 22.1 us at
      // (2,4,128^3)); the kernel still writes one statistics row per 4 planes, in the rows and the order of the 4-plane plan
      a.tiles = v.tiles / 2;
      a.fT = FastDiv((uint32_t)a.tiles);
      const size_t pstride2 = ((size_t)(4 + 2 * g->dil) * 10 + 63) / 64 * 64, wslots = (((size_t)27 * 16 * 2 + 15) / 16 + 63) / 64 * 64;
      const size_t lds8 = ((size_t)(8 + 2 * g->dil) * pstride2 + wslots) * 16;
      if (g->dil == 1) {
        if (acc) hipLaunchKernelGGL((conv_vox64b_kernel<4, 8, 1, 1, true, 1>), dim3(a.tiles * g->B), dim3(64), lds8, s, a);
        else hipLaunchKernelGGL((conv_vox64b_kernel<4, 8, 1, 1, true, 0>), dim3(a.tiles * g->B), dim3(64), lds8, s, a);
      } else {
        if (acc) hipLaunchKernelGGL((conv_vox64b_kernel<4, 8, 2, 1, true, 1>), dim3(a.tiles * g->B), dim3(64), lds8, s, a);
        else hipLaunchKernelGGL((conv_vox64b_kernel<4, 8, 2, 1, true, 0>), dim3(a.tiles * g->B), dim3(64), lds8, s, a);
      }
      hipError_t e__ = hipGetLastError();
      if (e__ != hipSuccess) { set_error("conv(bf16 mfma): launch error: %s", hipGetErrorString(e__)); return N3D_ERR_HIP; }
      return 1;
    }
    if (v.C == 4) launch_vox16_c<4>(a, v, g->B, s, p2); else launch_vox16_c<8>(a, v, g->B, s, false);
  } else if (kind == 2) {
    const Vs2bPlan v = vs2b_plan(g, data_grad);
    Vs2bArgs a;
    a.src = (const bf16_t*)src; a.sld = sld; a.D = g->Di; a.H = g->Hi; a.W = g->Wi; a.dst = (bf16_t*)dst; a.dld = dld; a.oD = g->Do; a.oH = g->Ho; a.oW = g->Wo;
    a.wq = (const bf16_t*)ws; a.bias = bias; a.flags = flags; a.stats = stats; a.rows_per_sample = v.tiles; a.tiles = v.tiles; a.zero_page = zp;
    a.fT = FastDiv((uint32_t)v.tiles); a.fTw = FastDiv((uint32_t)(g->Wo / 16)); a.fTh = FastDiv((uint32_t)(g->Ho / 4));
    const dim3 grid(v.tiles * g->B), blk(64);
    if (v.C == 4) {
      if (v.td == 2) { if (v.dil == 1) hipLaunchKernelGGL((conv_vox_s2b_kernel<4, 2, 1>), grid, blk, v.lds, s, a); else hipLaunchKernelGGL((conv_vox_s2b_kernel<4, 2, 2>), grid, blk, v.lds, s, a); }
      else { if (v.dil == 1) hipLaunchKernelGGL((conv_vox_s2b_kernel<4, 1, 1>), grid, blk, v.lds, s, a); else hipLaunchKernelGGL((conv_vox_s2b_kernel<4, 1, 2>), grid, blk, v.lds, s, a); }
    } else {
      if (v.dil == 1) hipLaunchKernelGGL((conv_vox_s2b_kernel<8, 1, 1>), grid, blk, v.lds, s, a); else hipLaunchKernelGGL((conv_vox_s2b_kernel<8, 1, 2>), grid, blk, v.lds, s, a);
    }
  } else {
    const VupbPlan v = vupb_plan(g, data_grad);
    VupbArgs a;
    a.src = (const bf16_t*)src; a.sld = sld; a.D = g->Do; a.H = g->Ho; a.W = g->Wo; a.dst = (bf16_t*)dst; a.dld = dld;
    a.wq = (const bf16_t*)ws; a.bias = bias; a.flags = flags; a.stats = stats; a.rows_per_sample = v.tiles; a.tiles = v.tiles; a.zero_page = zp;
    a.fT = FastDiv((uint32_t)v.tiles); a.fTw = FastDiv((uint32_t)(g->Wo / 16)); a.fTh = FastDiv((uint32_t)(g->Ho / 4));
    const dim3 grid(v.tiles * g->B), blk(64);
    if (v.C == 4) { if (v.dil == 1) hipLaunchKernelGGL((conv_vox_upb_kernel<4, 1>), grid, blk, v.lds, s, a); else hipLaunchKernelGGL((conv_vox_upb_kernel<4, 2>), grid, blk, v.lds, s, a); }
    else { if (v.dil == 1) hipLaunchKernelGGL((conv_vox_upb_kernel<8, 1>), grid, blk, v.lds, s, a); else hipLaunchKernelGGL((conv_vox_upb_kernel<8, 2>), grid, blk, v.lds, s, a); }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("conv(bf16 mfma) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
  return 1;
}

}  // namespace n3d
