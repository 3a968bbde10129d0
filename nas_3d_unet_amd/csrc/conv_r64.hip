// rows64 -- the convolution kernel of the DEEP U-net levels (<= 8^3 voxels per sample, 16 / 32 / 64 channels): forward,
// data gradient and transposed forward of the 3x3x3 convs through one gather map (src = (dst * sn + off + tap * dt) / den).
//
// Why (round 6, profiles/r06_pmc_gemm16_before.json): on these levels the K-split gemm16 kernel gives every 16-row x 16-column
// tile a 1024-thread workgroup that pulls 108 KB of gathered voxels AND 108 KB of weights through ONE compute unit's load path
// (~60 GB/s from L2): 3.6 us of a 7.3 us launch, on 32 of the 256 compute units.  Here a workgroup owns 64 ROWS x 4 (8) COLUMNS:
//   * v_mfma_f32_4x4x1_16b_f32 with the WEIGHTS as A operand (lane & 3 = output channel of the quad, identical in all 16 blocks) and
//     the 64 voxels of the tile as B operand (lane = row): a lane ends with the four output channels of its own voxel -- one 16-byte
//     store, and a weight traffic of K x 16 bytes per workgroup (27 KB at 64 channels) instead of K x 64;
//   * the source voxels the tile can reach (whole D planes of one sample, or whole samples on the 2^3 level) are staged ONCE in LDS
//     by LDS-DMA (16-64 KB; channel quads XOR-swizzled by the voxel index so that 64 lanes reading one quad of 64 different voxels
//     hit 64 banks), the workgroup's weight columns beside them; taps outside the volume read one zero slot;
//   * K (taps x channel quads) is split over the 16 waves; partial tiles meet in LDS and wave 0 adds them in wave order (fixed
//     order: bit-reproducible), applies the epilogue (bias, ReLU mask, per-(b,c) gate, accumulate) and writes the GroupNorm
//     partial row of the tile.
// Global bytes per workgroup: 43-90 KB instead of 216 KB; one launch takes up to four independent convs (the two convs of a
// searched-cell node, the plain convs of a supernet node).
#include "n3d_common.h"
#include "conv_r64.h"

namespace n3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// weights: Wq[tap][cd >> 2][cs >> 2][cd & 3][cs & 3] <- native (Co, Ci, taps); transpose = 1 (data gradient / transposed forward):
// cs = co, cd = ci, taps as they are (the gather map carries the direction)
__global__ void pack_r64_kernel(const float* __restrict__ w, float* __restrict__ wq, int Co, int Ci, int taps, int transpose) {
  const int Cs = transpose ? Co : Ci, Cd = transpose ? Ci : Co;
  const int E = Cs * Cd;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= taps * E) return;
  const int tap = i / E, r = i - tap * E;
  const int e = r & 3, j = (r >> 2) & 3, rest = r >> 4, Q = Cs >> 2;
  const int q = rest % Q, cdq = rest / Q;
  const int cs = q * 4 + e, cd = cdq * 4 + j;
  const int co = transpose ? cs : cd, ci = transpose ? cd : cs;
  wq[i] = w[((int64_t)co * Ci + ci) * taps + tap];
}

// floor(x / den) for den in {1, 2} and any sign of x
__device__ __host__ __forceinline__ int fdiv_den(int x, int den) { return den == 2 ? (x >> 1) : x; }

template <int CT>
__device__ __forceinline__ void r64_body(const R64Args& a, const int wg, float4* const lds) {
  N3D_CHAIN_PRIO();
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 3;
  const int Q = a.Cs >> 2;                                    // channel quads of a source voxel: 4, 8 or 16
  const int lq = Q == 16 ? 4 : (Q == 8 ? 3 : 2);
  const int swz_sh = 4 - lq;                                  // quad' = quad ^ ((voxel >> swz_sh) & (Q - 1)): see the header
  const int taps = a.k * a.k * a.k;
  const int wslots = taps * Q * 4 * CT;                       // float4 slots of the workgroup's weight columns
  float4* const xs = lds;
  const int ZS = a.xslots;                                    // the zero slot
  float4* const wl = lds + a.xslots + 64;
  float4* const gl = wl + ((wslots + 63) & ~63);              // [samples of the tile][Q] input gates
  f32x4* const red = reinterpret_cast<f32x4*>(gl + a.gslots);  // [15][CT][64]

  uint32_t urt, uct;
  a.fnct.divmod((uint32_t)wg, urt, uct);
  const int rt = (int)urt, ct = (int)uct;
  const int Mtot = a.B * a.Nd;
  const int row0 = rt * 64;
  // ---- the lane's destination voxel
  const int i = row0 + lane;
  const bool valid = i < Mtot;
  uint32_t ub, uv, ud, ur, uh, uw;
  a.fNd.divmod((uint32_t)(valid ? i : row0), ub, uv);
  a.fHWd.divmod(uv, ud, ur);
  a.fWd.divmod(ur, uh, uw);
  const int b = (int)ub, dd = (int)ud, dh = (int)uh, dw = (int)uw;
  // ---- the source voxels this tile can reach: planes [p_lo, p_hi] of ONE sample (tiles of >= one plane, or parts of one), or whole
  // samples (2^3 level: several samples per tile).  Uniform over the workgroup.
  int v_lo, nvox, b0;
  if (a.Nd >= 64) {
    uint32_t tb, tv;
    a.fNd.divmod((uint32_t)row0, tb, tv);
    b0 = (int)tb;
    const int d_lo = (int)a.fHWd.div(tv), d_hi = (int)a.fHWd.div(tv + 63);
    const int reach = (a.k - 1) * a.dt;
    int p_lo = fdiv_den(d_lo * a.sn + a.off + (reach < 0 ? reach : 0), a.den);
    int p_hi = fdiv_den(d_hi * a.sn + a.off + (reach > 0 ? reach : 0), a.den);
    if (p_lo < 0) p_lo = 0;
    if (p_hi > a.Ds - 1) p_hi = a.Ds - 1;
    v_lo = (b0 * a.Ds + p_lo) * a.HWs;
    nvox = p_hi >= p_lo ? (p_hi - p_lo + 1) * a.HWs : 0;
  } else {
    b0 = (int)a.fNd.div((uint32_t)row0);
    int b1 = (int)a.fNd.div((uint32_t)(row0 + 63));
    if (b1 > a.B - 1) b1 = a.B - 1;
    v_lo = b0 * a.Ns;
    nvox = (b1 - b0 + 1) * a.Ns;
  }
  v_lo = __builtin_amdgcn_readfirstlane(v_lo); nvox = __builtin_amdgcn_readfirstlane(nvox); b0 = __builtin_amdgcn_readfirstlane(b0);

  // ---- epilogue operands of the writer wave: requested ahead of everything else, so that the epilogue never waits on memory
  const bool accum = a.flags & N3D_ACCUMULATE;
  float4 e_bias[CT], e_relu[CT], e_gate[CT], e_prev[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    e_bias[c] = make_float4(0.f, 0.f, 0.f, 0.f); e_relu[c] = make_float4(1.f, 1.f, 1.f, 1.f);
    e_gate[c] = make_float4(1.f, 1.f, 1.f, 1.f); e_prev[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int n0 = (ct * CT + c) * 4;
      if (a.bias) e_bias[c] = ld4(a.bias + n0);
      if (valid) {
        if (a.relu_src) e_relu[c] = ld4(a.relu_src + (int64_t)i * a.rld + n0);
        if (a.out_gate) e_gate[c] = ld4(a.out_gate + (int64_t)b * a.Cd + n0);
        if (accum) e_prev[c] = ld4(a.dst + (int64_t)i * a.dld + n0);
      }
    }
  }

  // ---- fill: source voxels (swizzled quads), weight columns, the zero slot, the input gates
  {
    const int F = nvox << lq;
    const float* sb = a.src + (int64_t)v_lo * a.sld;
    for (int s0 = wave * 64; s0 < F; s0 += 1024) {
      const int s = s0 + lane;
      const int u = s >> lq, x = s & (Q - 1);
      const int q = x ^ ((u >> swz_sh) & (Q - 1));
      if (s < F) __builtin_amdgcn_global_load_lds((gptr_t)(sb + (int64_t)u * a.sld + q * 4), (lptr_t)(xs + s0), 16, 0, 0);
    }
    const int per_tap = Q * 4 * CT, lpt = lq + 2 + (CT == 2 ? 1 : 0);
    const float4* wq4 = reinterpret_cast<const float4*>(a.wq) + (int64_t)ct * per_tap;
    const int E4 = (a.Cs * a.Cd) >> 2;
    for (int s0 = wave * 64; s0 < wslots; s0 += 1024) {
      const int s = s0 + lane;
      const int tap = s >> lpt, r = s & (per_tap - 1);
      if (s < wslots) __builtin_amdgcn_global_load_lds((gptr_t)(wq4 + (int64_t)tap * E4 + r), (lptr_t)(wl + s0), 16, 0, 0);
    }
    if (threadIdx.x == 0) xs[ZS] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.in_gate) {
      const int nsmp = a.Nd >= 64 ? 1 : 64 / a.Nd;
      const int t = threadIdx.x;
      if (t < nsmp * Q) {
        const int sb_ = b0 + (t >> lq);
        gl[t] = sb_ < a.B ? ld4(a.in_gate + (int64_t)sb_ * a.Cs + (t & (Q - 1)) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- K loop: this wave's run of (tap, quad) pairs
  const int NP = taps << lq;
  const int p0 = (NP * wave) >> 4, p1 = (NP * (wave + 1)) >> 4;
  f32x4 acc[CT][2];
#pragma unroll
  for (int c = 0; c < CT; ++c) { acc[c][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[c][1] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const float relu_floor = (a.flags & N3D_RELU_IN) ? 0.f : -INFINITY;
  const bool den2 = a.den == 2;
  const int bl = a.Nd >= 64 ? 0 : (int)((uint32_t)lane / (uint32_t)a.Nd);     // the lane's sample inside the tile (gate table row)
  const int gbase = bl << lq;
  const int per_tap = Q * 4 * CT;
  int cur_tap = -1, ubase = 0, uswz = 0;
  bool ok = false;
  for (int p = p0; p < p1; ++p) {
    const int tap = p >> lq, q = p & (Q - 1);
    if (tap != cur_tap) {
      cur_tap = tap;
      const int kd = a.k == 3 ? tap / 9 : 0, kh = a.k == 3 ? (tap % 9) / 3 : 0, kw = a.k == 3 ? tap % 3 : 0;
      int nd = dd * a.sn + a.off + kd * a.dt, nh = dh * a.sn + a.off + kh * a.dt, nw = dw * a.sn + a.off + kw * a.dt;
      ok = valid;
      if (den2) { ok = ok & (((nd | nh | nw) & 1) == 0); nd >>= 1; nh >>= 1; nw >>= 1; }
      ok = ok & ((unsigned)nd < (unsigned)a.Ds) & ((unsigned)nh < (unsigned)a.Hs) & ((unsigned)nw < (unsigned)a.Ws);
      const int u = ((b * a.Ds + nd) * a.Hs + nh) * a.Ws + nw - v_lo;
      ok = ok & ((unsigned)u < (unsigned)nvox);
      ubase = u << lq;
      uswz = (u >> swz_sh) & (Q - 1);
    }
    const int slot = ok ? (ubase + (q ^ uswz)) : ZS;
    float4 x4 = xs[slot];
    x4.x = fmaxf(x4.x, relu_floor); x4.y = fmaxf(x4.y, relu_floor); x4.z = fmaxf(x4.z, relu_floor); x4.w = fmaxf(x4.w, relu_floor);
    if (a.in_gate) { const float4 g4 = gl[gbase + q]; x4.x *= g4.x; x4.y *= g4.y; x4.z *= g4.z; x4.w *= g4.w; }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const float4 w4 = wl[tap * per_tap + ((c << lq) + q) * 4 + j];
      acc[c][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.x, x4.x, acc[c][0], 0, 0, 0);
      acc[c][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.y, x4.y, acc[c][1], 0, 0, 0);
      acc[c][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.z, x4.z, acc[c][0], 0, 0, 0);
      acc[c][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4.w, x4.w, acc[c][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c][0] += acc[c][1];

  // ---- the K slices meet in LDS; wave 0 adds them in wave order
  if (wave > 0) {
#pragma unroll
    for (int c = 0; c < CT; ++c) red[((wave - 1) * CT + c) * 64 + lane] = acc[c][0];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < 15; ++w)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c][0] += red[(w * CT + c) * 64 + lane];

  // ---- epilogue: the lane holds the CT x 4 output channels of its own voxel
  float vs[CT][4];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int n0 = (ct * CT + c) * 4;
    float4 v = make_float4(acc[c][0][0] + e_bias[c].x, acc[c][0][1] + e_bias[c].y, acc[c][0][2] + e_bias[c].z, acc[c][0][3] + e_bias[c].w);
    if (!(e_relu[c].x > 0.f)) v.x = 0.f;
    if (!(e_relu[c].y > 0.f)) v.y = 0.f;
    if (!(e_relu[c].z > 0.f)) v.z = 0.f;
    if (!(e_relu[c].w > 0.f)) v.w = 0.f;
    v.x = v.x * e_gate[c].x + e_prev[c].x; v.y = v.y * e_gate[c].y + e_prev[c].y;
    v.z = v.z * e_gate[c].z + e_prev[c].z; v.w = v.w * e_gate[c].w + e_prev[c].w;
    if (valid) st4(a.dst + (int64_t)i * a.dld + n0, v);
    vs[c][0] = valid ? v.x : 0.f; vs[c][1] = valid ? v.y : 0.f; vs[c][2] = valid ? v.z : 0.f; vs[c][3] = valid ? v.w : 0.f;
  }
  if (a.stats) {
    if (a.Nd >= 64) {
      // all 64 rows belong to sample b0: one partial row per tile
      const int tis = rt - b0 * (a.Nd >> 6);
      const int sel = classsum4_sel(lane);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float s = wave_classsum4_f<1>(vs[c][0], vs[c][1], vs[c][2], vs[c][3]);
        const float q2 = wave_classsum4_f<1>(vs[c][0] * vs[c][0], vs[c][1] * vs[c][1], vs[c][2] * vs[c][2], vs[c][3] * vs[c][3]);
        if ((lane & 15) == 0) {
          double* o = a.stats + (((int64_t)b0 * a.rows_per_sample + tis) * a.Cd + (ct * CT + c) * 4 + sel) * 2;
          reinterpret_cast<double2*>(o)[0] = make_double2((double)s, (double)q2);
        }
      }
    } else {
      // several samples per tile (Nd a power of two < 64): sums over each run of Nd lanes, one row per sample
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = vs[c][r], q2 = vs[c][r] * vs[c][r];
          for (int o = 1; o < a.Nd; o <<= 1) { s += __shfl_xor(s, o, 64); q2 += __shfl_xor(q2, o, 64); }
          if (valid && (lane & (a.Nd - 1)) == 0) {
            double* o = a.stats + (((int64_t)b * a.rows_per_sample) * a.Cd + (ct * CT + c) * 4 + r) * 2;
            reinterpret_cast<double2*>(o)[0] = make_double2((double)s, (double)q2);
          }
        }
    }
  }
}

struct R64Multi { R64Args a[4]; int start[5]; };

template <int CT>
__global__ __launch_bounds__(1024, 4) void conv_r64_kernel(R64Multi q) {
  extern __shared__ __attribute__((aligned(16))) float4 r64_lds[];
  const int L = blockIdx.x;
  const int k = (L >= q.start[1]) + (L >= q.start[2]) + (L >= q.start[3]);
  // (static indexing: a run-time index into the kernel arguments would copy the whole table to scratch)
  switch (k) {
    case 0: r64_body<CT>(q.a[0], L - q.start[0], r64_lds); break;
    case 1: r64_body<CT>(q.a[1], L - q.start[1], r64_lds); break;
    case 2: r64_body<CT>(q.a[2], L - q.start[2], r64_lds); break;
    default: r64_body<CT>(q.a[3], L - q.start[3], r64_lds); break;
  }
}

// ------------------------------------------------------------------------------------------------ host side
#ifndef N3D_R64_MAX_ROWS
#define N3D_R64_MAX_ROWS 1024     // rows (batch x voxels) up to which a conv goes here: the <= 8^3 levels at batch 2
#endif
static int r64_max_rows() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("N3D_R64_MAX_ROWS"); v = e ? atoi(e) : N3D_R64_MAX_ROWS; }
  return v;
}

R64Plan r64_plan(const n3d_conv_geom* g, bool data_grad) {
  R64Plan p; p.ok = false; p.ct = 1; p.ntile = 0; p.xslots = 0; p.gslots = 64; p.lds = 0;
  if (g->depthwise || g->k != 3) return p;
  const int Cs = data_grad ? g->Co : g->Ci, Cd = data_grad ? g->Ci : g->Co;
  if (!(Cs == 16 || Cs == 32 || Cs == 64) || Cd % 16 != 0 || Cd > 64) return p;
  if (g->pad > 4 || g->dil > 2 || g->dil < 1 || (g->stride != 1 && g->stride != 2)) return p;
  const int Dd = data_grad ? g->Di : g->Do, Hd = data_grad ? g->Hi : g->Ho, Wd = data_grad ? g->Wi : g->Wo;
  const int Ds = data_grad ? g->Do : g->Di, Hs = data_grad ? g->Ho : g->Hi, Ws = data_grad ? g->Wo : g->Wi;
  const int64_t Nd = (int64_t)Dd * Hd * Wd, Ns = (int64_t)Ds * Hs * Ws;
  const int64_t M = (int64_t)g->B * Nd;
  if (M < 1 || M > r64_max_rows()) return p;
  if (Nd >= 64 ? (Nd % 64 != 0) : ((Nd & (Nd - 1)) != 0)) return p;
  // tiles of >= 64 rows inside one plane or made of whole planes
  const int HWd = Hd * Wd;
  if (Nd >= 64 && !(HWd % 64 == 0 || 64 % HWd == 0)) return p;
  const int sn = data_grad ? 1 : g->stride, den = data_grad ? g->stride : 1;
  int64_t maxvox, nsmp = 1;
  if (Nd >= 64) {
    const int pt = HWd >= 64 ? 1 : 64 / HWd;                          // destination planes of a tile
    int64_t planes = ((int64_t)(pt - 1) * sn + 2 * g->dil) / den + 2;   // source planes it reaches (conservative)
    if (planes > Ds) planes = Ds;
    maxvox = planes * Hs * Ws;
  } else {
    int64_t ns = 64 / Nd;
    if (ns > g->B) ns = g->B;
    maxvox = ns * Ns;
    nsmp = 64 / Nd;
  }
  const int Q = Cs / 4;
  p.ntile = (int)cdiv(M, 64);
  p.ct = ((int64_t)p.ntile * (Cd / 4) > 192 && Cd % 8 == 0) ? 2 : 1;
  p.xslots = (int)((maxvox * Q + 63) / 64 * 64);
  p.gslots = (int)((nsmp * Q + 63) / 64 * 64);
  const size_t wslots = ((size_t)27 * Q * 4 * p.ct + 63) / 64 * 64;
  p.lds = ((size_t)p.xslots + 64 + wslots + p.gslots + (size_t)15 * p.ct * 64) * 16;
  if (p.lds > 150 * 1024) return p;
  p.ok = true;
  return p;
}

int r64_stats_rows(const n3d_conv_geom* g, bool data_grad) {
  const int64_t Nd = data_grad ? (int64_t)g->Di * g->Hi * g->Wi : (int64_t)g->Do * g->Ho * g->Wo;
  return Nd >= 64 ? (int)(Nd / 64) : 1;
}

// fills the arguments for one conv (packing the weights unless pre-packed); 1 = ready, 0 = shape not served, < 0 error
int r64_prepare(const n3d_conv_geom* g, bool data_grad, const float* src, int64_t sld, const float* w, const float* bias, float* dst,
                int64_t dld, int flags, const float* in_gate, const float* relu_src, int64_t rld, const float* out_gate, double* stats,
                void* ws, size_t ws_bytes, hipStream_t s, R64Args* out, R64Plan* plan) {
  if (flags & (N3D_NO_MFMA | N3D_SRC_BF16 | N3D_DST_BF16)) return 0;
  const R64Plan p = r64_plan(g, data_grad);
  if (!p.ok) return 0;
  if (sld % 4 != 0 || dld % 4 != 0 || !aligned16(src) || !aligned16(dst) || (relu_src && (rld % 4 != 0 || !aligned16(relu_src))) ||
      (bias && !aligned16(bias)) || (in_gate && !aligned16(in_gate)) || (out_gate && !aligned16(out_gate))) {
    if (stats || (flags & N3D_PREPACKED)) { set_error("conv(rows64): misaligned operand with statistics or pre-packed weights"); return N3D_ERR_UNSUPPORTED; }
    return 0;
  }
  R64Args a;
  a.src = src; a.sld = sld; a.dst = dst; a.dld = dld; a.bias = bias; a.k = g->k; a.flags = flags; a.B = g->B;
  a.in_gate = in_gate; a.relu_src = relu_src; a.rld = rld; a.out_gate = out_gate; a.stats = stats;
  if (!data_grad) { a.Ds = g->Di; a.Hs = g->Hi; a.Ws = g->Wi; a.Cs = g->Ci; a.Dd = g->Do; a.Hd = g->Ho; a.Wd = g->Wo; a.Cd = g->Co;
    a.sn = g->stride; a.off = -g->pad; a.dt = g->dil; a.den = 1; }
  else { a.Ds = g->Do; a.Hs = g->Ho; a.Ws = g->Wo; a.Cs = g->Co; a.Dd = g->Di; a.Hd = g->Hi; a.Wd = g->Wi; a.Cd = g->Ci;
    a.sn = 1; a.off = g->pad; a.dt = -g->dil; a.den = g->stride; }
  a.Nd = a.Dd * a.Hd * a.Wd; a.Ns = a.Ds * a.Hs * a.Ws; a.HWs = a.Hs * a.Ws; a.HWd = a.Hd * a.Wd;
  if ((int64_t)g->B * a.Ns * sld * 4 >= (1ll << 31) || (int64_t)g->B * a.Nd * dld * 4 >= (1ll << 31)) return 0;
  a.nct = a.Cd / (4 * p.ct);
  a.xslots = p.xslots; a.gslots = p.gslots;
  a.rows_per_sample = stats ? r64_stats_rows(g, data_grad) : 0;
  a.fNd = FastDiv((uint32_t)a.Nd); a.fWd = FastDiv((uint32_t)a.Wd); a.fHd = FastDiv((uint32_t)a.Hd);
  a.fnct = FastDiv((uint32_t)a.nct); a.fHWd = FastDiv((uint32_t)a.HWd);
  const int taps = g->k * g->k * g->k;
  const size_t need = (size_t)taps * a.Cs * a.Cd * 4;
  if (!ws || ws_bytes < need) { set_error("conv(rows64): workspace too small (%zu < %zu)", ws_bytes, need); return N3D_ERR_WORKSPACE; }
  a.wq = (const float*)ws;
  if (!(flags & N3D_PREPACKED))
    hipLaunchKernelGGL(pack_r64_kernel, dim3((unsigned)cdiv((int64_t)taps * a.Cs * a.Cd, 256)), dim3(256), 0, s, w, (float*)ws, g->Co, g->Ci, taps,
                       data_grad ? 1 : 0);
  *out = a; *plan = p;
  return 1;
}

// n = 1 .. 4 prepared convs of one column-tile width in one launch
int r64_launch(int n, const R64Args* as, const R64Plan* ps, hipStream_t s) {
  R64Multi q;
  int total = 0;
  size_t lds = 0;
  for (int i = 0; i < 4; ++i) {
    if (i < n) {
      q.a[i] = as[i];
      q.start[i] = total;
      total += ps[i].ntile * as[i].nct;
      if (ps[i].lds > lds) lds = ps[i].lds;
      if (ps[i].ct != ps[0].ct) { set_error("conv(rows64): mixed column-tile widths in one launch"); return N3D_ERR_INVALID; }
    } else {
      q.a[i] = as[0]; q.start[i] = 0x7fffffff;
    }
  }
  q.start[4] = total;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_r64_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_r64_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (ps[0].ct == 2) hipLaunchKernelGGL(conv_r64_kernel<2>, dim3((unsigned)total), dim3(1024), lds, s, q);
  else hipLaunchKernelGGL(conv_r64_kernel<1>, dim3((unsigned)total), dim3(1024), lds, s, q);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("conv(rows64) launch: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
  return 1;
}

}  // namespace n3d
