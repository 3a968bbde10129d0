// Generic (any k/stride/dilation/channel count) convolution family on the VALU.
// One "gather" kernel serves Conv3d forward, its data gradient, ConvTranspose3d forward and its
// data gradient: dst[v][cd] = bias + sum_tap sum_cs src[map(v,tap)][cs] * W[tap][cs][cd].
// Weights are wave-uniform -> the compiler fetches them with scalar loads (s_load_dwordx4..16)
// from the packed [tap][cs][cd] copy; activations are 16-byte vector loads of NDHWC voxels.
// These kernels cover every shape of the path; the MFMA kernels in conv_mfma.hip take over the
// FLOP-heavy 3x3x3 shapes.
#include "n3d_common.h"
#define N3D_WG16_SLABS 1024   // 256-float partial slabs reserved for the gemm16 weight-gradient kernels (+ one per tile)
#include <algorithm>
#include <vector>

namespace n3d {

struct GatherArgs {
  const void* src; int64_t sld; int Ds, Hs, Ws, Cs;   // TS elements (fp32 or bf16 storage)
  void* dst; int64_t dld; int Dd, Hd, Wd, Cd;         // TD elements
  const float* wp; int Cdp;
  const float* bias;
  int k, sn, off, dt, den;
  int flags;
  const float* in_gate;
  const void* relu_src; int64_t rld;                  // TD elements (the tensor whose gradient dst is)
  const float* out_gate;
  double* stats;
  FastDiv fWd, fHd;
  int cls;            // 1: parity-class mode (den == 2, even output dims): workgroup = one output parity class
  FastDiv fWh, fHh;   // Wd / 2, Hd / 2
};

// pack native (Co, Ci, k^3) weights into wp[tap][cs][cdp]; transpose=0: cs=ci, cd=co (forward);
// transpose=1: cs=co, cd=ci (data gradient / transposed forward)
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Co, int Ci, int taps, int Cdp, int transpose) {
  const int Cs = transpose ? Co : Ci, Cd = transpose ? Ci : Co;
  const int total = taps * Cs * Cdp;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cd = i % Cdp, cs = (i / Cdp) % Cs, tap = i / (Cdp * Cs);
  float v = 0.f;
  if (cd < Cd) {
    const int co = transpose ? cs : cd, ci = transpose ? cd : cs;
    v = w[((int64_t)co * Ci + ci) * taps + tap];
  }
  wp[i] = v;
}

template <int CO_T, int SV, int CSQ, typename TS = float, typename TD = float>
__global__ __launch_bounds__(256) void conv_gather_kernel(GatherArgs a) {
  N3D_CHAIN_PRIO();
  __shared__ double red[4][CO_T * 2];
  const int b = blockIdx.z, cot = blockIdx.y;
  const int64_t Nd = (int64_t)a.Dd * a.Hd * a.Wd, Ns = (int64_t)a.Ds * a.Hs * a.Ws;
  // den == 2 (transposed conv forward / strided conv data gradient): an output voxel only receives the taps whose
  // source index (o + off + k*dt) is even.  In class mode a workgroup serves ONE output parity class, so the valid
  // tap set is uniform and the other taps (19 of 27 on average, 26 of 27 for dilation 2 off-lattice classes) are
  // skipped instead of being loaded and masked.
  int64_t v;
  bool valid;
  int w_, h_, d_;
  int kmd = (1 << a.k) - 1, kmh = kmd, kmw = kmd;  // valid-tap bit masks per dimension (uniform); 1x1x1 convs: tap 0 only
  if (a.cls) {
    const int cls = blockIdx.x & 7;
    const int pd = cls >> 2, ph = (cls >> 1) & 1, pw = cls & 1;
    const int64_t jv = (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    valid = jv < (Nd >> 3);
    uint32_t qw, rw_, qd, rh_;
    a.fWh.divmod(valid ? (uint32_t)jv : 0u, qw, rw_);
    a.fHh.divmod(qw, qd, rh_);
    w_ = 2 * (int)rw_ + pw; h_ = 2 * (int)rh_ + ph; d_ = 2 * (int)qd + pd;
    v = ((int64_t)d_ * a.Hd + h_) * a.Wd + w_;
    kmd = kmh = kmw = 0;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
      kmd |= (((pd + a.off + kk * a.dt) & 1) == 0) << kk;
      kmh |= (((ph + a.off + kk * a.dt) & 1) == 0) << kk;
      kmw |= (((pw + a.off + kk * a.dt) & 1) == 0) << kk;
    }
  } else {
    v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    valid = v < Nd;
    const uint32_t vv = valid ? (uint32_t)v : 0u;
    uint32_t qw, rw_, qd, rh_;
    a.fWd.divmod(vv, qw, rw_);
    a.fHd.divmod(qw, qd, rh_);
    w_ = (int)rw_; h_ = (int)rh_; d_ = (int)qd;
  }
  float acc[CO_T];
#pragma unroll
  for (int j = 0; j < CO_T; ++j) {
    const int c = cot * CO_T + j;
    acc[j] = (a.bias && c < a.Cd) ? a.bias[c] : 0.f;
  }
  const TS* srcb = reinterpret_cast<const TS*>(a.src) + (int64_t)b * Ns * a.sld;
  const float* gate = a.in_gate ? a.in_gate + (int64_t)b * a.Cs : nullptr;
  const bool relu_in = a.flags & N3D_RELU_IN;
  const int k = a.k;
  // Weights: every lane of the block needs the same [tap][cs][CO_T] tile.  Scalar (SMEM) loads serialise on one
  // memory latency per tap (measured: ~0.7 us per tap, 22 us per k3 conv whatever its size), so the small-Cs path
  // stages the tile once in LDS (<= 27*12*16 floats) and reads it back with broadcast ds_read_b128.
  extern __shared__ __attribute__((aligned(16))) float wlds[];
  if (SV == 4 && CSQ > 0) {
    // eight weight elements per thread requested before the first LDS store (a rolled load -> store loop pays one
    // memory latency per trip: 7 - 21 trips for a 3x3x3 tile)
    const int ntile = k * k * k * CSQ * 4 * CO_T;
    for (int i0 = threadIdx.x; i0 < ntile; i0 += 256 * 8) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        const int ic = i < ntile ? i : 0;
        const int co = ic % CO_T, r = ic / CO_T;  // r = tap*Cs + cs
        wv[u] = a.wp[(int64_t)r * a.Cdp + cot * CO_T + co];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        if (i < ntile) wlds[i] = wv[u];
      }
    }
    __syncthreads();
  }
  if (SV == 4 && CSQ > 0) {
    // Few input channels (Cs = 4*CSQ <= 12): the loads of a whole (kh, kw) plane of taps (CSQ <= 2) or of one kw row
    // (CSQ == 3) are issued back to back from clamped addresses, then consumed -- no branch between a load and the
    // next one, so a thread pays one memory latency per plane instead of one per tap.
    constexpr int RB = (CSQ <= 2) ? 3 : 1;   // kh rows per batch
    const float relu_floor = relu_in ? 0.f : -INFINITY;
    float gv[CSQ > 0 ? CSQ * 4 : 1];
#pragma unroll
    for (int e = 0; e < CSQ * 4; ++e) gv[e] = gate ? gate[e] : 1.f;
    for (int kd = 0; kd < k; ++kd) {
      if (!((kmd >> kd) & 1)) continue;  // uniform: no tap of this plane reaches this parity class
      int nd = d_ * a.sn + a.off + kd * a.dt;
      bool okd = valid;
      if (a.den == 2) { okd = okd && !(nd & 1); nd >>= 1; }
      okd = okd && nd >= 0 && nd < a.Ds;
      const int cd_ = min(max(nd, 0), a.Ds - 1);
      for (int kh0 = 0; kh0 < k; kh0 += RB) {
        float4 xq[RB][3][CSQ > 0 ? CSQ : 1];
        bool okv[RB][3];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int kh = kh0 + r;
          if (!((kmh >> kh) & 1)) continue;  // uniform
          int nh = h_ * a.sn + a.off + kh * a.dt;
          bool okh = okd && kh < k;
          if (a.den == 2) { okh = okh && !(nh & 1); nh >>= 1; }
          okh = okh && nh >= 0 && nh < a.Hs;
          const int ch_ = min(max(nh, 0), a.Hs - 1);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            if (!((kmw >> kw) & 1)) continue;  // uniform
            int nw = w_ * a.sn + a.off + kw * a.dt;
            bool ok = okh && kw < k;
            if (a.den == 2) { ok = ok && !(nw & 1); nw >>= 1; }
            ok = ok && nw >= 0 && nw < a.Ws;
            okv[r][kw] = ok;
            const int cw_ = min(max(nw, 0), a.Ws - 1);
            const TS* xs = srcb + (((int64_t)cd_ * a.Hs + ch_) * a.Ws + cw_) * a.sld;
#pragma unroll
            for (int q = 0; q < CSQ; ++q) xq[r][kw][q] = ld4(xs + q * 4);
          }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const int kh = kh0 + r;
          if (kh >= k || !((kmh >> kh) & 1)) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            if (kw >= k || !((kmw >> kw) & 1)) continue;
            const int tap = (kd * k + kh) * k + kw;
            const float* wr = wlds + tap * (CSQ * 4) * CO_T;
            const bool ok = okv[r][kw];
            // branch-free operand preparation (a uniform branch here would fence every LDS weight read behind a wait)
            float xe[CSQ > 0 ? CSQ * 4 : 1];
#pragma unroll
            for (int q = 0; q < CSQ; ++q) {
              const float xv[4] = {xq[r][kw][q].x, xq[r][kw][q].y, xq[r][kw][q].z, xq[r][kw][q].w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float x = ok ? xv[j] : 0.f;
                x = fmaxf(x, relu_floor);
                xe[q * 4 + j] = x * gv[q * 4 + j];
              }
            }
#pragma unroll
            for (int e = 0; e < CSQ * 4; ++e) {
              const float4* wrow = reinterpret_cast<const float4*>(wr + e * CO_T);
#pragma unroll
              for (int c4 = 0; c4 < CO_T / 4; ++c4) {
                const float4 wv = wrow[c4];
                acc[c4 * 4 + 0] = fmaf(xe[e], wv.x, acc[c4 * 4 + 0]);
                acc[c4 * 4 + 1] = fmaf(xe[e], wv.y, acc[c4 * 4 + 1]);
                acc[c4 * 4 + 2] = fmaf(xe[e], wv.z, acc[c4 * 4 + 2]);
                acc[c4 * 4 + 3] = fmaf(xe[e], wv.w, acc[c4 * 4 + 3]);
              }
            }
          }
        }
      }
    }
  } else
  for (int kd = 0; kd < k; ++kd) {
    int nd = d_ * a.sn + a.off + kd * a.dt;
    bool okd = true;
    if (a.den == 2) { okd = !(nd & 1); nd >>= 1; }
    okd = okd && nd >= 0 && nd < a.Ds;
    for (int kh = 0; kh < k; ++kh) {
      int nh = h_ * a.sn + a.off + kh * a.dt;
      bool okh = okd;
      if (a.den == 2) { okh = okh && !(nh & 1); nh >>= 1; }
      okh = okh && nh >= 0 && nh < a.Hs;
      for (int kw = 0; kw < k; ++kw) {
        int nw = w_ * a.sn + a.off + kw * a.dt;
        bool ok = okh && valid;
        if (a.den == 2) { ok = ok && !(nw & 1); nw >>= 1; }
        ok = ok && nw >= 0 && nw < a.Ws;
        const int tap = (kd * k + kh) * k + kw;
        const float* wr = a.wp + (int64_t)tap * a.Cs * a.Cdp + cot * CO_T;
        if (ok) {
          const TS* xs = srcb + (((int64_t)nd * a.Hs + nh) * a.Ws + nw) * a.sld;
          if (SV == 4) {
            // four 16-byte loads in flight per step
            for (int c16 = 0; c16 < a.Cs; c16 += 16) {
              float4 q4[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) q4[u] = ld4(xs + min(c16 + u * 4, a.Cs - 4));
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int c4 = c16 + u * 4;
                if (c4 >= a.Cs) break;
                const float xv[4] = {q4[u].x, q4[u].y, q4[u].z, q4[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  float x = xv[j];
                  if (relu_in) x = fmaxf(x, 0.f);
                  if (gate) x *= gate[c4 + j];
                  const float* wrow = wr + (int64_t)(c4 + j) * a.Cdp;
#pragma unroll
                  for (int co = 0; co < CO_T; ++co) acc[co] = fmaf(x, wrow[co], acc[co]);
                }
              }
            }
          } else {
            for (int c = 0; c < a.Cs; ++c) {
              float x = ld1(xs + c);
              if (relu_in) x = fmaxf(x, 0.f);
              if (gate) x *= gate[c];
              const float* wrow = wr + (int64_t)c * a.Cdp;
#pragma unroll
              for (int co = 0; co < CO_T; ++co) acc[co] = fmaf(x, wrow[co], acc[co]);
            }
          }
        }
      }
    }
  }
  // ---- epilogue
  const int c0 = cot * CO_T;
  if (valid) {
    if (a.relu_src) {
      const TD* rs = reinterpret_cast<const TD*>(a.relu_src) + ((int64_t)b * Nd + v) * a.rld + c0;
#pragma unroll
      for (int j = 0; j < CO_T; ++j)
        if (c0 + j < a.Cd && !(ld1(rs + j) > 0.f)) acc[j] = 0.f;
    }
    if (a.out_gate) {
#pragma unroll
      for (int j = 0; j < CO_T; ++j)
        if (c0 + j < a.Cd) acc[j] *= a.out_gate[(int64_t)b * a.Cd + c0 + j];
    }
    TD* o = reinterpret_cast<TD*>(a.dst) + ((int64_t)b * Nd + v) * a.dld + c0;
    const bool accum = a.flags & N3D_ACCUMULATE;
    if (CO_T % 4 == 0 && (a.Cd % 4 == 0) && (a.dld % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.dst) & (4 * sizeof(TD) - 1)) == 0)) {
#pragma unroll
      for (int j = 0; j < CO_T; j += 4) {
        float4 r = make_float4(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
        if (accum) { const float4 p = ld4(o + j); r.x += p.x; r.y += p.y; r.z += p.z; r.w += p.w; acc[j] = r.x; acc[j + 1] = r.y; acc[j + 2] = r.z; acc[j + 3] = r.w; }
        st4(o + j, r);
      }
    } else {
#pragma unroll
      for (int j = 0; j < CO_T; ++j)
        if (c0 + j < a.Cd) { if (accum) acc[j] += ld1(o + j); st1(o + j, acc[j]); }
    }
  }
  if (a.stats) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // fp32 tree over the 64 lanes of a wave (as the MFMA kernels do), four channels at a time (wave_classsum4_f: row r of the wave
    // ends up with channel {0, 2, 1, 3}[r] of the quad); waves and rows are added in fp64
    static_assert(CO_T % 4 == 0, "conv_gather_kernel: CO_T must be a multiple of 4");
    const int sel = classsum4_sel(lane);
#pragma unroll
    for (int j = 0; j < CO_T; j += 4) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = valid ? acc[j + e] : 0.f;
      const float s = wave_classsum4_f<1>(v[0], v[1], v[2], v[3]);
      const float ss = wave_classsum4_f<1>(v[0] * v[0], v[1] * v[1], v[2] * v[2], v[3] * v[3]);
      if ((lane & 15) == 0) { red[wave][(j + sel) * 2] = (double)s; red[wave][(j + sel) * 2 + 1] = (double)ss; }
    }
    __syncthreads();
    if (threadIdx.x < CO_T * 2) {
      const int j = threadIdx.x >> 1, kk = threadIdx.x & 1;
      if (c0 + j < a.Cd) {
        const double s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        a.stats[(((int64_t)b * gridDim.x + blockIdx.x) * a.Cd + c0 + j) * 2 + kk] = s;
      }
    }
  }
}

static int pick_cot(int Cd) {
  if (Cd % 16 == 0) return 16;
  if (Cd % 12 == 0) return 12;
  if (Cd % 8 == 0) return 8;
  return 4;
}

// parity-class mode of the gather kernel: den == 2, 3x3x3, even output dims, small-Cs vector path
static bool gather_class_mode(int den, int k, int Dd, int Hd, int Wd, int Cs, bool vec) {
  return den == 2 && k == 3 && Dd % 2 == 0 && Hd % 2 == 0 && Wd % 2 == 0 && vec && (Cs == 4 || Cs == 8 || Cs == 12);
}

template <int CO_T, typename TS = float, typename TD = float>
static void launch_gather_t(GatherArgs a, int B, hipStream_t s) {
  const int64_t Nd = (int64_t)a.Dd * a.Hd * a.Wd;
  const bool vec = (a.Cs % 4 == 0) && (a.sld % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.src) & (4 * sizeof(TS) - 1)) == 0);
  a.cls = gather_class_mode(a.den, a.k, a.Dd, a.Hd, a.Wd, a.Cs, vec) ? 1 : 0;
  a.fWh = FastDiv((uint32_t)(a.Wd / 2 > 0 ? a.Wd / 2 : 1)); a.fHh = FastDiv((uint32_t)(a.Hd / 2 > 0 ? a.Hd / 2 : 1));
  dim3 grid((unsigned)(a.cls ? 8 * cdiv(Nd / 8, 256) : cdiv(Nd, 256)), (unsigned)(a.Cdp / CO_T), (unsigned)B);
  const size_t wbytes = (size_t)a.k * a.k * a.k * a.Cs * CO_T * sizeof(float);
  if (!vec) hipLaunchKernelGGL((conv_gather_kernel<CO_T, 1, 0, TS, TD>), grid, dim3(256), 0, s, a);
  else if (a.Cs == 4) hipLaunchKernelGGL((conv_gather_kernel<CO_T, 4, 1, TS, TD>), grid, dim3(256), wbytes, s, a);
  else if (a.Cs == 8) hipLaunchKernelGGL((conv_gather_kernel<CO_T, 4, 2, TS, TD>), grid, dim3(256), wbytes, s, a);
  else if (a.Cs == 12) hipLaunchKernelGGL((conv_gather_kernel<CO_T, 4, 3, TS, TD>), grid, dim3(256), wbytes, s, a);
  else hipLaunchKernelGGL((conv_gather_kernel<CO_T, 4, 0, TS, TD>), grid, dim3(256), 0, s, a);
}

template <typename TS, typename TD>
static void launch_gather_d(const GatherArgs& a, int B, hipStream_t s) {
  switch (pick_cot(a.Cd)) {
    case 16: launch_gather_t<16, TS, TD>(a, B, s); break;
    case 12: launch_gather_t<12, TS, TD>(a, B, s); break;
    case 8: launch_gather_t<8, TS, TD>(a, B, s); break;
    default: launch_gather_t<4, TS, TD>(a, B, s); break;
  }
}

// storage types of the source / destination tensors: flags N3D_SRC_BF16 / N3D_DST_BF16
static void launch_gather(const GatherArgs& a, int B, hipStream_t s) {
  const bool sb = a.flags & N3D_SRC_BF16, db = a.flags & N3D_DST_BF16;
  if (sb && db) launch_gather_d<bf16_t, bf16_t>(a, B, s);
  else if (sb) launch_gather_d<bf16_t, float>(a, B, s);
  else if (db) launch_gather_d<float, bf16_t>(a, B, s);
  else launch_gather_d<float, float>(a, B, s);
}

// ------------------------------------------------------------------------------------------------
// 1x1x1 stride-1 convs with few channels on the large levels (stems, the preprocess convs of the outer cells, their data
// gradients): pure streaming.  One thread owns four voxels (coalesced: voxel = base + i*256 + lane), the [Cs][Cd] weight
// tile sits in LDS and every broadcast read of it feeds four voxels, a workgroup writes ONE statistics row for its 1024
// voxels.  The generic gather kernel spends a workgroup (weight staging, barrier, block reduction) on every 256 voxels.
// ------------------------------------------------------------------------------------------------
struct K1Args {
  const void* src; int64_t sld; void* dst; int64_t dld; const float* wp; int Cdp; const float* bias;   // src: TS elements, dst / relu_src: TD
  int flags; const void* relu_src; int64_t rld; double* stats; int64_t N;
  // up: the data gradient of a stride-2 1x1x1 conv -- destination voxel (d,h,w) takes source voxel (d/2,h/2,w/2) when all three
  // are even and nothing otherwise; Ns = source voxels per sample
  int up, Wd, Hd; int64_t Ns; FastDiv fWd, fHd;
  // sparse (up + accumulate): only the destination voxels with three even coordinates receive anything, the rest keep their value --
  // the launch walks the Ns SOURCE voxels and read-modify-writes their destinations instead of rewriting the whole tensor
  // (12 -> 8 at 2 x 128^3 -> 64^3: 400 MB of traffic for 50 MB of work)
  int sparse; FastDiv fWs, fHs;
  // "weight_norm" recompute form (n3d_conv_k1_norm_fwd): nostore = statistics only, nothing is written; oscale / oshift [B][Cd] = the
  // GroupNorm coefficients applied to the conv result before it is stored (out = oscale * (W x + bias) + oshift)
  int nostore; const float* oscale; const float* oshift;
  // flat: the destination is dense (dld == Cd) with >= 2 quads per voxel and 16-byte aligned 64-voxel runs: a wave transposes its
  // 64 voxel records through LDS and writes them as consecutive 16-byte pieces (per-voxel stores of a 12-channel tensor put 16 bytes
  // on every 48-byte pitch: three partial-line instructions per line)
  int flat;
  // node-planar operands (round 5; include/n3d.h, "node-planar tensors": a pitch SMALLER than the channel count = the channels come as
  // C / pitch dense node tensors of `pitch` channels each, node k at base + k * node_stride elements).  Source (forward): sld < Cs, quad q
  // is quad q % (sld/4) of node q / (sld/4).  Destination (data gradient; also its ReLU mask source and previous value): one node per
  // blockIdx.z -- the kernel's Cd is the NODE's channel count, its weight columns start at blockIdx.z * Cd
  int64_t src_node_stride, dst_node_stride, relu_node_stride;
};
constexpr int K1_VPB = 1024;   // voxels per workgroup

// EXTRA: the data-gradient extras (accumulate into dst, ReLU mask source) are in use -- they cost 8 * CDQ registers per voxel
// storage-form quad (what one load instruction returns): the data-gradient form keeps its ReLU mask source and the destination's
// previous value this way until they are used -- 2 registers per bf16 quad instead of 4
template <typename T> struct K1Raw;
template <> struct K1Raw<float> { typedef float4 type; };
template <> struct K1Raw<bf16_t> { typedef uint2 type; };
__device__ __forceinline__ float4 k1_ldraw(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint2 k1_ldraw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 k1_f4(const float4 v) { return v; }
__device__ __forceinline__ float4 k1_f4(const uint2 v) {
  return make_float4(__builtin_bit_cast(float, v.x << 16), __builtin_bit_cast(float, v.x & 0xffff0000u),
                     __builtin_bit_cast(float, v.y << 16), __builtin_bit_cast(float, v.y & 0xffff0000u));
}
// Minimum waves per SIMD the compiler must leave room for.  The data-gradient form is a pure streaming pass (56 bytes per voxel at
// 4 -> 12 channels, bf16) and lives on the bytes it keeps in flight: with the forward form's extras compiled in (statistics, output
// coefficients) and 256 VGPRs allowed it took 204 of them -- 2 waves per SIMD, 3.6 TB/s at 2 x 128^3; without them and held to 128:
// 112 VGPRs, 4 waves, 6.0 TB/s (profiles/r05_k1_ab.log).  Held to 80 it spills (0.2 of peak), so the wider shapes keep 2.
// The forward form of the 4 -> 12 conv (the stem's two passes: 213 VGPRs / 2 waves when left alone) is held to 170: 150-158 used, 3 waves,
// 57 -> 45 us at 2 x 128^3 fp32 (0.59 -> 0.74 of 8 TB/s), bf16 output 45 -> 38; the other forward shapes spill under that cap and keep 2.
constexpr int k1_min_waves(int csq, int cdq, bool extra) { return (extra && csq * cdq <= 3) ? 4 : ((!extra && csq == 1 && cdq == 3) ? 3 : 2); }
template <int CSQ, int CDQ, bool EXTRA, typename TS = float, typename TD = float>   // Cs = 4 * CSQ, Cd = 4 * CDQ
__global__ __launch_bounds__(256, k1_min_waves(CSQ, CDQ, EXTRA)) void conv_k1_kernel(K1Args a) {
  N3D_CHAIN_PRIO();
  constexpr int VPT = EXTRA ? 2 : K1_VPB / 256;   // data-gradient form: no statistics rows to agree on, fewer registers per voxel
  __shared__ __attribute__((aligned(16))) float4 wl[CSQ * 4 * CDQ];
  __shared__ double red[4][CDQ * 8];
  constexpr bool FLAT_OK = CDQ >= 2 && CDQ <= 3;
  __shared__ __attribute__((aligned(16))) TD stage[FLAT_OK ? 4 * 64 * CDQ * 4 : 4];   // [wave][voxel][Cd]
  const int t = threadIdx.x, b = blockIdx.y, z = blockIdx.z;     // z: destination node (node-planar data gradient), else 0
  for (int i = t; i < CSQ * 4 * CDQ; i += 256) {
    const int cs = i / CDQ, q = i - cs * CDQ;
    wl[i] = *reinterpret_cast<const float4*>(a.wp + (int64_t)cs * a.Cdp + (z * CDQ + q) * 4);
  }
  const int64_t base = (int64_t)blockIdx.x * (VPT * 256) + t;
  const TS* sb = reinterpret_cast<const TS*>(a.src) + (int64_t)b * (EXTRA && a.up ? a.Ns : a.N) * a.sld;
  // per-quad source bases (uniform): a node-planar source has sld < Cs
  const TS* sq[CSQ];
  {
    const int nq = (int)(a.sld >> 2);           // quads per node (planar) -- or per voxel record
#pragma unroll
    for (int q = 0; q < CSQ; ++q) {
      const int node = (nq > 0 && nq < CSQ) ? q / nq : 0;
      sq[q] = sb + node * a.src_node_stride + (q - node * (nq < CSQ ? nq : 0)) * 4;
    }
  }
  TD* db = reinterpret_cast<TD*>(a.dst) + z * a.dst_node_stride + (int64_t)b * a.N * a.dld;
  const TD* rb = a.relu_src ? reinterpret_cast<const TD*>(a.relu_src) + z * a.relu_node_stride + (int64_t)b * a.N * a.rld : nullptr;
  const bool accum = EXTRA && (a.flags & N3D_ACCUMULATE);
  const float floor_ = (a.flags & N3D_RELU_IN) ? 0.f : -INFINITY;
  typedef typename K1Raw<TD>::type raw_t;
  float4 x[VPT][CSQ];
  raw_t prev[EXTRA ? VPT : 1][CDQ], msk[EXTRA ? VPT : 1][CDQ];
  bool ok[VPT];
  int64_t vdst[VPT];      // destination voxel of slot i
  const bool sparse = EXTRA && a.sparse;
  const int64_t nwalk = sparse ? a.Ns : a.N;
  // every global operand is requested before the first use
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int64_t v = base + i * 256;
    ok[i] = v < nwalk;
    int64_t vc = ok[i] ? v : nwalk - 1;
    int64_t vs = vc;
    bool hit = true;
    if constexpr (EXTRA) {
      if (sparse) {
        uint32_t q1, uw, ud, uh;
        a.fWs.divmod((uint32_t)vc, q1, uw);
        a.fHs.divmod(q1, ud, uh);
        vc = ((int64_t)(2 * ud) * a.Hd + 2 * uh) * a.Wd + 2 * uw;
      } else if (a.up) {
        uint32_t q1, uw, ud, uh;
        a.fWd.divmod((uint32_t)vc, q1, uw);
        a.fHd.divmod(q1, ud, uh);
        hit = !((uw | uh | ud) & 1u);
        vs = ((int64_t)(ud >> 1) * (a.Hd >> 1) + (uh >> 1)) * (a.Wd >> 1) + (uw >> 1);
        if (!hit) vs = 0;
      }
    }
    vdst[i] = vc;
#pragma unroll
    for (int q = 0; q < CSQ; ++q) {
      x[i][q] = ld4(sq[q] + vs * a.sld);
      if (EXTRA && !hit) x[i][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (EXTRA) {
#pragma unroll
      for (int q = 0; q < CDQ; ++q) {
        if (accum) prev[i][q] = k1_ldraw(db + vc * a.dld + q * 4);
        if (rb) msk[i][q] = k1_ldraw(rb + vc * a.rld + q * 4);
      }
    }
  }
  float4 acc[VPT][CDQ], osc[CDQ], osh[CDQ];
#pragma unroll
  for (int q = 0; q < CDQ; ++q) {
    const float4 bq = a.bias ? *reinterpret_cast<const float4*>(a.bias + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < VPT; ++i) acc[i][q] = bq;
    if constexpr (!EXTRA) {     // (the data-gradient form carries neither output coefficients nor statistics: run_gather refuses them)
      osc[q] = a.oscale ? *reinterpret_cast<const float4*>(a.oscale + (int64_t)b * (CDQ * 4) + q * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
      osh[q] = a.oshift ? *reinterpret_cast<const float4*>(a.oshift + (int64_t)b * (CDQ * 4) + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
#pragma unroll
  for (int cs = 0; cs < CSQ * 4; ++cs) {
    float xv[VPT];
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const float4 xq = x[i][cs >> 2];
      xv[i] = fmaxf((cs & 3) == 0 ? xq.x : ((cs & 3) == 1 ? xq.y : ((cs & 3) == 2 ? xq.z : xq.w)), floor_);
    }
#pragma unroll
    for (int q = 0; q < CDQ; ++q) {
      const float4 w = wl[cs * CDQ + q];
#pragma unroll
      for (int i = 0; i < VPT; ++i) {
        acc[i][q].x = fmaf(xv[i], w.x, acc[i][q].x); acc[i][q].y = fmaf(xv[i], w.y, acc[i][q].y);
        acc[i][q].z = fmaf(xv[i], w.z, acc[i][q].z); acc[i][q].w = fmaf(xv[i], w.w, acc[i][q].w);
      }
    }
  }
  float s1[CDQ * 4], s2[CDQ * 4];
#pragma unroll
  for (int j = 0; j < CDQ * 4; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
#pragma unroll
    for (int q = 0; q < CDQ; ++q) {
      float4 r = acc[i][q];
      if constexpr (EXTRA) {
        if (rb) {
          const float4 mk = k1_f4(msk[i][q]);
          r.x = mk.x > 0.f ? r.x : 0.f; r.y = mk.y > 0.f ? r.y : 0.f; r.z = mk.z > 0.f ? r.z : 0.f; r.w = mk.w > 0.f ? r.w : 0.f;
        }
        if (accum) {
          const float4 pv = k1_f4(prev[i][q]);
          r.x += pv.x; r.y += pv.y; r.z += pv.z; r.w += pv.w;
        }
      }
      if (ok[i]) {
        if constexpr (!EXTRA) {
          if (a.oscale) {      // (the statistics of this pass, if any, are those of the stored values)
            r.x = fmaf(osc[q].x, r.x, osh[q].x); r.y = fmaf(osc[q].y, r.y, osh[q].y); r.z = fmaf(osc[q].z, r.z, osh[q].z); r.w = fmaf(osc[q].w, r.w, osh[q].w);
          }
        }
        if (!EXTRA && a.nostore) {}
        else if (FLAT_OK && a.flat) st4(stage + ((t >> 6) * 64 + (t & 63)) * (CDQ * 4) + q * 4, r);
        else st4(db + vdst[i] * a.dld + q * 4, r);
        if constexpr (!EXTRA) {
          s1[q * 4] += r.x; s1[q * 4 + 1] += r.y; s1[q * 4 + 2] += r.z; s1[q * 4 + 3] += r.w;
          s2[q * 4] = fmaf(r.x, r.x, s2[q * 4]); s2[q * 4 + 1] = fmaf(r.y, r.y, s2[q * 4 + 1]);
          s2[q * 4 + 2] = fmaf(r.z, r.z, s2[q * 4 + 2]); s2[q * 4 + 3] = fmaf(r.w, r.w, s2[q * 4 + 3]);
        }
      }
    }
    if constexpr (FLAT_OK) {
      if (a.flat && !(!EXTRA && a.nostore)) {
        // the wave's 64 voxels (base + i*256 + lane) are one contiguous run of the dense destination
        const int wave = t >> 6, lane = t & 63;
        const int64_t v0 = (int64_t)blockIdx.x * (VPT * 256) + i * 256 + wave * 64;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (v0 < a.N) {
          const int64_t nvox = a.N - v0 < 64 ? a.N - v0 : 64;
          const int npieces = (int)(nvox * (CDQ * 4) * (int)sizeof(TD) / 16);        // whole 16-byte pieces of the valid run
          const uint4* sp = reinterpret_cast<const uint4*>(stage + wave * 64 * (CDQ * 4));
          uint4* dp = reinterpret_cast<uint4*>(db + v0 * (CDQ * 4));
          constexpr int NPC = 64 * CDQ * 4 * (int)sizeof(TD) / 16;
#pragma unroll
          for (int pc = 0; pc < (NPC + 63) / 64; ++pc) {
            const int k = pc * 64 + lane;
            if (k < npieces) dp[k] = sp[k];
          }
          // a run that ends in the middle of a piece (bf16, odd number of quads, ragged tail): the last voxels one by one
          const int64_t done = (int64_t)npieces * 16 / ((CDQ * 4) * (int)sizeof(TD));
          if (lane >= done && lane < nvox) {
#pragma unroll
            for (int q = 0; q < CDQ; ++q) st4(db + (v0 + lane) * (CDQ * 4) + q * 4, ld4(stage + (wave * 64 + lane) * (CDQ * 4) + q * 4));
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  if (!EXTRA && a.stats) {
    const int wave = t >> 6, lane = t & 63;
    // (four channels per wave sum, wave_classsum4_f: row r of the wave ends up with channel {0, 2, 1, 3}[r] of the quad)
    const int sel = classsum4_sel(lane);
#pragma unroll
    for (int q = 0; q < CDQ; ++q) {
      const float s = wave_classsum4_f<1>(s1[q * 4], s1[q * 4 + 1], s1[q * 4 + 2], s1[q * 4 + 3]);
      const float ss = wave_classsum4_f<1>(s2[q * 4], s2[q * 4 + 1], s2[q * 4 + 2], s2[q * 4 + 3]);
      if ((lane & 15) == 0) { red[wave][(q * 4 + sel) * 2] = (double)s; red[wave][(q * 4 + sel) * 2 + 1] = (double)ss; }
    }
    __syncthreads();
    if (t < CDQ * 8)
      a.stats[(((int64_t)b * gridDim.x + blockIdx.x) * (CDQ * 4)) * 2 + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
  }
}

// shapes the 1x1x1 streaming kernel takes (the statistics row count depends on it: n3d_conv_stats_rows)
static bool k1_dims_ok(const n3d_conv_geom* g, bool data_grad, int Cs, int Cd);
static bool k1_shape_ok(const n3d_conv_geom* g, bool data_grad) {
  return k1_dims_ok(g, data_grad, data_grad ? g->Co : g->Ci, data_grad ? g->Ci : g->Co);
}
// Cs / Cd: source / destination channels ONE launch handles (a node-planar destination is written node by node: Cd = the node's)
static bool k1_dims_ok(const n3d_conv_geom* g, bool data_grad, int Cs, int Cd) {
  if (g->k != 1 || g->depthwise) return false;
  // stride 2: only the data gradient (zero-upsampling form), even input dims, no bias
  if (g->stride != 1 && !(g->stride == 2 && data_grad && g->pad == 0 && g->Di % 2 == 0 && g->Hi % 2 == 0 && g->Wi % 2 == 0)) return false;
  const int64_t N = (int64_t)g->Di * g->Hi * g->Wi;
  // register budget: (Cs/4) * (Cd/4) <= 6 covers the nets' shapes (4->12, 12->4, 12->8, 24->4 and their data gradients).  (A 24-channel
  // destination -- the data gradient of the last up cell's 24 -> 4 preprocess conv -- was tried in round 4: with two voxels per thread the
  // kernel spills, +0.47 ms per 128^3 step; with one it is 0.02-0.04 ms slower than the gather kernel it replaces: profiles/r04_128_ab.log)
  // volume: >= 32768 voxels (N3D_K1_MIN_N).  Round 4 tried 4096 -- the supernet's pointwise convs at the 16^3 level take 11.7 us per data
  // gradient on the gather kernel against 5.6 us for the 32^3 level here -- and the search step did not move (12.88 vs 12.87 ms: those
  // launches are off its chain), so the threshold stays where every launch form of a supernet node shares its kernels bit for bit
  constexpr int64_t min_n = 32768;
  return Cs % 4 == 0 && Cs >= 4 && Cs <= 24 && Cd % 4 == 0 && Cd >= 4 && Cd <= 12 && (Cs / 4) * (Cd / 4) <= 6 && N >= min_n;
}

template <int CSQ, bool EXTRA, typename TS, typename TD>
static void launch_k1_e(const K1Args& a, int Cd, dim3 grid, hipStream_t s) {
  switch (Cd / 4) {
    case 1: hipLaunchKernelGGL((conv_k1_kernel<CSQ, 1, EXTRA, TS, TD>), grid, dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((conv_k1_kernel<CSQ, 2, EXTRA, TS, TD>), grid, dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((conv_k1_kernel<CSQ, 3, EXTRA, TS, TD>), grid, dim3(256), 0, s, a); break;
  }
}
template <int CSQ, typename TS, typename TD>
static void launch_k1_d(const K1Args& a, int Cd, dim3 grid, hipStream_t s) {
  if ((a.flags & N3D_ACCUMULATE) || a.relu_src || a.up) launch_k1_e<CSQ, true, TS, TD>(a, Cd, grid, s);
  else launch_k1_e<CSQ, false, TS, TD>(a, Cd, grid, s);
}
template <int CSQ>
static void launch_k1_c(const K1Args& a, int Cd, dim3 grid, hipStream_t s) {
  const bool sb = a.flags & N3D_SRC_BF16, db = a.flags & N3D_DST_BF16;
  if (sb && db) launch_k1_d<CSQ, bf16_t, bf16_t>(a, Cd, grid, s);
  else if (sb) launch_k1_d<CSQ, bf16_t, float>(a, Cd, grid, s);
  else if (db) launch_k1_d<CSQ, float, bf16_t>(a, Cd, grid, s);
  else launch_k1_d<CSQ, float, float>(a, Cd, grid, s);
}

// weight gradient of the same 1x1x1 convs (stride 1 or 2): dW[co][ci] = sum_v f(x[i(v)][ci]) * dy[v][co], dbias = sum_v dy[v].
// A thread streams output voxels (coalesced over the workgroup), keeps the whole Ci x Co outer product in registers and the
// workgroup leaves ONE partial slab ([Ci][Co] + [Co]) for the common fixed-order finalize.  The tiled generic kernel reads dy
// once per input-channel tile and x in 16-byte pieces of 48-byte voxels.
struct K1WgArgs {
  const void* x; int64_t xld; const void* dy; int64_t dyld; float* partial; float* pbias; int64_t total, chunk; int flags;   // x: TS elements, dy: TD elements
  int stride, Wo, Ho, Wi, Hi; int64_t No, Ni; FastDiv fNo, fWo, fHo;
  int64_t x_node_stride;     // node-planar x (xld < Ci: see K1Args): elements between two nodes
};
constexpr int K1W_CHUNK = 2048;   // output voxels per workgroup

template <int CIQ, int COQ, typename TS = float, typename TD = float>
__global__ __launch_bounds__(256) void conv_k1_wgrad_kernel(K1WgArgs a) {
  constexpr int CI = CIQ * 4, CO = COQ * 4, NV = CI * CO + CO;
  __shared__ float red[4][NV];
  const int t = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * a.chunk;
  int64_t i1 = i0 + a.chunk;
  if (i1 > a.total) i1 = a.total;
  const float floor_ = (a.flags & N3D_RELU_IN) ? 0.f : -INFINITY;
  float acc[CI][CO], accb[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) accb[c] = 0.f;
#pragma unroll
  for (int ci = 0; ci < CI; ++ci)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[ci][c] = 0.f;
  // per-quad bases of x (uniform): a node-planar x has xld < CI
  const TS* xqb[CIQ];
  {
    const int nq = (int)(a.xld >> 2);
#pragma unroll
    for (int q = 0; q < CIQ; ++q) {
      const int node = (nq > 0 && nq < CIQ) ? q / nq : 0;
      xqb[q] = reinterpret_cast<const TS*>(a.x) + node * a.x_node_stride + (q - node * (nq < CIQ ? nq : 0)) * 4;
    }
  }
  for (int64_t i = i0 + t; i < i1; i += 512) {
    // two voxels in flight per trip
    float4 xq[2][CIQ], gq[2][COQ];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t v = i + u * 256;
      ok[u] = v < i1;
      const int64_t vc = ok[u] ? v : i;
      int64_t xi = vc;
      if (a.stride == 2) {
        uint32_t ub, uo, q1, uw, ud, uh;
        a.fNo.divmod((uint32_t)vc, ub, uo);
        a.fWo.divmod(uo, q1, uw);
        a.fHo.divmod(q1, ud, uh);
        xi = (int64_t)ub * a.Ni + ((int64_t)(2 * ud) * a.Hi + 2 * uh) * a.Wi + 2 * uw;
      }
#pragma unroll
      for (int q = 0; q < CIQ; ++q) xq[u][q] = ld4(xqb[q] + xi * a.xld);
#pragma unroll
      for (int q = 0; q < COQ; ++q) gq[u][q] = ld4(reinterpret_cast<const TD*>(a.dy) + vc * a.dyld + q * 4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!ok[u]) continue;
      float g[CO];
#pragma unroll
      for (int q = 0; q < COQ; ++q) { g[q * 4] = gq[u][q].x; g[q * 4 + 1] = gq[u][q].y; g[q * 4 + 2] = gq[u][q].z; g[q * 4 + 3] = gq[u][q].w; }
#pragma unroll
      for (int c = 0; c < CO; ++c) accb[c] += g[c];
#pragma unroll
      for (int q = 0; q < CIQ; ++q) {
        const float xv[4] = {fmaxf(xq[u][q].x, floor_), fmaxf(xq[u][q].y, floor_), fmaxf(xq[u][q].z, floor_), fmaxf(xq[u][q].w, floor_)};
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < CO; ++c) acc[q * 4 + e][c] = fmaf(xv[e], g[c], acc[q * 4 + e][c]);
      }
    }
  }
  const int wave = t >> 6, lane = t & 63;
  static_assert(CO % 4 == 0, "k1 weight gradient: CO must be a multiple of 4");
  const int sel = classsum4_sel(lane);       // (four accumulators per wave sum: see conv_wgrad_kernel)
  const bool wr = (lane & 15) == 0;
#pragma unroll
  for (int ci = 0; ci < CI; ++ci)
#pragma unroll
    for (int c = 0; c < CO; c += 4) {
      const float sv = wave_classsum4_f<1>(acc[ci][c], acc[ci][c + 1], acc[ci][c + 2], acc[ci][c + 3]);
      if (wr) red[wave][ci * CO + c + sel] = sv;
    }
#pragma unroll
  for (int c = 0; c < CO; c += 4) {
    const float sv = wave_classsum4_f<1>(accb[c], accb[c + 1], accb[c + 2], accb[c + 3]);
    if (wr) red[wave][CI * CO + c + sel] = sv;
  }
  __syncthreads();
  for (int j = t; j < NV; j += 256) {
    const float sv = red[0][j] + red[1][j] + red[2][j] + red[3][j];
    if (j < CI * CO) a.partial[(int64_t)blockIdx.x * (CI * CO) + j] = sv;
    else a.pbias[(int64_t)blockIdx.x * CO + (j - CI * CO)] = sv;
  }
}

static bool k1_wgrad_shape_ok(const n3d_conv_geom* g) {
  if (g->k != 1 || g->depthwise || g->pad != 0 || (g->stride != 1 && g->stride != 2)) return false;
  const int64_t total = (int64_t)g->B * g->Do * g->Ho * g->Wo;
  return g->Ci % 4 == 0 && g->Ci >= 4 && g->Ci <= 24 && g->Co % 4 == 0 && g->Co >= 4 && g->Co <= 12 && (g->Ci / 4) * (g->Co / 4) <= 6 &&
         total >= 32768 && total < (1ll << 31);
}

template <int CIQ, typename TS, typename TD>
static void launch_k1_wgrad_d(const K1WgArgs& a, int Co, int nchunks, hipStream_t s) {
  switch (Co / 4) {
    case 1: hipLaunchKernelGGL((conv_k1_wgrad_kernel<CIQ, 1, TS, TD>), dim3(nchunks), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL((conv_k1_wgrad_kernel<CIQ, 2, TS, TD>), dim3(nchunks), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL((conv_k1_wgrad_kernel<CIQ, 3, TS, TD>), dim3(nchunks), dim3(256), 0, s, a); break;
  }
}
template <int CIQ>
static void launch_k1_wgrad_c(const K1WgArgs& a, int Co, int nchunks, hipStream_t s) {
  const bool sb = a.flags & N3D_SRC_BF16, db = a.flags & N3D_DST_BF16;
  if (sb && db) launch_k1_wgrad_d<CIQ, bf16_t, bf16_t>(a, Co, nchunks, s);
  else if (sb) launch_k1_wgrad_d<CIQ, bf16_t, float>(a, Co, nchunks, s);
  else if (db) launch_k1_wgrad_d<CIQ, float, bf16_t>(a, Co, nchunks, s);
  else launch_k1_wgrad_d<CIQ, float, float>(a, Co, nchunks, s);
}

// ------------------------------------------------------------------------------------------------
// "weight_norm" 1x1x1 conv with few channels on a large volume (stem0: 4 -> 12 at 64^3 / 128^3; nas.py:28, searched.py:69;
// prim_ops.py:68-83), backward WITHOUT the raw conv output: raw = W x + bias costs Ci FMAs per channel to recompute from the
// Ci-channel input, while storing it and reading it back (forward epilogue, both backward passes) and writing d(raw) for the
// weight gradient move 12-channel tensors five more times.  Thread = voxel, all Co channels:
//   k1n_bwd_reduce:      rows (sum g, sum g raw, sum dout z) per channel, as affine_bwd_reduce_kernel writes them (gn_bwd_coeffs reads them)
//   k1n_bwd_apply_wgrad: d(raw) = A g + (Cc raw + Bc) in registers, dW[co][ci] += d(raw)[co] x[ci] -> one slab per workgroup for the
//                        common fixed-order finalize; d(raw) is never written (nothing else reads it: the op has no input gradient)
// raw is formed in the forward kernel's order (bias, then ci ascending, fmaf): bit-identical to what conv_k1_kernel computed.
// ------------------------------------------------------------------------------------------------
struct K1nArgs {
  const void* x; int64_t xld; const void* dout; int64_t dld;      // TS / TD elements
  const float* w; const float* bias;                               // native (Co, Ci) weight, (Co) bias or NULL
  const float* a; const float* b;                                  // forward GroupNorm coefficients [B][Co]
  const float* A; const float* Bc; const float* Cc;                // backward coefficients [B][Co] (apply)
  double* sums; float* partial;                                    // reduce: [B][rows][Co][3]; apply: [B * rows][Ci][Co]
  int64_t N; int chunk; int relu;
};

template <int CIQ, int COQ, bool APPLY, typename TS, typename TD>
__global__ __launch_bounds__(256) void k1n_bwd_kernel(K1nArgs q) {
  constexpr int CI = CIQ * 4, CO = COQ * 4, NV = APPLY ? CI * CO : CO * 3;
  __shared__ float wsm[CO][CI + 1];
  __shared__ float red[4][NV];
  const int t = threadIdx.x, b = blockIdx.y;
  for (int i = t; i < CO * CI; i += 256) wsm[i / CI][i % CI] = q.w[i];
  float bias[CO], av[CO], bv[CO], Av[APPLY ? CO : 1], Bv[APPLY ? CO : 1], Cv[APPLY ? CO : 1];
#pragma unroll
  for (int c = 0; c < CO; ++c) {
    bias[c] = q.bias ? q.bias[c] : 0.f;
    av[c] = q.a[b * CO + c]; bv[c] = q.b[b * CO + c];
    if constexpr (APPLY) { Av[c] = q.A[b * CO + c]; Bv[c] = q.Bc[b * CO + c]; Cv[c] = q.Cc[b * CO + c]; }
  }
  float acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.f;
  const TS* xb = reinterpret_cast<const TS*>(q.x) + (int64_t)b * q.N * q.xld;
  const TD* db = reinterpret_cast<const TD*>(q.dout) + (int64_t)b * q.N * q.dld;
  const int64_t i0 = (int64_t)blockIdx.x * q.chunk;
  int64_t i1 = i0 + q.chunk;
  if (i1 > q.N) i1 = q.N;
  __syncthreads();
  for (int64_t i = i0 + t; i < i1; i += 512) {
    float4 xq[2][CIQ], gq[2][COQ];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {      // two voxels in flight per trip
      const int64_t v = i + u * 256;
      ok[u] = v < i1;
      const int64_t vc = ok[u] ? v : i;
#pragma unroll
      for (int k = 0; k < CIQ; ++k) xq[u][k] = ld4(xb + vc * q.xld + k * 4);
#pragma unroll
      for (int k = 0; k < COQ; ++k) gq[u][k] = ld4(db + vc * q.dld + k * 4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!ok[u]) continue;
      float xv[CI], d[CO];
#pragma unroll
      for (int k = 0; k < CIQ; ++k) { xv[k * 4] = xq[u][k].x; xv[k * 4 + 1] = xq[u][k].y; xv[k * 4 + 2] = xq[u][k].z; xv[k * 4 + 3] = xq[u][k].w; }
#pragma unroll
      for (int k = 0; k < COQ; ++k) { d[k * 4] = gq[u][k].x; d[k * 4 + 1] = gq[u][k].y; d[k * 4 + 2] = gq[u][k].z; d[k * 4 + 3] = gq[u][k].w; }
#pragma unroll
      for (int c = 0; c < CO; ++c) {
        float raw = bias[c];
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) raw = fmaf(xv[ci], wsm[c][ci], raw);
        float z = fmaf(av[c], raw, bv[c]);
        float g = d[c];
        if (q.relu) { g = z > 0.f ? g : 0.f; z = fmaxf(z, 0.f); }
        if constexpr (APPLY) {
          const float dr = fmaf(Av[c], g, fmaf(Cv[c], raw, Bv[c]));
#pragma unroll
          for (int ci = 0; ci < CI; ++ci) acc[ci * CO + c] = fmaf(xv[ci], dr, acc[ci * CO + c]);
        } else {
          acc[c * 3] += g;
          acc[c * 3 + 1] = fmaf(g, raw, acc[c * 3 + 1]);
          acc[c * 3 + 2] = fmaf(d[c], z, acc[c * 3 + 2]);
        }
      }
    }
  }
  // block sums: four accumulators per packed wave sum (see conv_k1_wgrad_kernel), the four waves added in a fixed order
  const int wave = t >> 6, lane = t & 63;
  static_assert(NV % 4 == 0, "k1n: NV must be a multiple of 4");
  const int sel = classsum4_sel(lane);
  const bool wr = (lane & 15) == 0;
#pragma unroll
  for (int j = 0; j < NV; j += 4) {
    const float sv = wave_classsum4_f<1>(acc[j], acc[j + 1], acc[j + 2], acc[j + 3]);
    if (wr) red[wave][j + sel] = sv;
  }
  __syncthreads();
  const int64_t row = (int64_t)b * gridDim.x + blockIdx.x;
  for (int j = t; j < NV; j += 256) {
    if constexpr (APPLY) q.partial[row * NV + j] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
    else q.sums[row * NV + j] = (double)red[0][j] + (double)red[1][j] + (double)red[2][j] + (double)red[3][j];
  }
}

static bool k1n_shape_ok(const n3d_conv_geom* g) {
  const int64_t N = (int64_t)g->Do * g->Ho * g->Wo;
  return g->k == 1 && g->stride == 1 && g->pad == 0 && !g->depthwise && (g->Ci == 4 || g->Ci == 8) && (g->Co == 4 || g->Co == 8 || g->Co == 12) &&
         (g->Ci / 4) * (g->Co / 4) <= 3 && N >= 32768 && N < (1ll << 31);
}
static int k1n_chunk(int64_t N) {      // voxels per workgroup: 2048, fewer on the small levels (>= 256 workgroups per sample where possible)
  int64_t chunk = 2048;
  while (chunk > 512 && cdiv(N, chunk) < 256) chunk >>= 1;
  return (int)chunk;
}

template <bool APPLY, typename TS, typename TD>
static bool launch_k1n_t(const K1nArgs& q, int Ci, int Co, dim3 grid, hipStream_t s) {
  if (Ci == 4 && Co == 4) hipLaunchKernelGGL((k1n_bwd_kernel<1, 1, APPLY, TS, TD>), grid, dim3(256), 0, s, q);
  else if (Ci == 4 && Co == 8) hipLaunchKernelGGL((k1n_bwd_kernel<1, 2, APPLY, TS, TD>), grid, dim3(256), 0, s, q);
  else if (Ci == 4 && Co == 12) hipLaunchKernelGGL((k1n_bwd_kernel<1, 3, APPLY, TS, TD>), grid, dim3(256), 0, s, q);
  else if (Ci == 8 && Co == 4) hipLaunchKernelGGL((k1n_bwd_kernel<2, 1, APPLY, TS, TD>), grid, dim3(256), 0, s, q);
  else return false;
  return true;
}
template <bool APPLY>
static bool launch_k1n(const K1nArgs& q, int Ci, int Co, int flags, dim3 grid, hipStream_t s) {
  const bool sb = flags & N3D_SRC_BF16, db = flags & N3D_DST_BF16;
  if (sb && db) return launch_k1n_t<APPLY, bf16_t, bf16_t>(q, Ci, Co, grid, s);
  if (sb) return launch_k1n_t<APPLY, bf16_t, float>(q, Ci, Co, grid, s);
  if (db) return launch_k1n_t<APPLY, float, bf16_t>(q, Ci, Co, grid, s);
  return launch_k1n_t<APPLY, float, float>(q, Ci, Co, grid, s);
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dW[co][ci][tap] = sum_{b,o} dy[b,o,co] * f(x[b, o*s - pad + tap*dil, ci])
// one block = (chunk of flattened (b,o), one (tap, ci tile, co tile)); per-thread register outer
// products, wave reduction, partial slabs, then a fixed-order final reduction (deterministic).
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
  const void* x; int64_t xld; int Di, Hi, Wi, Ci;      // TS elements
  const void* dy; int64_t dyld; int Do, Ho, Wo, Co;    // TD elements
  int B, k, stride, dil, pad, flags;
  const float* in_gate;
  float* partial;   // [nchunks][ntiles][CI_T*CO_T]
  float* pbias;     // [nchunks][tco][CO_T]
  int tci, tco;
  int64_t chunk;
  FastDiv fNo, fWo, fHo;
};

template <int CI_T, int CO_T, typename TS = float, typename TD = float>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  __shared__ float red[4][CI_T * CO_T + CO_T];
  const int tile = blockIdx.y;
  const int cot = tile % a.tco, cit = (tile / a.tco) % a.tci, tap = tile / (a.tco * a.tci);
  const int kw = tap % a.k, kh = (tap / a.k) % a.k, kd = tap / (a.k * a.k);
  const int64_t No = (int64_t)a.Do * a.Ho * a.Wo, Ni = (int64_t)a.Di * a.Hi * a.Wi;
  const int64_t total = (int64_t)a.B * No;
  const int64_t i0 = (int64_t)blockIdx.x * a.chunk;
  int64_t i1 = i0 + a.chunk;
  if (i1 > total) i1 = total;
  const bool relu_in = a.flags & N3D_RELU_IN;
  const bool do_bias = (tap == 0 && cit == 0);
  const bool covec = (a.Co % 4 == 0) && (a.dyld % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.dy) & (4 * sizeof(TD) - 1)) == 0);
  float acc[CI_T][CO_T];
  float bacc[CO_T];
#pragma unroll
  for (int i = 0; i < CI_T; ++i)
#pragma unroll
    for (int j = 0; j < CO_T; ++j) acc[i][j] = 0.f;
#pragma unroll
  for (int j = 0; j < CO_T; ++j) bacc[j] = 0.f;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    uint32_t ub, uo, q1, uw, ud, uh;
    a.fNo.divmod((uint32_t)i, ub, uo);
    a.fWo.divmod(uo, q1, uw);
    a.fHo.divmod(q1, ud, uh);
    const int b = (int)ub, ow = (int)uw, oh = (int)uh, od = (int)ud;
    float dyv[CO_T];
    const TD* dp = reinterpret_cast<const TD*>(a.dy) + i * a.dyld + cot * CO_T;
    if (covec) {
#pragma unroll
      for (int j = 0; j < CO_T; j += 4) {
        const float4 q = ld4(dp + j);
        dyv[j] = q.x; dyv[j + 1] = q.y; dyv[j + 2] = q.z; dyv[j + 3] = q.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < CO_T; ++j) dyv[j] = (cot * CO_T + j < a.Co) ? ld1(dp + j) : 0.f;
    }
    if (do_bias) {
#pragma unroll
      for (int j = 0; j < CO_T; ++j) bacc[j] += dyv[j];
    }
    const int id = od * a.stride - a.pad + kd * a.dil, ih = oh * a.stride - a.pad + kh * a.dil, iw = ow * a.stride - a.pad + kw * a.dil;
    if (id < 0 || id >= a.Di || ih < 0 || ih >= a.Hi || iw < 0 || iw >= a.Wi) continue;
    const TS* xp = reinterpret_cast<const TS*>(a.x) + ((int64_t)b * Ni + ((int64_t)id * a.Hi + ih) * a.Wi + iw) * a.xld + cit * CI_T;
    float xv[CI_T];
#pragma unroll
    for (int c = 0; c < CI_T; c += 4) {
      const float4 q = ld4(xp + c);
      xv[c] = q.x; xv[c + 1] = q.y; xv[c + 2] = q.z; xv[c + 3] = q.w;
    }
#pragma unroll
    for (int c = 0; c < CI_T; ++c) {
      if (relu_in) xv[c] = fmaxf(xv[c], 0.f);
      if (a.in_gate) xv[c] *= a.in_gate[(int64_t)b * a.Ci + cit * CI_T + c];
    }
#pragma unroll
    for (int c = 0; c < CI_T; ++c)
#pragma unroll
      for (int j = 0; j < CO_T; ++j) acc[c][j] = fmaf(xv[c], dyv[j], acc[c][j]);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // wave sums four accumulators at a time (wave_classsum4_f: 3 permlane swaps per four values instead of 8): row r of the wave ends up
  // with the total of value {0, 2, 1, 3}[r], its first lane writes it
  static_assert(CO_T % 4 == 0, "conv_wgrad_kernel: CO_T must be a multiple of 4");
  const int sel = classsum4_sel(lane);
  const bool wr = (lane & 15) == 0;
#pragma unroll
  for (int c = 0; c < CI_T; ++c)
#pragma unroll
    for (int j = 0; j < CO_T; j += 4) {
      const float s = wave_classsum4_f<1>(acc[c][j], acc[c][j + 1], acc[c][j + 2], acc[c][j + 3]);
      if (wr) red[wave][c * CO_T + j + sel] = s;
    }
#pragma unroll
  for (int j = 0; j < CO_T; j += 4) {
    const float s = wave_classsum4_f<1>(bacc[j], bacc[j + 1], bacc[j + 2], bacc[j + 3]);
    if (wr) red[wave][CI_T * CO_T + j + sel] = s;
  }
  __syncthreads();
  const int ntiles = gridDim.y;
  for (int q = threadIdx.x; q < CI_T * CO_T; q += 256)
    a.partial[((int64_t)blockIdx.x * ntiles + tile) * (CI_T * CO_T) + q] = red[0][q] + red[1][q] + red[2][q] + red[3][q];
  if (do_bias && threadIdx.x < CO_T) {
    const int q = CI_T * CO_T + threadIdx.x;
    a.pbias[((int64_t)blockIdx.x * a.tco + cot) * CO_T + threadIdx.x] = red[0][q] + red[1][q] + red[2][q] + red[3][q];
  }
}


// ------------------------------------------------------------------------------------------------
// depthwise family (groups == C): per-lane weights, float4 over 4 channels
// ------------------------------------------------------------------------------------------------
struct DwArgs {
  const void* src; int64_t sld; int Ds, Hs, Ws;      // TS elements (fp32, or bf16 storage: round 5)
  void* dst; int64_t dld; int Dd, Hd, Wd;            // TD elements
  int C;
  const float* w;   // native (C,1,k,k,k)
  const float* bias;
  int k, sn, off, dt, den, flags;
  FastDiv fcpb, fWd, fHd;
};

template <typename TS = float, typename TD = float>
__device__ __forceinline__ void dw_gather_body(const DwArgs& a) {
  // k = 3, dilation 1 (check_geom).  Branch-free: the weights of all channel quads are staged once in LDS as float4
  // [c4][tap] (one broadcast ds_read_b128 per tap), and the nine source loads of a kd plane are issued from clamped
  // addresses before their first use -- the rolled tap loop with `continue`s paid one memory latency per tap (27).
  extern __shared__ __attribute__((aligned(16))) float4 dwl[];   // [C/4][27]
  const int cpb = a.C / 4;
  for (int i = threadIdx.x; i < cpb * 27; i += 256) {
    const int c4 = i / 27, tap = i - c4 * 27;
    const float* wc = a.w + (int64_t)c4 * 4 * 27 + tap;
    dwl[i] = make_float4(wc[0], wc[27], wc[54], wc[81]);
  }
  __syncthreads();
  const int64_t Nd = (int64_t)a.Dd * a.Hd * a.Wd, Ns = (int64_t)a.Ds * a.Hs * a.Ws;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= Nd * cpb) return;
  const int b = blockIdx.y;
  uint32_t uv, uc, q1, uw, ud, uh;
  a.fcpb.divmod((uint32_t)idx, uv, uc);
  a.fWd.divmod(uv, q1, uw);
  a.fHd.divmod(q1, ud, uh);
  const int c4 = (int)uc;
  const int64_t v = uv;
  const int w_ = (int)uw, h_ = (int)uh, d_ = (int)ud;
  TD* op = reinterpret_cast<TD*>(a.dst) + ((int64_t)b * Nd + v) * a.dld + c4 * 4;
  const bool accum = a.flags & N3D_ACCUMULATE;
  float4 prev = make_float4(0.f, 0.f, 0.f, 0.f);
  if (accum) prev = ld4(op);
  float4 acc = a.bias ? *reinterpret_cast<const float4*>(a.bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const TS* srcb = reinterpret_cast<const TS*>(a.src) + (int64_t)b * Ns * a.sld + c4 * 4;
  const float4* wq = dwl + c4 * 27;
#pragma unroll
  for (int kd = 0; kd < 3; ++kd) {
    int nd = d_ * a.sn + a.off + kd * a.dt;
    bool okd = true;
    if (a.den == 2) { okd = !(nd & 1); nd >>= 1; }
    okd = okd && nd >= 0 && nd < a.Ds;
    const int cd_ = min(max(nd, 0), a.Ds - 1);
    float4 q[3][3];
    float m[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      int nh = h_ * a.sn + a.off + kh * a.dt;
      bool okh = okd;
      if (a.den == 2) { okh = okh && !(nh & 1); nh >>= 1; }
      okh = okh && nh >= 0 && nh < a.Hs;
      const int ch_ = min(max(nh, 0), a.Hs - 1);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        int nw = w_ * a.sn + a.off + kw * a.dt;
        bool ok = okh;
        if (a.den == 2) { ok = ok && !(nw & 1); nw >>= 1; }
        ok = ok && nw >= 0 && nw < a.Ws;
        const int cw_ = min(max(nw, 0), a.Ws - 1);
        m[kh][kw] = ok ? 1.f : 0.f;
        q[kh][kw] = ld4(srcb + (((int64_t)cd_ * a.Hs + ch_) * a.Ws + cw_) * a.sld);
      }
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float4 wv = wq[(kd * 3 + kh) * 3 + kw];
        const float mm = m[kh][kw];
        acc.x = fmaf(q[kh][kw].x * mm, wv.x, acc.x);
        acc.y = fmaf(q[kh][kw].y * mm, wv.y, acc.y);
        acc.z = fmaf(q[kh][kw].z * mm, wv.z, acc.z);
        acc.w = fmaf(q[kh][kw].w * mm, wv.w, acc.w);
      }
  }
  st4(op, make_float4(acc.x + prev.x, acc.y + prev.y, acc.z + prev.z, acc.w + prev.w));
}
template <typename TS = float, typename TD = float>
__global__ __launch_bounds__(256) void dw_gather_kernel(DwArgs a) { N3D_CHAIN_PRIO(); dw_gather_body<TS, TD>(a); }
// the depthwise convs of up to 8 primitives of a supernet node (one per edge, cell.py:76-81) in one launch: grid.z = job; all
// jobs share the output shape and channel count (blockIdx.x covers the output voxels), their sources may differ in shape
struct DwArgsN { DwArgs j[8]; };
__global__ __launch_bounds__(256) void dw_gatherN_kernel(DwArgsN js) { N3D_CHAIN_PRIO();
  DwArgs a;
  switch (blockIdx.z) { case 0: a = js.j[0]; break; case 1: a = js.j[1]; break; case 2: a = js.j[2]; break; case 3: a = js.j[3]; break;
                        case 4: a = js.j[4]; break; case 5: a = js.j[5]; break; case 6: a = js.j[6]; break; default: a = js.j[7]; break; }
  dw_gather_body<float, float>(a);
}

struct DwWgradArgs {
  const void* x; int64_t xld; int Di, Hi, Wi;        // TS elements
  const void* dy; int64_t dyld; int Do, Ho, Wo;      // TD elements
  int B, C, k, stride, pad;
  float* partial;  // [nchunks][27][C]  (the layout n3d_wgrad_finalize_batch reads with ci_t = 1, co_t = C)
  float* pbias;    // [nchunks][C]
  int64_t chunk;   // flattened (b,o) voxels per block
  FastDiv fNo, fWo, fHo;
};

// POW2: C / 4 is a power of two (every depthwise op of the reference: C = 4 .. 64) -> DPP / permlane class sums, fully
// unrolled so the 28 x 4 accumulators stay in registers (the generic strided sums made the compiler spill them)
// CPB: C / 4 when it is a power of two (compile-time, so the 112 class sums are straight-line DPP code), 0 = generic
template <int CPB, typename TS = float, typename TD = float>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(DwWgradArgs a) {
  constexpr bool POW2 = CPB > 0;
  extern __shared__ float dyn[];
  const int cpb = CPB > 0 ? CPB : a.C / 4, vpb = 256 / cpb;
  const int t = threadIdx.x, c4 = t % cpb, vl = t / cpb;
  const int64_t No = (int64_t)a.Do * a.Ho * a.Wo, Ni = (int64_t)a.Di * a.Hi * a.Wi;
  const int64_t total = (int64_t)a.B * No;
  const int64_t i0 = (int64_t)blockIdx.x * a.chunk;
  int64_t i1 = i0 + a.chunk;
  if (i1 > total) i1 = total;
  float acc[28][4];
#pragma unroll
  for (int q = 0; q < 28; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[q][j] = 0.f;
  if (vl < vpb) {
    for (int64_t i = i0 + vl; i < i1; i += vpb) {
      uint32_t ub, uo, q1, uw, ud, uh;
      a.fNo.divmod((uint32_t)i, ub, uo);
      a.fWo.divmod(uo, q1, uw);
      a.fHo.divmod(q1, ud, uh);
      const int b = (int)ub, ow = (int)uw, oh = (int)uh, od = (int)ud;
      const float4 g = ld4(reinterpret_cast<const TD*>(a.dy) + i * a.dyld + c4 * 4);
      acc[27][0] += g.x; acc[27][1] += g.y; acc[27][2] += g.z; acc[27][3] += g.w;
      const TS* xb = reinterpret_cast<const TS*>(a.x) + (int64_t)b * Ni * a.xld + c4 * 4;
      // one kd plane at a time: its nine loads are issued from clamped addresses before the first use (a branch per
      // tap serialises on one memory latency per tap: 27 per voxel; all 27 at once measured no better and takes 256 VGPRs)
#pragma unroll
      for (int kd = 0; kd < 3; ++kd) {
        const int id = od * a.stride - a.pad + kd;
        const bool okd = id >= 0 && id < a.Di;
        const int cd_ = min(max(id, 0), a.Di - 1);
        float4 q[3][3];
        bool ok[3][3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int ih = oh * a.stride - a.pad + kh;
          const bool okh = okd && ih >= 0 && ih < a.Hi;
          const int ch_ = min(max(ih, 0), a.Hi - 1);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * a.stride - a.pad + kw;
            ok[kh][kw] = okh && iw >= 0 && iw < a.Wi;
            const int cw_ = min(max(iw, 0), a.Wi - 1);
            q[kh][kw] = ld4(xb + (((int64_t)cd_ * a.Hi + ch_) * a.Wi + cw_) * a.xld);
          }
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int tap = (kd * 3 + kh) * 3 + kw;
            const float m = ok[kh][kw] ? 1.f : 0.f;
            acc[tap][0] = fmaf(q[kh][kw].x * m, g.x, acc[tap][0]);
            acc[tap][1] = fmaf(q[kh][kw].y * m, g.y, acc[tap][1]);
            acc[tap][2] = fmaf(q[kh][kw].z * m, g.z, acc[tap][2]);
            acc[tap][3] = fmaf(q[kh][kw].w * m, g.w, acc[tap][3]);
          }
      }
    }
  }
  // wave-level strided reduction, then LDS across waves: dyn[wave][class c4][28][4]
  const int wave = t >> 6, lane = t & 63;
#pragma unroll
  for (int q = 0; q < 28; ++q)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[q][j] = POW2 ? wave_classsum_f(acc[q][j], cpb) : wave_sum_strided_f(acc[q][j], cpb);
  if (lane < cpb) {
    const int cls = (wave * 64 + lane) % cpb;
#pragma unroll
    for (int q = 0; q < 28; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) dyn[((wave * cpb + cls) * 28 + q) * 4 + j] = acc[q][j];
  }
  __syncthreads();
  const int nq = cpb * 28 * 4;
  for (int i = t; i < nq; i += 256) {
    const int j = i % 4, q = (i / 4) % 28, cls = i / (4 * 28);
    float s = 0.f;
    for (int w = 0; w < 4; ++w) s += dyn[((w * cpb + cls) * 28 + q) * 4 + j];
    if (q < 27) a.partial[((int64_t)blockIdx.x * 27 + q) * a.C + cls * 4 + j] = s;
    else a.pbias[(int64_t)blockIdx.x * a.C + cls * 4 + j] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// per-channel sum over all (b, voxel): bias gradient of a transposed convolution
// ------------------------------------------------------------------------------------------------
template <typename TX>
__global__ __launch_bounds__(256) void channel_sum_kernel(const TX* __restrict__ x, int64_t ld, int64_t total, int C, int64_t chunk,
                                                          float* __restrict__ partial /*[nblk][C]*/) {
  extern __shared__ float dyn[];
  const int cpb = C / 4, vpb = 256 / cpb;
  const int t = threadIdx.x, c4 = t % cpb, vl = t / cpb;
  const int64_t i0 = (int64_t)blockIdx.x * chunk;
  int64_t i1 = i0 + chunk;
  if (i1 > total) i1 = total;
  float acc[4] = {0, 0, 0, 0};
  if (vl < vpb)
    for (int64_t i = i0 + vl; i < i1; i += vpb) {
      const float4 q = ld4(x + i * ld + c4 * 4);
      acc[0] += q.x; acc[1] += q.y; acc[2] += q.z; acc[3] += q.w;
    }
  const int wave = t >> 6, lane = t & 63;
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = wave_sum_strided_f(acc[j], cpb);
  if (lane < cpb) {
    const int cls = (wave * 64 + lane) % cpb;
#pragma unroll
    for (int j = 0; j < 4; ++j) dyn[(wave * cpb + cls) * 4 + j] = acc[j];
  }
  __syncthreads();
  for (int i = t; i < C; i += 256) {
    const int cls = i / 4, j = i % 4;
    float s = 0.f;
    for (int w = 0; w < 4; ++w) s += dyn[(w * cpb + cls) * 4 + j];
    partial[(int64_t)blockIdx.x * C + i] = s;
  }
}
__global__ void channel_sum_final_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int k = 0; k < nblk; ++k) s += partial[(int64_t)k * C + c];
  out[c] = s;
}


// ------------------------------------------------------------------------------------------------
// depthwise 3x3x3 stride-1 weight gradient on tileable volumes: dW[c][tap] = sum_v x[v + tap][c] * dy[v][c].
// dw_wgrad_kernel reads its 27 neighbours per voxel from global memory (432 bytes per voxel and channel quad through L1 / L2)
// and pays a 112-value wave reduction per 512 voxels.  Here a workgroup stages the halo tile of one channel quad for 4 x 4 x 16
// output voxels in LDS (LDS-DMA), a thread owns one voxel (27 ds_read_b128 + 108 FMAs), walks several tiles and reduces once.
// One workgroup column per channel quad (blockIdx.y).  Same partial-slab layout as dw_wgrad_kernel.
// ------------------------------------------------------------------------------------------------
__device__ float4 n3d_dw_zero_page[1];   // zero-initialised source of the padding for the LDS-DMA fill
static const void* dw_zero_page() {
  static thread_local const void* p = nullptr;
  static thread_local int dev = -1;
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) return nullptr;
  if (!p || d != dev) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(n3d_dw_zero_page)) != hipSuccess) return nullptr;
    p = q; dev = d;
  }
  return p;
}
struct DwTileArgs {
  const float* x; int64_t xld; const float* dy; int64_t dyld;
  int D, H, W, B, C;
  float* partial;  // [chunks][27][C]
  float* pbias;    // [chunks][C]
  int tiles_per_sample, tiles_total, tiles_per_wg;
  const void* zero_page;
};

__global__ __launch_bounds__(256) void dw_wgrad_tile_kernel(DwTileArgs a) {
  constexpr int TD = 4, TH = 4, TW = 16, LD = TD + 2, LH = TH + 2, LW = TW + 2, NV = LD * LH * LW, NIT = (NV + 255) / 256;
  __shared__ __attribute__((aligned(16))) float4 tl[NIT * 256];
  __shared__ float red[4][28 * 4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int quad = blockIdx.y;
  const int tx = t & 15, ty = (t >> 4) & 3, tz = t >> 6;   // this thread's voxel of the tile (a wave = one z plane)
  const int D = a.D, H = a.H, W = a.W;
  const int tw_n = W / TW, th_n = H / TH;
  const int64_t N = (int64_t)D * H * W;
  float4 acc[27], bs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const float4* zp = reinterpret_cast<const float4*>(a.zero_page);
  for (int it = 0; it < a.tiles_per_wg; ++it) {
    const int tg = (int)blockIdx.x * a.tiles_per_wg + it;
    if (tg >= a.tiles_total) break;
    const int b = tg / a.tiles_per_sample, tid = tg - b * a.tiles_per_sample;
    const int w0 = (tid % tw_n) * TW, h0 = ((tid / tw_n) % th_n) * TH, d0 = (tid / (tw_n * th_n)) * TD;
    const float4 g = *reinterpret_cast<const float4*>(a.dy + ((int64_t)b * N + ((int64_t)(d0 + tz) * H + h0 + ty) * W + w0 + tx) * a.dyld + quad * 4);
    const float* xb = a.x + (int64_t)b * N * a.xld + quad * 4;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int v = (i * 4 + wave) * 64 + lane;
      const int x = v % LW, y = (v / LW) % LH, z = v / (LW * LH);
      const int gd = d0 - 1 + z, gh = h0 - 1 + y, gw = w0 - 1 + x;
      const bool ok = v < NV && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
      const float* sp = xb + (((int64_t)gd * H + gh) * W + gw) * a.xld;
      __builtin_amdgcn_global_load_lds((gptr_t)(ok ? reinterpret_cast<const float4*>(sp) : zp), (lptr_t)(tl + (i * 4 + wave) * 64), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float4* c0 = tl + (tz * LH + ty) * LW + tx;
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float4 q = c0[(kd * LH + kh) * LW + kw];
          float4& r = acc[(kd * 3 + kh) * 3 + kw];
          r.x = fmaf(q.x, g.x, r.x); r.y = fmaf(q.y, g.y, r.y); r.z = fmaf(q.z, g.z, r.z); r.w = fmaf(q.w, g.w, r.w);
        }
    bs.x += g.x; bs.y += g.y; bs.z += g.z; bs.w += g.w;
    __syncthreads();   // the next tile's fill overwrites the image
  }
  // wave sums, then the four waves through LDS
#pragma unroll
  for (int k = 0; k < 28; ++k) {
    const float4 v = k < 27 ? acc[k < 27 ? k : 0] : bs;
    const float s0 = wave_sum_f(v.x), s1 = wave_sum_f(v.y), s2 = wave_sum_f(v.z), s3 = wave_sum_f(v.w);
    if (lane == 0) { red[wave][k * 4] = s0; red[wave][k * 4 + 1] = s1; red[wave][k * 4 + 2] = s2; red[wave][k * 4 + 3] = s3; }
  }
  __syncthreads();
  if (t < 28 * 4) {
    const int k = t >> 2, j = t & 3;
    const float s = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    if (k < 27) a.partial[((int64_t)blockIdx.x * 27 + k) * a.C + quad * 4 + j] = s;
    else a.pbias[(int64_t)blockIdx.x * a.C + quad * 4 + j] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// batched weight packing / batched weight-gradient reduction: the job table travels BY VALUE in the kernel
// arguments (graph-capturable, no device-side table to maintain), compacted to 16 / 36 bytes per job so that a whole
// train step fits one 4 KB argument block: a pointer becomes (segment : 3 bits, float offset : 29 bits) against up to
// eight 2 GB address segments of the batch (parameters, gradients, packed slots and workspaces live in a few allocations).
// layouts: 0 generic [tap][cs][cdp]; 1 gemm16 [tap][cs/16][kk][cd][j]; 2 vox64 [tap][cd][cs] (flipped for data grad);
//          3 vox_up (as 2, never flipped); 4 / 5 = 2 / 3 in bfloat16
// ------------------------------------------------------------------------------------------------
#define N3D_PACK_JOBS 160
#define N3D_FINAL_JOBS 80
#define N3D_JOB_UNITS 512      // entries of the workgroup -> job map
#define N3D_PACK_GSHIFT 2      // a map entry covers 4 workgroups of the pack kernel ...
#define N3D_FINAL_GSHIFT 4     // ... and 16 of the reduction kernel (a job's workgroup count is rounded up to whole entries)
#define N3D_NO_OFF 0xffffffffu
struct PackJobD { uint32_t w, dst, start; uint16_t Co, Ci, cdp; uint8_t taps, mode; };   // mode = layout | data_grad << 4
struct FinalJobD { uint32_t partial, pbias, dw, dbias, start; int32_t nchunks; uint16_t ntiles, Co, Ci, ci_t, co_t; uint8_t tci, tco, taps, pad_; };
struct SegBases { uintptr_t b[8]; };
// The kernel-argument block is slow memory for anything but the one scalar fetch every kernel starts with: each DEPENDENT
// read is a ~1-2 us round trip (a binary search over start offsets cost 8 of them: 20 us for a 4 us kernel) and vector
// reads of it are slower still (every wave crossing to host-visible memory: 43 us).  So a workgroup makes exactly two scalar
// trips: (1) the eight bases and its entry of the workgroup -> job byte map, (2) its job record.  The empty asm statements
// pin the loads where they are written (the compiler otherwise sinks them behind branches and turns the register selects
// back into indexed loads = more trips).
// the bases stay eight scalar VALUES (macro-declared locals passed by value): kept in a private struct or array, the
// compiler turns the select chain into an indexed load and moves the array to LDS
__device__ __forceinline__ float* seg_sel(uint32_t off, uint64_t b0, uint64_t b1, uint64_t b2, uint64_t b3, uint64_t b4, uint64_t b5, uint64_t b6,
                                          uint64_t b7) {
  const uint32_t k = off >> 29;
  uint64_t b = b0;
  b = k == 1 ? b1 : b; b = k == 2 ? b2 : b; b = k == 3 ? b3 : b; b = k == 4 ? b4 : b;
  b = k == 5 ? b5 : b; b = k == 6 ? b6 : b; b = k == 7 ? b7 : b;
  return reinterpret_cast<float*>(b) + (off & 0x1fffffffu);
}
// trip 1: the eight bases (sb0..sb7) and the map entry of this workgroup (-> jx), one wait for all of them
#define N3D_FETCH_HEAD(jobs, GSHIFT)                                                                                              \
  int u_ = (int)blockIdx.x >> (GSHIFT);                                                                                           \
  u_ = u_ < N3D_JOB_UNITS ? u_ : N3D_JOB_UNITS - 1; /* (a single job larger than the map: every entry names it) */                \
  uint32_t w_ = (jobs).unit[u_ >> 2];                                                                                             \
  uint64_t sb0 = (jobs).seg.b[0], sb1 = (jobs).seg.b[1], sb2 = (jobs).seg.b[2], sb3 = (jobs).seg.b[3], sb4 = (jobs).seg.b[4],     \
           sb5 = (jobs).seg.b[5], sb6 = (jobs).seg.b[6], sb7 = (jobs).seg.b[7];                                                   \
  asm volatile("" : "+s"(w_), "+s"(sb0), "+s"(sb1), "+s"(sb2), "+s"(sb3), "+s"(sb4), "+s"(sb5), "+s"(sb6), "+s"(sb7));            \
  const int jx = (w_ >> ((u_ & 3) * 8)) & 255
#define N3D_SEG(off) seg_sel(off, sb0, sb1, sb2, sb3, sb4, sb5, sb6, sb7)
template <typename J>
__device__ __forceinline__ J fetch_job(const J& src) {
  struct Words { uint32_t w[sizeof(J) / 4]; };
  static_assert(sizeof(J) % 4 == 0, "job records are whole dwords");
  Words t = *reinterpret_cast<const Words*>(&src);
#pragma unroll
  for (unsigned i = 0; i < sizeof(J) / 4; ++i) asm volatile("" : "+s"(t.w[i]));
  return __builtin_bit_cast(J, t);
}
struct PackJobs { SegBases seg; uint32_t unit[N3D_JOB_UNITS / 4]; PackJobD j[N3D_PACK_JOBS]; };
struct FinalJobs { SegBases seg; uint32_t unit[N3D_JOB_UNITS / 4]; FinalJobD j[N3D_FINAL_JOBS]; };
static_assert(sizeof(PackJobs) <= 4000 && sizeof(FinalJobs) <= 4000, "job tables must fit the kernel argument block");
static_assert(N3D_PACK_JOBS <= 256 && N3D_FINAL_JOBS <= 256, "job ids are bytes");

// elements of one tap of the packed form (the tap is the slowest index of every layout)
__host__ __device__ inline int pack_tap_elems(int layout, int Cs, int Cd, int cdp, int Co) {
  return layout == 0 ? Cs * cdp : (layout == 1 ? Cs * Cd : Co * Co);
}

// destination channels of the packed form (layout 0 pads them to cdp)
__host__ __device__ inline int pack_dst_channels(int layout, int Cd, int cdp) { return layout == 0 ? cdp : Cd; }
// workgroups of one pack job: 3x3x3 weights go by 16 x 16 (source, destination) channel tiles, others by 256 elements of a tap
__host__ __device__ inline int pack_job_blocks(int layout, int Cs, int Cd, int cdp, int Co, int taps) {
  if (taps == 27) return ((Cs + 15) / 16) * ((pack_dst_channels(layout, Cd, cdp) + 15) / 16);
  return (pack_tap_elems(layout, Cs, Cd, cdp, Co) + 255) / 256;
}
#define N3D_PACK_PITCH 433   // 16 input channels x 27 taps + 1: odd, so that lanes walking output channels spread over the LDS banks

__global__ __launch_bounds__(256) void pack_batch_kernel(PackJobs jobs) {
  // flattened grid: the byte map names the job of a workgroup, the job record its first workgroup
  __shared__ float tile[16 * N3D_PACK_PITCH];
  N3D_FETCH_HEAD(jobs, N3D_PACK_GSHIFT);
  const PackJobD jb = fetch_job(jobs.j[jx]);
  const int lb = (int)blockIdx.x - (int)jb.start;
  const int Co = jb.Co, Ci = jb.Ci, taps = jb.taps, layout = jb.mode & 15;
  const bool data_grad = (jb.mode >> 4) != 0;
  const int Cs = data_grad ? Co : Ci, Cd = data_grad ? Ci : Co;
  const int E = pack_tap_elems(layout, Cs, Cd, jb.cdp, Co);
  const float* __restrict__ w = N3D_SEG(jb.w);
  float* __restrict__ dst = N3D_SEG(jb.dst);
  const bool bf16 = layout >= 4;
  const int t = threadIdx.x;
  if (taps == 27) {
    // The native weight is (Co, Ci, 27): for one output channel, 16 input channels are 432 consecutive floats.  A workgroup
    // takes a 16 x 16 channel tile: the 16 runs are read as they lie (coalesced) into LDS, then every thread writes the 27
    // taps of one (source, destination) pair, consecutive lanes to consecutive packed addresses.  A train step packs 3.5 M
    // weights twice (forward and data-gradient form): 28 MB moved in 14 us inside the step's graph, of which 2 us are the
    // two argument-block trips (measured with the loads / the stores compiled out: 9 / 8 us).
    const int CdP = pack_dst_channels(layout, Cd, jb.cdp);
    const int tiles_d = (CdP + 15) / 16;
    if (lb >= ((Cs + 15) / 16) * tiles_d) return;
    const int cs0 = (lb / tiles_d) * 16, cd0 = (lb % tiles_d) * 16;
    const int co0 = data_grad ? cs0 : cd0, ci0 = data_grad ? cd0 : cs0;
#pragma unroll
    for (int i = 0; i < 27; ++i) {
      const int e = i * 256 + t;
      const int co_l = e / 432, wq = e - co_l * 432, ci_l = wq / 27;
      const bool ok = co0 + co_l < Co && ci0 + ci_l < Ci;
      tile[co_l * N3D_PACK_PITCH + wq] = ok ? w[((int64_t)(co0 + co_l) * Ci + ci0) * 27 + wq] : 0.f;
    }
    __syncthreads();
    int cs_l, cd_l;
    if (layout == 0) { cd_l = t & 15; cs_l = t >> 4; }
    else if (layout == 1) { cd_l = (t >> 2) & 15; cs_l = (t >> 6) * 4 + (t & 3); }
    else { cs_l = t & 15; cd_l = t >> 4; }
    const int cs = cs0 + cs_l, cd = cd0 + cd_l;
    if (cs >= Cs || cd >= CdP) return;
    int r;
    if (layout == 0) r = cs * jb.cdp + cd;
    else if (layout == 1) r = (((cs >> 4) * 4 + ((cs >> 2) & 3)) * Cd + cd) * 4 + (cs & 3);
    else r = cd * Co + cs;
    const bool live = cd < Cd;
    const float* src = tile + (data_grad ? cs_l : cd_l) * N3D_PACK_PITCH + (data_grad ? cd_l : cs_l) * 27;
    // layouts 2 / 4: the data gradient is the same kernel run with the taps mirrored
    const bool flip = data_grad && (layout == 2 || layout == 4);
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const float v = live ? src[flip ? 26 - k : k] : 0.f;
      if (bf16) st1(reinterpret_cast<bf16_t*>(dst) + (int64_t)k * E + r, v);
      else dst[(int64_t)k * E + r] = v;
    }
    return;
  }
  // other kernel sizes (1x1x1 above all): one thread per element of a tap
  const int r = lb * 256 + t;
  if (r >= E) return;
  int cs, cd;
  if (layout == 0) {
    cd = r % jb.cdp; cs = r / jb.cdp;
  } else if (layout == 1) {
    const int j = r & 3, rest = (r >> 2) / Cd;
    cd = (r >> 2) % Cd;
    cs = (rest >> 2) * 16 + (rest & 3) * 4 + j;
  } else {
    cs = r % Co; cd = r / Co;
  }
  const bool live = cd < Cd;
  const int co = data_grad ? cs : cd, ci = data_grad ? cd : cs;
  const float* src = w + ((int64_t)co * Ci + ci) * taps;
  const bool flip = data_grad && (layout == 2 || layout == 4);
  for (int k = 0; k < taps; ++k) {
    const float v = live ? src[flip ? taps - 1 - k : k] : 0.f;
    if (bf16) st1(reinterpret_cast<bf16_t*>(dst) + (int64_t)k * E + r, v);
    else dst[(int64_t)k * E + r] = v;
  }
}

#define N3D_FINAL_DIRECT_MAX 16  // jobs with at most this many chunks: one thread sums all chunks of its slab position

// many-chunk jobs: a workgroup covers 2^k slab positions x 256 / 2^k chunk segments; the more chunks, the more segments,
// so that no thread walks more than ~32 rows (a head gradient at 128^3 has 2048 rows of 36 floats: with a fixed
// 32 x 8 split two workgroups walked 256 rows each, one memory round trip per eight rows).  Measured on the 75 jobs of
// a 64^3 train step (32 MB of partial slabs): 29 us in two launches before, 21 us in one now.
__host__ __device__ inline int final_pos_log2(int nchunks) { return nchunks <= 64 ? 6 : (nchunks <= 256 ? 4 : (nchunks <= 1024 ? 3 : 2)); }
// few-chunk jobs whose tile fits a workgroup go by (ci tile, co tile): all taps of the tile, transposed through LDS
#define N3D_FINAL_TILE_FLOATS 7168   // >= co_t * (ci_t * taps + 1) for ci_t * co_t <= 256, taps <= 27
__host__ __device__ inline bool final_tile_path(int nchunks, int ci_t, int co_t, int taps) {
  return nchunks <= 4 && ci_t * co_t <= 256 && taps <= 27;
}
__host__ __device__ inline int final_job_blocks(int nchunks, int ntiles, int tci, int tco, int ci_t, int co_t, int taps) {
  if (final_tile_path(nchunks, ci_t, co_t, taps)) return tci * tco;
  const int el = ntiles * ci_t * co_t + tco * co_t;
  return nchunks <= N3D_FINAL_DIRECT_MAX ? (el + 255) / 256 : (el + (1 << final_pos_log2(nchunks)) - 1) >> final_pos_log2(nchunks);
}

__device__ __forceinline__ void final_store(const FinalJobD& jb, float* dwp, float* dbp, const int p, const int nslab, const int nb, const int T, const float tot) {
  if (p < nslab) {
    if (jb.dw != N3D_NO_OFF) {
      const int tile = p / T, q = p - tile * T;
      const int cot = tile % jb.tco, cit = (tile / jb.tco) % jb.tci, tap = tile / (jb.tco * jb.tci);
      const int ci = cit * jb.ci_t + q / jb.co_t, co = cot * jb.co_t + q % jb.co_t;
      if (ci < jb.Ci && co < jb.Co) dwp[((int64_t)co * jb.Ci + ci) * jb.taps + tap] = tot;
    }
  } else if (p < nslab + nb && jb.dbias != N3D_NO_OFF) {
    const int co = p - nslab;
    if (co < jb.Co) dbp[co] = tot;
  }
}

template <int NL>
__device__ __forceinline__ float final_direct(const float* __restrict__ partial, const int nchunks, const int64_t cs, const int p) {
  float v[NL];
#pragma unroll
  for (int c = 0; c < NL; ++c) v[c] = partial[(c < nchunks ? c : 0) * cs + p];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NL; ++c) s += (c < nchunks) ? v[c] : 0.f;
  return s;
}

__global__ __launch_bounds__(256) void wgrad_final_batch_kernel(FinalJobs jobs) {
  // Slab positions are walked in the order the partial slabs are stored in (tile-major), so every slab row is read
  // as contiguous pieces (reading in weight-tensor order walks the slabs with a multi-KB stride per lane); sums are
  // formed in a fixed order (deterministic) and scattered to the native (Co, Ci, taps) weight layout.
  __shared__ float tile[N3D_FINAL_TILE_FLOATS];
  float* seg = tile;
  N3D_FETCH_HEAD(jobs, N3D_FINAL_GSHIFT);
  const FinalJobD jb = fetch_job(jobs.j[jx]);
  const int first = (int)jb.start;
  const float* partial = N3D_SEG(jb.partial);
  const float* pbias = jb.pbias != N3D_NO_OFF ? N3D_SEG(jb.pbias) : nullptr;
  float* dwp = N3D_SEG(jb.dw);
  float* dbp = N3D_SEG(jb.dbias);
  const bool has_dw = jb.dw != N3D_NO_OFF, has_db = jb.dbias != N3D_NO_OFF;
  const int lb = blockIdx.x - first;
  const int T = jb.ci_t * jb.co_t;
  const int nslab = jb.ntiles * T;
  const int nb = jb.tco * jb.co_t;
  const int64_t cs = nslab;
  if (final_tile_path(jb.nchunks, jb.ci_t, jb.co_t, jb.taps)) {
    // At most four chunks, tile <= 256 positions: the workgroup owns one (ci tile, co tile) with ALL its taps.  Thread q sums its slab
    // position of every tap (each tap's tile is a contiguous piece of every slab row) into LDS laid out as the native weight
    // wants it, [co][ci][tap]; the tile then leaves as co_t runs of ci_t * taps consecutive floats.  (Scattering every sum
    // straight to (co, ci, tap) writes 4 bytes per cache line and the 27 taps of a line come from 27 different workgroups.)
    if (lb >= jb.tci * jb.tco) return;
    const int cit = lb / jb.tco, cot = lb - cit * jb.tco;
    const int q = threadIdx.x, taps = jb.taps, run = jb.ci_t * taps, pitch = run | 1;   // odd pitch: co_l-consecutive lanes spread over the banks
    if (has_dw) {
      if (q < T) {
        const int ci_l = q / jb.co_t, co_l = q - ci_l * jb.co_t;
        float* trow = tile + co_l * pitch + ci_l * taps;
        const float* pq = partial + (int64_t)(cit * jb.tco + cot) * T + q;
        const int64_t tap_stride = (int64_t)jb.tci * jb.tco * T;
        if (taps == 27) {
          // every row of every tap requested before the first sum: one memory round trip per thread
          float v[27][4];
#pragma unroll
          for (int k = 0; k < 27; ++k)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[k][c] = pq[k * tap_stride + (c < jb.nchunks ? c : 0) * cs];
#pragma unroll
          for (int k = 0; k < 27; ++k) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) a += (c < jb.nchunks) ? v[k][c] : 0.f;
            trow[k] = a;
          }
        } else {
#pragma unroll 3
          for (int k = 0; k < taps; ++k) trow[k] = final_direct<4>(pq + k * tap_stride, jb.nchunks, cs, 0);
        }
      }
      __syncthreads();
      const int total = jb.co_t * run;
      for (int e = q; e < total; e += 256) {
        const int co_l = e / run, wq = e - co_l * run, ci_l = wq / taps;
        const int co = cot * jb.co_t + co_l, ci = cit * jb.ci_t + ci_l;
        if (co < jb.Co && ci < jb.Ci) dwp[((int64_t)co * jb.Ci + cit * jb.ci_t) * taps + wq] = tile[co_l * pitch + wq];
      }
    }
    if (has_db && cit == 0 && q < jb.co_t) {
      const int co = cot * jb.co_t + q;
      float sb_ = 0.f;
      for (int c = 0; c < jb.nchunks; ++c) sb_ += pbias[(int64_t)c * nb + co];
      if (co < jb.Co) dbp[co] = sb_;
    }
    return;
  }
  if (jb.nchunks <= N3D_FINAL_DIRECT_MAX) {
    // few chunks, large tiles (1x1x1 convs: the tile is the whole matrix): 256 positions per workgroup, all rows of a position requested up front
    const int p = lb * 256 + threadIdx.x;
    float s = 0.f;
    if (p < nslab) {
      if (!has_dw) return;
      s = jb.nchunks <= 4 ? final_direct<4>(partial, jb.nchunks, cs, p) : final_direct<N3D_FINAL_DIRECT_MAX>(partial, jb.nchunks, cs, p);
    } else if (p < nslab + nb && has_db) {
      for (int c = 0; c < jb.nchunks; ++c) s += pbias[(int64_t)c * nb + (p - nslab)];
    }
    final_store(jb, dwp, dbp, p, nslab, nb, T, s);
    return;
  }
  // many chunks: P positions x S chunk segments per workgroup, segments combined in LDS
  const int pl = final_pos_log2(jb.nchunks), P = 1 << pl, S = 256 >> pl;
  const int oi = threadIdx.x & (P - 1), sg = threadIdx.x >> pl;
  const int p = lb * P + oi;
  float s = 0.f;
  if (p < nslab) {
    if (has_dw) {
      // eight slab rows in flight per step (a rolled load -> add loop pays one memory latency per row)
      int c = sg;
      for (; c + 7 * S < jb.nchunks; c += 8 * S) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = partial[(c + S * u) * cs + p];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; c < jb.nchunks; c += S) s += partial[c * cs + p];
    }
  } else if (p < nslab + nb && has_db) {
    for (int c = sg; c < jb.nchunks; c += S) s += pbias[(int64_t)c * nb + (p - nslab)];
  }
  seg[sg * P + oi] = s;
  __syncthreads();
  if (sg == 0) {
    float tot = 0.f;
    for (int k = 0; k < S; ++k) tot += seg[k * P + oi];
    final_store(jb, dwp, dbp, p, nslab, nb, T, tot);
  }
}

// job grouping: the pointers of a group are sorted and cut into <= 8 segments of < 2 GB; a launch takes as many consecutive
// jobs as the table holds and the segments cover (one job alone always fits: it has at most four pointers)
struct SegTable {
  SegBases sb; int n = 0;
  static constexpr uintptr_t SPAN = ((uintptr_t)1 << 29) * sizeof(float);
  bool build(uintptr_t* ptrs, int np) {
    std::sort(ptrs, ptrs + np);
    n = 0;
    for (int i = 0; i < np; ++i) {
      if (!ptrs[i]) continue;
      if (n == 0 || ptrs[i] - sb.b[n - 1] >= SPAN - sizeof(float)) {
        if (n == 8) return false;
        sb.b[n++] = ptrs[i];
      }
    }
    for (int i = n; i < 8; ++i) sb.b[i] = n ? sb.b[0] : 0;
    return true;
  }
  uint32_t enc(const void* p) const {
    if (!p) return N3D_NO_OFF;
    const uintptr_t a = (uintptr_t)p;
    int k = 0;
    while (k + 1 < n && sb.b[k + 1] <= a) ++k;
    return ((uint32_t)k << 29) | (uint32_t)((a - sb.b[k]) / sizeof(float));
  }
};

template <int NP, typename Job, typename GetPtrs>
static int seg_group(const Job* jobs, int remaining, int cap, SegTable& st, GetPtrs get) {
  int n = remaining < cap ? remaining : cap;
  std::vector<uintptr_t> ptrs;
  for (;; n = (n + 1) / 2) {
    ptrs.clear();
    for (int i = 0; i < n; ++i) { const void* q[NP]; get(jobs[i], q); for (int k = 0; k < NP; ++k) ptrs.push_back((uintptr_t)q[k]); }
    if (st.build(ptrs.data(), (int)ptrs.size()) || n == 1) return n;
  }
}

// workgroup -> job byte map: job i takes blocks[i] workgroups rounded up to whole map entries; returns how many of the n
// jobs fit the map (>= 1: a single job larger than the map is named by every entry), fills start[] and *nblk
static int unit_map(const int* blocks, int n, int gshift, uint32_t* unit, uint32_t* start, int* nblk) {
  const int G = 1 << gshift;
  uint8_t* u8 = reinterpret_cast<uint8_t*>(unit);
  memset(u8, 0, N3D_JOB_UNITS);
  int units = 0, m = 0;
  for (; m < n; ++m) {
    const int uj = (blocks[m] + G - 1) >> gshift;
    if (units + uj > N3D_JOB_UNITS) break;
    start[m] = (uint32_t)units << gshift;
    for (int k = 0; k < uj; ++k) u8[units + k] = (uint8_t)m;
    units += uj;
  }
  if (m == 0) { start[0] = 0; *nblk = blocks[0]; return 1; }
  *nblk = units << gshift;
  return m;
}

// ------------------------------------------------------------------------------------------------
// host-side launch helpers shared with the C ABI
// ------------------------------------------------------------------------------------------------
static int wgrad_tiles(int Ci, int Co, int* ci_t, int* co_t) {
  *ci_t = (Ci % 8 == 0) ? 8 : 4;
  *co_t = (Co % 16 == 0) ? 16 : (Co % 8 == 0 ? 8 : 4);
  return 0;
}

struct WgradPlan { int ci_t, co_t, tci, tco, ntiles, nchunks; int64_t chunk; size_t partial_floats, pbias_floats; };

static WgradPlan wgrad_plan(int B, int64_t No, int Ci, int Co, int taps) {
  WgradPlan p;
  wgrad_tiles(Ci, Co, &p.ci_t, &p.co_t);
  p.tci = Ci / p.ci_t;
  p.tco = (int)cdiv(Co, p.co_t);
  p.ntiles = taps * p.tci * p.tco;
  const int64_t total = (int64_t)B * No;
  // short chunks: the voxel loop of the kernel is a dependent load -> FMA chain (one memory latency per
  // iteration), so per-thread trip count, not block count, sets the kernel time on these small problems
  constexpr int chunk0 = 512, max_wgs = 8192;
  int64_t nch = cdiv(total, chunk0);
  // keep the grid around a few thousand blocks
  while (nch * p.ntiles > max_wgs && nch > 1) nch = (nch + 1) / 2;
  if (nch > 512) nch = 512;
  p.chunk = cdiv(total, nch);
  p.chunk = cdiv(p.chunk, 256) * 256;
  p.nchunks = (int)cdiv(total, p.chunk);
  p.partial_floats = (size_t)p.nchunks * p.ntiles * p.ci_t * p.co_t;
  p.pbias_floats = (size_t)p.nchunks * p.tco * p.co_t;
  return p;
}

template <int CI_T, int CO_T>
static void launch_wgrad_t(const WgradArgs& a, const WgradPlan& p, hipStream_t s) {
  const bool sb = a.flags & N3D_SRC_BF16, db = a.flags & N3D_DST_BF16;
  const dim3 grid(p.nchunks, p.ntiles);
  if (sb && db) hipLaunchKernelGGL((conv_wgrad_kernel<CI_T, CO_T, bf16_t, bf16_t>), grid, dim3(256), 0, s, a);
  else if (sb) hipLaunchKernelGGL((conv_wgrad_kernel<CI_T, CO_T, bf16_t, float>), grid, dim3(256), 0, s, a);
  else if (db) hipLaunchKernelGGL((conv_wgrad_kernel<CI_T, CO_T, float, bf16_t>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<CI_T, CO_T, float, float>), grid, dim3(256), 0, s, a);
}

}  // namespace n3d

using namespace n3d;

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
// 4-channel groups of a tensor are loaded with one access: 16-byte aligned for fp32, 8-byte for bf16 storage
static bool aligned_quad(const void* p, bool bf16) { return (reinterpret_cast<uintptr_t>(p) & (bf16 ? 7 : 15)) == 0; }
#define N3D_ANY_BF16 (N3D_SRC_BF16 | N3D_DST_BF16)

static size_t packed_floats(const n3d_conv_geom* g) {
  const int taps = g->k * g->k * g->k;
  const int cmax = g->Ci > g->Co ? g->Ci : g->Co;
  const int cpad = (int)align_up(cmax, 16);
  return (size_t)taps * cmax * cpad;
}

static int check_geom(const n3d_conv_geom* g, const char* who) {
  if (!g) { set_error("%s: null geometry", who); return N3D_ERR_INVALID; }
  if (g->B <= 0 || g->Di <= 0 || g->Hi <= 0 || g->Wi <= 0 || g->Ci <= 0 || g->Do <= 0 || g->Ho <= 0 || g->Wo <= 0 || g->Co <= 0) {
    set_error("%s: non-positive dimension", who); return N3D_ERR_INVALID;
  }
  if (!(g->k == 1 || g->k == 3) || !(g->stride == 1 || g->stride == 2) || !(g->dil == 1 || g->dil == 2) || g->pad < 0) {
    set_error("%s: unsupported k=%d stride=%d dil=%d", who, g->k, g->stride, g->dil); return N3D_ERR_UNSUPPORTED;
  }
  auto odim = [&](int i) { return (i + 2 * g->pad - g->dil * (g->k - 1) - 1) / g->stride + 1; };
  // for transposed use the o side is smaller by construction; accept any o with odim(i) == o
  if (odim(g->Di) != g->Do || odim(g->Hi) != g->Ho || odim(g->Wi) != g->Wo) {
    set_error("%s: geometry mismatch: i=(%d,%d,%d) o=(%d,%d,%d) k=%d s=%d d=%d p=%d", who, g->Di, g->Hi, g->Wi, g->Do, g->Ho, g->Wo, g->k,
              g->stride, g->dil, g->pad);
    return N3D_ERR_INVALID;
  }
  if (g->depthwise && (g->Ci != g->Co || g->Ci % 4 != 0 || g->k != 3 || g->dil != 1)) {
    set_error("%s: depthwise needs Ci == Co, C %% 4 == 0, k=3, dil=1", who); return N3D_ERR_UNSUPPORTED;
  }
  return 0;
}

namespace n3d {
// implemented in conv_mfma.hip: returns 1 if it handled the call, 0 to fall through, <0 on error
int mfma_conv_try(const n3d_conv_geom* g, bool data_grad, const float* src, int64_t sld, const float* w, const float* bias, float* dst,
                  int64_t dld, int flags, const float* in_gate, const float* relu_src, int64_t rld, const float* out_gate, double* stats,
                  void* ws, size_t ws_bytes, hipStream_t s);
int mfma_conv_stats_rows(const n3d_conv_geom* g, bool data_grad, int flags);
// bf16-storage vox64 family (conv_bf16.hip)
int vox16_layout(const n3d_conv_geom* g, bool data_grad, int flags);
int vox16_stats_rows(const n3d_conv_geom* g, bool data_grad, int flags);
int vox16_conv_try(const n3d_conv_geom* g, bool data_grad, const void* src, int64_t sld, const float* w, const float* bias, void* dst,
                   int64_t dld, int flags, const float* in_gate, const void* relu_src, const float* out_gate, double* stats, void* ws,
                   size_t ws_bytes, hipStream_t s);
int mfma_pack_layout(const n3d_conv_geom* g, bool data_grad, int flags);  // 1 gemm16, 2 vox64, 0 none
int vox_wgrad_try(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, int flags, const float* in_gate,
                  float* partial, size_t avail_floats, int* nchunks_out, hipStream_t s);
int vox_wgrad_s2_try(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, int flags, const float* in_gate,
                     float* partial, size_t avail_floats, int* nchunks_out, hipStream_t s);
struct Wg16Args;
int wgrad_tile16_try(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, int flags, const float* in_gate,
                     float* partial, size_t avail_floats, int* nchunks_out, float** pbias_out, hipStream_t s);
bool wgrad_tile16_plan(const n3d_conv_geom* g, int* chunks, int* tiles_per_wg, int* tw_out = nullptr);
int mfma_wgrad_try(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, int flags, const float* in_gate,
                   float* partial, float* pbias, size_t avail_floats, int* nchunks_out, int* ntiles_out, hipStream_t s,
                   Wg16Args* prepared = nullptr);
struct DualArgs;
int mfma_bwd_dual_try(const n3d_conv_geom* g, bool transposed, const float* dy, int64_t dyld, const float* wp_packed, float* dx,
                      int64_t dxld, int flags_d, const float* relu_src, int64_t rld, const float* out_gate, const float* x, int64_t xld,
                      int flags_w, const float* in_gate, float* partial, float* pbias, size_t avail_floats, int* nchunks_out,
                      int* ntiles_out, hipStream_t s, DualArgs* prepared = nullptr, int* ksplit_out = nullptr);
int mfma_bwd_quad_ok(const n3d_conv_geom* g0, bool t0, const n3d_conv_geom* g1, bool t1);
struct BwdOne {  // one conv's backward operands for the quad launch (same layout as in conv_mfma.hip)
  const n3d_conv_geom* g; bool transposed; const float* dy; int64_t dyld; const float* wp; float* dx; int64_t dxld; int flags_d;
  const float* relu_src; int64_t rld; const float* out_gate; const float* x; int64_t xld; int flags_w; const float* in_gate;
  float* partial; float* pbias; size_t avail; int nch, ntl;
};
int mfma_bwd_quad_try(BwdOne* c0, BwdOne* c1, hipStream_t s);
int mfma_conv_multi_try(int n, const n3d_conv_geom* const* g, const bool* dg, const float* const* src, const int64_t* sld, const float* const* w,
                        const float* const* bias, float* const* dst, const int64_t* dld, const int* flags, const float* const* gate,
                        double* const* stats, void* const* ws, const size_t* wsb, hipStream_t s);
struct VoxCall {  // one conv of a multi-conv launch of the vox family (same layout as in conv_mfma.hip)
  const n3d_conv_geom* g; bool data_grad; const float* src; int64_t sld; const float* w; const float* bias; float* dst; int64_t dld; int flags;
  const float* in_gate; const float* relu_src; const float* out_gate; double* stats; void* ws; size_t ws_bytes;
};
int mfma_vox_multi_try(int n, const VoxCall* c, hipStream_t s);
// forward call -> gather operands (forward conv = gather with data_grad=false, transposed forward = data_grad=true: run_gather's convention)
static VoxCall vox_call_fwd(const n3d_conv_fwd_call* c) {
  return VoxCall{c->g, c->transposed != 0, c->x, c->xld, c->w, c->bias, c->y, c->yld, c->flags, c->in_gate, nullptr, nullptr, c->stats, c->ws, c->ws_bytes};
}
int mfma_conv_pair_try(const n3d_conv_geom* g0, bool dg0, const float* src0, int64_t sld0, const float* w0, const float* bias0, float* dst0,
                       int64_t dld0, int flags0, const float* gate0, double* stats0, void* ws0, size_t wsb0, const n3d_conv_geom* g1, bool dg1,
                       const float* src1, int64_t sld1, const float* w1, const float* bias1, float* dst1, int64_t dld1, int flags1,
                       const float* gate1, double* stats1, void* ws1, size_t wsb1, hipStream_t s, const PairExtras* x0, const PairExtras* x1);
void mfma_pack16(const float* w, float* wp, int Co, int Ci, int taps, int data_grad, hipStream_t s);
}

// request to fold the data gradient into the weight-gradient launch (n3d_conv_bwd_both)
struct DualReq {
  const float* w; float* dx; int64_t dxld; int flags; const float* relu_src; int64_t rld; const float* out_gate;
  void* ws; size_t ws_bytes; bool done;
};

extern "C" {

size_t n3d_conv_workspace_bytes(const n3d_conv_geom* g) {
  if (!g) return 0;
  const int taps = g->k * g->k * g->k;
  size_t bytes = align_up(packed_floats(g) * 4, 256);
  // weight-gradient partial slabs (dense) or depthwise partials
  const int64_t No = (int64_t)g->Do * g->Ho * g->Wo, Ni = (int64_t)g->Di * g->Hi * g->Wi;
  if (g->depthwise) {
    bytes += align_up((size_t)1024 * 28 * g->Ci * 4, 256);
  } else {
    WgradPlan p1 = wgrad_plan(g->B, No, g->Ci, g->Co, taps);
    WgradPlan p2 = wgrad_plan(g->B, No, g->Co, g->Ci, taps);  // transposed roles
    size_t a = (p1.partial_floats + p1.pbias_floats) * 4, b = (p2.partial_floats + p2.pbias_floats) * 4;
    if (g->k == 3 && g->Ci == g->Co && (g->Ci == 4 || g->Ci == 8)) {  // vox64 weight-gradient slabs: <= 1024 workgroups
      const size_t c = (size_t)1024 * 27 * g->Ci * g->Ci * 4;
      if (c > a) a = c;
    }
    if (g->k == 3 && g->stride == 2 && g->Ci == 4 && g->Co % 4 == 0 && g->Co > 4 && g->Co <= 16) {  // the stems' 4 -> 12 conv, co-tiled (vox_wgrad_s2_try)
      const size_t c = (size_t)1024 * 27 * g->Ci * g->Co * 4;
      if (c > a) a = c;
    }
    if (g->Ci % 16 == 0 && g->Co % 16 == 0) {  // MFMA weight-gradient slabs (conv_mfma.hip)
      const size_t nt = (size_t)taps * (g->Ci / 16) * (g->Co / 16);
      const size_t c = (N3D_WG16_SLABS + nt) * (256 + 16) * 4;
      if (c > a) a = c;
      int chunks = 0, tpw = 0;
      if (wgrad_tile16_plan(g, &chunks, &tpw)) {   // LDS-tile weight gradient: one slab set per chunk of tiles
        const size_t c2 = ((size_t)chunks * nt * 256 + (size_t)chunks * (g->Co / 16) * 16) * 4;
        if (c2 > a) a = c2;
      }
    }
    bytes += align_up(a > b ? a : b, 256) + 256;
  }
  (void)Ni;
  return bytes;
}

// number of partial-statistics rows per sample the forward (transposed=0) or transposed-forward
// (transposed=1) kernel writes into `stats` ([B][rows][Cout][2] doubles)
int n3d_conv_stats_rows(const n3d_conv_geom* g, int transposed, int flags) {
  if (!g) return 0;
  if (int r16 = vox16_stats_rows(g, transposed != 0, flags)) return r16;
  int r = mfma_conv_stats_rows(g, transposed != 0, flags);
  if (r > 0) return r;
  if (r < 0) return 0;  // the kernel for this shape cannot emit statistics: use n3d_channel_stats
  const int64_t Nd = transposed ? (int64_t)g->Di * g->Hi * g->Wi : (int64_t)g->Do * g->Ho * g->Wo;
  // transposed forward runs the gather kernel with den = stride; its parity-class mode has 8 row groups
  // (activation tensors on this path are 16-byte aligned pitched views, which the class mode requires)
  if (transposed && gather_class_mode(g->stride, g->k, g->Di, g->Hi, g->Wi, g->Co, true)) return (int)(8 * cdiv(Nd / 8, 256));
  if (g->stride == 1 && k1_shape_ok(g, transposed != 0)) return (int)cdiv(Nd, K1_VPB);   // the 1x1x1 streaming kernel: one row per 1024 voxels
  return (int)cdiv(Nd, 256);
}

static DwArgs dw_args(const n3d_conv_geom* g, bool data_grad, const float* src, int64_t sld, const float* w, const float* bias, float* dst,
                      int64_t dld, int flags) {
  DwArgs a;
  a.C = g->Ci; a.w = w; a.bias = bias; a.k = g->k; a.flags = flags;
  if (!data_grad) { a.src = src; a.sld = sld; a.Ds = g->Di; a.Hs = g->Hi; a.Ws = g->Wi; a.dst = dst; a.dld = dld; a.Dd = g->Do; a.Hd = g->Ho; a.Wd = g->Wo;
    a.sn = g->stride; a.off = -g->pad; a.dt = g->dil; a.den = 1; }
  else { a.src = src; a.sld = sld; a.Ds = g->Do; a.Hs = g->Ho; a.Ws = g->Wo; a.dst = dst; a.dld = dld; a.Dd = g->Di; a.Hd = g->Hi; a.Wd = g->Wi;
    a.sn = 1; a.off = g->pad; a.dt = -g->dil; a.den = g->stride; }
  a.fcpb = FastDiv((uint32_t)(a.C / 4)); a.fWd = FastDiv((uint32_t)a.Wd); a.fHd = FastDiv((uint32_t)a.Hd);
  return a;
}

// extras of n3d_conv_k1_norm_fwd for the 1x1x1 streaming kernel (set around its run_gather call on the calling thread)
struct K1Norm { const float* oscale; const float* oshift; bool nostore; bool used; };
static thread_local K1Norm* g_k1_norm = nullptr;

static int run_gather(const n3d_conv_geom* g, bool data_grad, const float* src, int64_t sld, const float* w, const float* bias, float* dst,
                      int64_t dld, int flags, const float* in_gate, const float* relu_src, int64_t rld, const float* out_gate,
                      double* stats, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const bool sb16 = flags & N3D_SRC_BF16, db16 = flags & N3D_DST_BF16;
  if (g->depthwise) {
    N3D_CHECK_ARG(!in_gate && !relu_src && !out_gate && !stats && !(flags & N3D_RELU_IN), "depthwise conv: gate/relu/stats not supported");
    N3D_CHECK_ARG(sld % 4 == 0 && dld % 4 == 0 && aligned_quad(src, sb16) && aligned_quad(dst, db16), "depthwise conv: needs quad-aligned pitched rows");
    DwArgs a = dw_args(g, data_grad, src, sld, w, bias, dst, dld, flags);
    const int64_t Nd = (int64_t)a.Dd * a.Hd * a.Wd;
    const dim3 grid((unsigned)cdiv(Nd * (a.C / 4), 256), g->B);
    const size_t shm = (size_t)(a.C / 4) * 27 * sizeof(float4);
    // storage types of source / destination (round 5: bf16 storage of the C <= 8 cells also with depthwise primitives)
    if (sb16 && db16) hipLaunchKernelGGL((dw_gather_kernel<bf16_t, bf16_t>), grid, dim3(256), shm, s, a);
    else if (sb16) hipLaunchKernelGGL((dw_gather_kernel<bf16_t, float>), grid, dim3(256), shm, s, a);
    else if (db16) hipLaunchKernelGGL((dw_gather_kernel<float, bf16_t>), grid, dim3(256), shm, s, a);
    else hipLaunchKernelGGL((dw_gather_kernel<float, float>), grid, dim3(256), shm, s, a);
    N3D_LAUNCH_CHECK();
    return N3D_OK;
  }
  // node-planar operands (a pitch smaller than the channel count; include/n3d.h): only the 1x1x1 streaming kernel reads / writes them
  const int Cs_all = data_grad ? g->Co : g->Ci, Cd_all = data_grad ? g->Ci : g->Co;
  const bool src_planar = sld > 0 && sld < Cs_all, dst_planar = dst && dld > 0 && dld < Cd_all;     // (dst NULL: the statistics-only pass of n3d_conv_k1_norm_fwd)
  if (src_planar || dst_planar) {
    const bool ok = g->k == 1 && g->stride == 1 && sld >= 4 && dld >= 4 && sld % 4 == 0 && dld % 4 == 0 && Cs_all % sld == 0 && Cd_all % dld == 0 &&
                    !in_gate && !out_gate && (!dst_planar || (data_grad && !bias && !stats && (!relu_src || rld == dld))) &&
                    k1_dims_ok(g, data_grad, Cs_all, dst_planar ? (int)dld : Cd_all);
    if (!ok) N3D_UNSUPPORTED("conv: node-planar operands (pitch < channels) are taken by the 1x1x1 streaming kernel only (stride 1, no gates, "
                             "<= 24 source / <= 12 destination channels per launch, >= 32768 voxels)");
  }
  if (!(flags & N3D_NO_MFMA) && !src_planar && !dst_planar) {
    int r = mfma_conv_try(g, data_grad, src, sld, w, bias, dst, dld, flags, in_gate, relu_src, rld, out_gate, stats, ws, ws_bytes, s);
    if (r != 0) return r < 0 ? r : N3D_OK;
  }
  if (sb16 && db16 && !src_planar && !dst_planar) {
    int r = vox16_conv_try(g, data_grad, src, sld, w, bias, dst, dld, flags, in_gate, relu_src, out_gate, stats, ws, ws_bytes, s);
    if (r != 0) return r < 0 ? r : N3D_OK;
  }
  const int taps = g->k * g->k * g->k;
  GatherArgs a;
  a.bias = bias; a.k = g->k; a.flags = flags; a.in_gate = in_gate; a.relu_src = relu_src; a.rld = rld; a.out_gate = out_gate; a.stats = stats;
  a.src = src; a.sld = sld; a.dst = dst; a.dld = dld;
  if (!data_grad) { a.Ds = g->Di; a.Hs = g->Hi; a.Ws = g->Wi; a.Cs = g->Ci; a.Dd = g->Do; a.Hd = g->Ho; a.Wd = g->Wo; a.Cd = g->Co;
    a.sn = g->stride; a.off = -g->pad; a.dt = g->dil; a.den = 1; }
  else { a.Ds = g->Do; a.Hs = g->Ho; a.Ws = g->Wo; a.Cs = g->Co; a.Dd = g->Di; a.Hd = g->Hi; a.Wd = g->Wi; a.Cd = g->Ci;
    a.sn = 1; a.off = g->pad; a.dt = -g->dil; a.den = g->stride; }
  a.fWd = FastDiv((uint32_t)a.Wd); a.fHd = FastDiv((uint32_t)a.Hd);
  const int cot = pick_cot(a.Cd);
  a.Cdp = (int)align_up(a.Cd, cot);
  const size_t need = (size_t)taps * a.Cs * a.Cdp * 4;
  if (!ws || ws_bytes < need) { set_error("conv: workspace too small (%zu < %zu)", ws_bytes, need); return N3D_ERR_WORKSPACE; }
  float* wp = (float*)ws;
  a.wp = wp;
  const int total = taps * a.Cs * a.Cdp;
  if (!(flags & N3D_PREPACKED))
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, wp, g->Co, g->Ci, taps, a.Cdp, data_grad ? 1 : 0);
  if (stats && gather_class_mode(a.den, a.k, a.Dd, a.Hd, a.Wd, a.Cs, true) && !((a.Cs % 4 == 0) && (a.sld % 4 == 0) && aligned_quad(a.src, sb16))) {
    set_error("conv: statistics on this shape need a 16-byte aligned source (n3d_conv_stats_rows assumed the parity-class kernel)");
    return N3D_ERR_UNSUPPORTED;
  }
  const int cd_eff = dst_planar ? (int)dld : a.Cd;       // destination channels of one launch slice (blockIdx.z = node)
  if (k1_dims_ok(g, data_grad, a.Cs, cd_eff)) {
    const bool nostore = g_k1_norm && g_k1_norm->nostore;
    const bool fits = !in_gate && !out_gate && sld % 4 == 0 && (nostore || (dld % 4 == 0 && aligned_quad(dst, db16))) && aligned_quad(src, sb16) && aligned16(a.wp) &&
                      (!bias || aligned16(bias)) && (!relu_src || (rld % 4 == 0 && aligned_quad(relu_src, db16))) && a.Cdp % 4 == 0 &&
                      !(a.den == 2 && (bias || stats));   // the zero-upsampling form carries neither
    if (fits) {
      K1Args q;
      q.src = src; q.sld = sld; q.dst = dst; q.dld = dld; q.wp = a.wp; q.Cdp = a.Cdp; q.bias = bias; q.flags = flags; q.relu_src = relu_src;
      q.rld = rld; q.stats = stats; q.N = (int64_t)a.Dd * a.Hd * a.Wd;
      q.up = a.den == 2 ? 1 : 0; q.Wd = a.Wd; q.Hd = a.Hd; q.Ns = (int64_t)a.Ds * a.Hs * a.Ws; q.fWd = a.fWd; q.fHd = a.fHd;
      {
        constexpr bool noflat = false;
        const size_t esz = db16 ? 2 : 4;
        // fp32 destinations only: measured at 2 x 128^3, 12-channel writes 54.5 -> 42.5 us (fp32) but 31.1 -> 35.2 us (bf16: the 8-byte LDS
        // stores on a 24-byte pitch cost more than the partial-line stores they replace)
        q.flat = (!noflat && !db16 && cd_eff >= 8 && dld == cd_eff && aligned16(dst) && ((size_t)q.N * cd_eff * esz) % 16 == 0) ? 1 : 0;
      }
      {
        constexpr bool nosparse = false;
        q.nostore = 0; q.oscale = q.oshift = nullptr;
        if (g_k1_norm) { q.nostore = g_k1_norm->nostore ? 1 : 0; q.oscale = g_k1_norm->oscale; q.oshift = g_k1_norm->oshift; g_k1_norm->used = true; if (q.nostore) q.flat = 0; }
        q.sparse = (q.up && (flags & N3D_ACCUMULATE) && !nosparse) ? 1 : 0;
        q.fWs = FastDiv((uint32_t)(a.Wd >> 1)); q.fHs = FastDiv((uint32_t)(a.Hd >> 1));
        if (q.sparse) q.flat = 0;
      }
      const bool extra = (flags & N3D_ACCUMULATE) || relu_src || q.up;
      N3D_CHECK_ARG(!(extra && (stats || q.oscale || q.nostore)), "conv(1x1x1): statistics / output coefficients together with accumulate / relu mask are not supported");
      // node-planar operands: node k of a tensor starts k * (B * voxels * pitch) elements behind node 0
      q.src_node_stride = src_planar ? (int64_t)g->B * q.N * sld : 0;
      q.dst_node_stride = dst_planar ? (int64_t)g->B * q.N * dld : 0;
      q.relu_node_stride = (dst_planar && relu_src) ? (int64_t)g->B * q.N * rld : 0;
      const dim3 grid((unsigned)cdiv(q.sparse ? q.Ns : q.N, extra ? 512 : K1_VPB), (unsigned)g->B, (unsigned)(a.Cd / cd_eff));
      switch (a.Cs / 4) {
        case 1: launch_k1_c<1>(q, cd_eff, grid, s); break;
        case 2: launch_k1_c<2>(q, cd_eff, grid, s); break;
        case 3: launch_k1_c<3>(q, cd_eff, grid, s); break;
        case 4: launch_k1_c<4>(q, cd_eff, grid, s); break;
        case 5: launch_k1_c<5>(q, cd_eff, grid, s); break;
        default: launch_k1_c<6>(q, cd_eff, grid, s); break;
      }
      N3D_LAUNCH_CHECK();
      return N3D_OK;
    }
    if (stats && g->stride == 1) {
      set_error("conv: statistics on this 1x1x1 shape need the streaming kernel (no gates, 16-byte aligned rows): n3d_conv_stats_rows assumed it");
      return N3D_ERR_UNSUPPORTED;
    }
  }
  launch_gather(a, g->B, s);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_conv_fwd(const n3d_conv_geom* g, const float* x, int64_t xld, const float* w, const float* bias, float* y, int64_t yld, int flags,
                 const float* in_gate, double* stats, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_geom(g, "conv_fwd")) return e;
  N3D_CHECK_ARG(x && w && y && (xld >= g->Ci || (xld >= 4 && g->Ci % xld == 0)) && yld >= g->Co, "conv_fwd: bad pointers/pitches");
  return run_gather(g, false, x, xld, w, bias, y, yld, flags, in_gate, nullptr, 0, nullptr, stats, ws, ws_bytes, stream);
}

static void fill_job(n3d_final_job* j, const float* partial, const float* pbias, float* dw, float* dbias, int nch, int ntl, int tci, int tco,
                     int ci_t, int co_t, int Co, int Ci, int taps);

int n3d_conv_k1_norm_ok(const n3d_conv_geom* g) { return g && check_geom(g, "conv_k1_norm_ok") == 0 && k1n_shape_ok(g) && k1_shape_ok(g, false) ? 1 : 0; }
int n3d_conv_k1_norm_rows(const n3d_conv_geom* g) { return g ? (int)cdiv((int64_t)g->Do * g->Ho * g->Wo, k1n_chunk((int64_t)g->Do * g->Ho * g->Wo)) : 0; }

int n3d_conv_k1_norm_fwd(const n3d_conv_geom* g, const float* x, int64_t xld, const float* w, const float* bias, float* y, int64_t yld,
                         int flags, const float* oscale, const float* oshift, double* stats, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_geom(g, "conv_k1_norm_fwd")) return e;
  N3D_CHECK_ARG(x && w && xld >= g->Ci && (y ? yld >= g->Co : stats != nullptr), "conv_k1_norm_fwd: bad pointers / pitches (y == NULL needs stats)");
  N3D_CHECK_ARG((oscale == nullptr) == (oshift == nullptr) && !(stats && oscale), "conv_k1_norm_fwd: oscale and oshift come together, without statistics");
  if (!n3d_conv_k1_norm_ok(g)) N3D_UNSUPPORTED("conv_k1_norm_fwd: 1x1x1 stride-1 convs with (Ci, Co) in {(4,4), (4,8), (4,12), (8,4)} on >= 32768 voxels");
  K1Norm nx = {oscale, oshift, y == nullptr, false};
  g_k1_norm = &nx;
  const int r = run_gather(g, false, x, xld, w, bias, y, yld, flags | N3D_NO_MFMA, nullptr, nullptr, 0, nullptr, stats, ws, ws_bytes, stream);
  g_k1_norm = nullptr;
  if (r) return r;
  if (!nx.used) N3D_UNSUPPORTED("conv_k1_norm_fwd: the operands do not fit the 1x1x1 streaming kernel (alignment / pitches)");
  return N3D_OK;
}

int n3d_conv_k1_norm_bwd_reduce(const n3d_conv_geom* g, const void* x, int64_t xld, const float* w, const float* bias, const void* dout,
                                int64_t dld, const float* a, const float* b, int flags, double* sums, void* stream) {
  if (int e = check_geom(g, "conv_k1_norm_bwd_reduce")) return e;
  N3D_CHECK_ARG(x && w && dout && a && b && sums && xld >= g->Ci && dld >= g->Co && xld % 4 == 0 && dld % 4 == 0, "conv_k1_norm_bwd_reduce: bad args");
  if (!k1n_shape_ok(g)) N3D_UNSUPPORTED("conv_k1_norm_bwd_reduce: shape");
  K1nArgs q = {};
  q.x = x; q.xld = xld; q.dout = dout; q.dld = dld; q.w = w; q.bias = bias; q.a = a; q.b = b; q.sums = sums;
  q.N = (int64_t)g->Do * g->Ho * g->Wo; q.chunk = k1n_chunk(q.N); q.relu = (flags & N3D_RELU) ? 1 : 0;
  if (!launch_k1n<false>(q, g->Ci, g->Co, flags, dim3((unsigned)cdiv(q.N, q.chunk), (unsigned)g->B), (hipStream_t)stream)) N3D_UNSUPPORTED("conv_k1_norm_bwd_reduce: channels");
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_conv_k1_norm_bwd_apply_wgrad(const n3d_conv_geom* g, const void* x, int64_t xld, const float* w, const float* bias, const void* dout,
                                     int64_t dld, const float* a, const float* b, const float* A, const float* Bc, const float* Cc, int flags,
                                     float* dw, void* ws, size_t ws_bytes, n3d_final_job* deferred, void* stream) {
  if (deferred) deferred->nchunks = 0;
  if (int e = check_geom(g, "conv_k1_norm_bwd_apply_wgrad")) return e;
  N3D_CHECK_ARG(x && w && dout && a && b && A && Bc && Cc && dw && ws && xld >= g->Ci && dld >= g->Co && xld % 4 == 0 && dld % 4 == 0,
                "conv_k1_norm_bwd_apply_wgrad: bad args");
  if (!k1n_shape_ok(g)) N3D_UNSUPPORTED("conv_k1_norm_bwd_apply_wgrad: shape");
  K1nArgs q = {};
  q.x = x; q.xld = xld; q.dout = dout; q.dld = dld; q.w = w; q.bias = bias; q.a = a; q.b = b; q.A = A; q.Bc = Bc; q.Cc = Cc;
  q.N = (int64_t)g->Do * g->Ho * g->Wo; q.chunk = k1n_chunk(q.N); q.relu = (flags & N3D_RELU) ? 1 : 0;
  const int rows = (int)cdiv(q.N, q.chunk), nchunks = rows * g->B;
  const size_t need = (size_t)nchunks * g->Ci * g->Co * sizeof(float);
  if (ws_bytes < need) { set_error("conv_k1_norm_bwd_apply_wgrad: workspace too small (%zu < %zu)", ws_bytes, need); return N3D_ERR_WORKSPACE; }
  q.partial = (float*)ws;
  if (!launch_k1n<true>(q, g->Ci, g->Co, flags, dim3((unsigned)rows, (unsigned)g->B), (hipStream_t)stream)) N3D_UNSUPPORTED("conv_k1_norm_bwd_apply_wgrad: channels");
  N3D_LAUNCH_CHECK();
  // one "tile" holding the whole [Ci][Co] slab per workgroup: ci_t = Ci, co_t = Co (as the 1x1x1 weight-gradient kernel)
  n3d_final_job job;
  fill_job(deferred ? deferred : &job, q.partial, nullptr, dw, nullptr, nchunks, 1, 1, 1, g->Ci, g->Co, g->Co, g->Ci, 1);
  if (!deferred) if (int e = n3d_wgrad_finalize_batch(&job, 1, stream)) return e;
  return N3D_OK;
}

int n3d_conv_bwd_data(const n3d_conv_geom* g, const float* dy, int64_t dyld, const float* w, float* dx, int64_t dxld, int flags,
                      const float* relu_src, int64_t rld, const float* out_gate, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_geom(g, "conv_bwd_data")) return e;
  N3D_CHECK_ARG(dy && w && dx && dyld >= g->Co && (dxld >= g->Ci || (dxld >= 4 && g->Ci % dxld == 0)), "conv_bwd_data: bad pointers/pitches");
  return run_gather(g, true, dy, dyld, w, nullptr, dx, dxld, flags & ~N3D_RELU_IN, nullptr, relu_src, rld, out_gate, nullptr, ws, ws_bytes,
                    stream);
}

int n3d_convT_fwd(const n3d_conv_geom* g, const float* x, int64_t xld, const float* w, const float* bias, float* y, int64_t yld, int flags,
                  const float* in_gate, double* stats, void* ws, size_t ws_bytes, void* stream) {
  if (int e = check_geom(g, "convT_fwd")) return e;
  N3D_CHECK_ARG(x && w && y && xld >= g->Co && yld >= g->Ci, "convT_fwd: bad pointers/pitches");
  return run_gather(g, true, x, xld, w, bias, y, yld, flags, in_gate, nullptr, 0, nullptr, stats, ws, ws_bytes, stream);
}

int n3d_convT_bwd_data(const n3d_conv_geom* g, const float* dy, int64_t dyld, const float* w, float* dx, int64_t dxld, int flags, void* ws,
                       size_t ws_bytes, void* stream) {
  if (int e = check_geom(g, "convT_bwd_data")) return e;
  N3D_CHECK_ARG(dy && w && dx && dyld >= g->Ci && dxld >= g->Co, "convT_bwd_data: bad pointers/pitches");
  return run_gather(g, false, dy, dyld, w, nullptr, dx, dxld, flags & ~N3D_RELU_IN, nullptr, nullptr, 0, nullptr, nullptr, ws, ws_bytes, stream);
}

static void fill_job(n3d_final_job* j, const float* partial, const float* pbias, float* dw, float* dbias, int nch, int ntl, int tci, int tco,
                     int ci_t, int co_t, int Co, int Ci, int taps) {
  j->partial = partial; j->pbias = pbias; j->dw = dw; j->dbias = dbias; j->nchunks = nch; j->ntiles = ntl; j->tci = tci; j->tco = tco;
  j->ci_t = ci_t; j->co_t = co_t; j->Co = Co; j->Ci = Ci; j->taps = taps; j->pad_ = 0;
}

static int run_wgrad(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, float* dw, float* dbias, int flags,
                     const float* in_gate, void* ws, size_t ws_bytes, void* stream, bool transposed, n3d_final_job* deferred,
                     DualReq* dual = nullptr) {
  if (deferred) deferred->nchunks = 0;  // 0 = nothing deferred
  hipStream_t s = (hipStream_t)stream;
  const int taps = g->k * g->k * g->k;
  const int64_t No = (int64_t)g->Do * g->Ho * g->Wo;
  const size_t skip = align_up(packed_floats(g) * 4, 256);
  if (!ws || ws_bytes <= skip) { set_error("conv_bwd_weight: workspace too small"); return N3D_ERR_WORKSPACE; }
  float* wsf = (float*)((char*)ws + skip);
  const size_t avail = (ws_bytes - skip) / 4;
  const bool sb16 = flags & N3D_SRC_BF16, db16 = flags & N3D_DST_BF16;   // storage of the kernel-role x / dy tensors
  if (g->depthwise) {
    // conv-side roles: X on the i side, DY on the o side (callers of the transposed form pass them swapped)
    N3D_CHECK_ARG(!in_gate && !(flags & N3D_RELU_IN), "depthwise wgrad: gate/relu not supported");
    DwWgradArgs a;
    a.x = x; a.xld = xld; a.Di = g->Di; a.Hi = g->Hi; a.Wi = g->Wi; a.dy = dy; a.dyld = dyld; a.Do = g->Do; a.Ho = g->Ho; a.Wo = g->Wo;
    a.B = g->B; a.C = g->Ci; a.k = 3; a.stride = g->stride; a.pad = g->pad; a.partial = wsf;
    a.fNo = FastDiv((uint32_t)No); a.fWo = FastDiv((uint32_t)g->Wo); a.fHo = FastDiv((uint32_t)g->Ho);
    const int64_t total = (int64_t)g->B * No;
    {
      // tileable stride-1 shapes with enough (tile, channel quad) units: the LDS-tile kernel
      const bool notile = flags & N3D_ANY_BF16;     // (the LDS-tile kernel fills its tile by 16-byte LDS-DMA of fp32 quads)
      const int quads = g->Ci / 4;
      if (!notile && g->k == 3 && g->stride == 1 && g->dil == 1 && g->pad == 1 && g->Wi % 16 == 0 && g->Hi % 4 == 0 && g->Di % 4 == 0 && g->Ci % 4 == 0 &&
          xld % 4 == 0 && dyld % 4 == 0 && aligned16(x) && aligned16(dy)) {
        const int tiles_per_sample = (g->Wi / 16) * (g->Hi / 4) * (g->Di / 4);
        const int64_t tiles_total = (int64_t)tiles_per_sample * g->B;
        const void* zp = dw_zero_page();
        if (tiles_total * quads >= 64 && tiles_total < (1 << 30) && zp) {
          int64_t nx = 256 / quads;
          if (nx < 1) nx = 1;
          if (nx > tiles_total) nx = tiles_total;
          const int tpw = (int)cdiv(tiles_total, nx);
          const int chunks = (int)cdiv(tiles_total, tpw);
          if ((size_t)chunks * 28 * g->Ci <= avail) {
            DwTileArgs q;
            q.x = x; q.xld = xld; q.dy = dy; q.dyld = dyld; q.D = g->Di; q.H = g->Hi; q.W = g->Wi; q.B = g->B; q.C = g->Ci;
            q.partial = wsf; q.pbias = wsf + (size_t)chunks * 27 * g->Ci;
            q.tiles_per_sample = tiles_per_sample; q.tiles_total = (int)tiles_total; q.tiles_per_wg = tpw; q.zero_page = zp;
            hipLaunchKernelGGL(dw_wgrad_tile_kernel, dim3((unsigned)chunks, (unsigned)quads), dim3(256), 0, s, q);
            n3d_final_job job;
            fill_job(&job, q.partial, q.pbias, dw, dbias, chunks, 27, 1, 1, 1, g->Ci, g->Ci, 1, 27);
            if (deferred) *deferred = job;
            else if (int e = n3d_wgrad_finalize_batch(&job, 1, stream)) return e;
            N3D_LAUNCH_CHECK();
            return N3D_OK;
          }
        }
      }
    }
    // short chunks (the voxel loop is a dependent load -> FMA chain): about two trips per thread -- a workgroup covers
    // 256 / (C/4) voxels per trip, so wide channel counts need far smaller chunks than 512 voxels; bounded by the slab workspace
    const int64_t vpb = 256 / (a.C / 4) > 0 ? 256 / (a.C / 4) : 1;
    constexpr int trips = 1;   // (2 until round 4: search step tail 0.45 -> 0.365 ms with 1)
    int64_t nch = cdiv(total, trips * vpb < 512 ? trips * vpb : 512);
    if (nch > 1024) nch = 1024;
    a.chunk = cdiv(total, nch);
    const int nchunks = (int)cdiv(total, a.chunk);
    if ((size_t)nchunks * 28 * a.C > avail) { set_error("dw wgrad: workspace too small"); return N3D_ERR_WORKSPACE; }
    a.pbias = wsf + (size_t)nchunks * 27 * a.C;
    const int cpb = a.C / 4;
    const size_t shm = (size_t)4 * cpb * 28 * 4 * sizeof(float);
    if (flags & N3D_ANY_BF16) {
      // bf16 storage (round 5: the C <= 8 cells of the bf16 configuration): both tensors bf16, 4 or 8 channels
      if (!(sb16 && db16) || (cpb != 1 && cpb != 2)) N3D_UNSUPPORTED("depthwise weight gradient: bf16 storage is built for x and dy both bf16, C = 4 / 8");
      if (cpb == 1) hipLaunchKernelGGL((dw_wgrad_kernel<1, bf16_t, bf16_t>), dim3(nchunks), dim3(256), shm, s, a);
      else hipLaunchKernelGGL((dw_wgrad_kernel<2, bf16_t, bf16_t>), dim3(nchunks), dim3(256), shm, s, a);
    } else
    switch (cpb) {
      case 1: hipLaunchKernelGGL(dw_wgrad_kernel<1>, dim3(nchunks), dim3(256), shm, s, a); break;
      case 2: hipLaunchKernelGGL(dw_wgrad_kernel<2>, dim3(nchunks), dim3(256), shm, s, a); break;
      case 4: hipLaunchKernelGGL(dw_wgrad_kernel<4>, dim3(nchunks), dim3(256), shm, s, a); break;
      case 8: hipLaunchKernelGGL(dw_wgrad_kernel<8>, dim3(nchunks), dim3(256), shm, s, a); break;
      case 16: hipLaunchKernelGGL(dw_wgrad_kernel<16>, dim3(nchunks), dim3(256), shm, s, a); break;
      default: hipLaunchKernelGGL(dw_wgrad_kernel<0>, dim3(nchunks), dim3(256), shm, s, a); break;
    }
    // fixed-order slab reduction through the common finalize: one "tile" per tap, ci_t = 1, co_t = C, Ci = 1
    n3d_final_job job;
    fill_job(&job, a.partial, a.pbias, dw, dbias, nchunks, 27, 1, 1, 1, a.C, a.C, 1, 27);
    if (deferred) *deferred = job;
    else if (int e = n3d_wgrad_finalize_batch(&job, 1, stream)) return e;
    N3D_LAUNCH_CHECK();
    return N3D_OK;
  }
  // dense: the kernel computes G[co'][ci'][tap] = sum dyK[o][co'] * xK[i(o,tap)][ci'] with xK on the i side.
  // forward conv: xK = x (Ci), dyK = dy (Co), dw native (Co,Ci,k^3) = G.
  // transposed conv (weight (CinT=Co_geom, CoutT=Ci_geom)): xK = dy_T (i side, Ci), dyK = x_T (o side, Co): same G layout.
  // (fp32 X with bf16 dY: the stem's stride-2 conv in the bf16 configuration, taken by vox_wgrad_s2_try alone)
  if ((!dbias || transposed) && (!(flags & N3D_ANY_BF16) || (sb16 && db16) || (!sb16 && db16 && !transposed && g->stride == 2 && g->Ci == 4))) {
    // vox64 weight gradient (3x3x3 stride 1, C = 4 / 8); it does not produce the bias gradient, which the callers on the
    // hot path obtain analytically from the GroupNorm backward sums (n3d_gn_bwd_coeffs)
    int nch = 0;
    int hv = vox_wgrad_try(g, x, xld, dy, dyld, flags, in_gate, wsf, avail, &nch, s);   // fp32, or both tensors in bf16 storage
    if (hv == 0) hv = vox_wgrad_s2_try(g, x, xld, dy, dyld, flags, in_gate, wsf, avail, &nch, s);
    if (hv < 0) return hv;
    if (hv == 1) {
      const int C = g->Ci, tco = g->Co / C;     // tco > 1: the stem's 4 -> 12 stride-2 conv as Co / 4 column tiles (vox_wgrad_s2_try)
      n3d_final_job job;
      fill_job(deferred ? deferred : &job, wsf, nullptr, dw, nullptr, nch, taps * tco, 1, tco, C, C, g->Co, C, taps);
      if (!deferred) if (int e = n3d_wgrad_finalize_batch(&job, 1, stream)) return e;
      N3D_LAUNCH_CHECK();
      return N3D_OK;
    }
  }
  if (!(flags & (N3D_NO_MFMA | N3D_ANY_BF16)) && xld % 4 == 0 && dyld % 4 == 0 && !transposed) {
    // many-voxel 16 -> 16 channel 3x3x3 stride-1 convs: LDS-tile MFMA weight gradient (conv_mfma.hip, wgrad_tile16)
    int nch = 0;
    float* pb = nullptr;
    const int h = wgrad_tile16_try(g, x, xld, dy, dyld, flags, in_gate, wsf, avail, &nch, &pb, s);
    if (h == 1) {
      n3d_final_job job;
      fill_job(deferred ? deferred : &job, wsf, pb, dw, dbias, nch, 27 * (g->Ci / 16) * (g->Co / 16), g->Ci / 16, g->Co / 16, 16, 16, g->Co, g->Ci, 27);
      if (!deferred) if (int e = n3d_wgrad_finalize_batch(&job, 1, stream)) return e;
      N3D_LAUNCH_CHECK();
      return N3D_OK;
    }
  }
  if (!(flags & (N3D_NO_MFMA | N3D_ANY_BF16)) && xld % 4 == 0 && dyld % 4 == 0) {
    int nch = 0, ntl = 0;
    const size_t nt16 = (size_t)taps * (g->Ci / 16) * (g->Co / 16);
    float* pb = wsf + (N3D_WG16_SLABS + nt16) * 256;
    int handled = 0;
    if (g->Ci % 16 == 0 && g->Co % 16 == 0 && (N3D_WG16_SLABS + nt16) * (256 + 16) <= avail) {
      if (dual && dual->ws && dual->ws_bytes >= (size_t)taps * g->Ci * g->Co * 4) {
        if (!(dual->flags & N3D_PREPACKED)) mfma_pack16(dual->w, (float*)dual->ws, g->Co, g->Ci, taps, transposed ? 0 : 1, s);
        // run_wgrad's (x, dy) are kernel roles (i side, o side); a transposed conv's output gradient sits on the i side
        const float* true_dy = transposed ? x : dy; const int64_t true_dyld = transposed ? xld : dyld;
        const float* true_x = transposed ? dy : x; const int64_t true_xld = transposed ? dyld : xld;
        handled = mfma_bwd_dual_try(g, transposed, true_dy, true_dyld, (const float*)dual->ws, dual->dx, dual->dxld, dual->flags,
                                    dual->relu_src, dual->rld, dual->out_gate, true_x, true_xld, flags, in_gate, wsf, pb,
                                    (N3D_WG16_SLABS + nt16) * 256, &nch, &ntl, s);
        if (handled < 0) return handled;
        if (handled == 1) dual->done = true;
      }
      if (!handled) handled = mfma_wgrad_try(g, x, xld, dy, dyld, flags, in_gate, wsf, pb, (N3D_WG16_SLABS + nt16) * 256, &nch, &ntl, s);
    }
    if (handled == 1) {
      n3d_final_job job;
      fill_job(deferred ? deferred : &job, wsf, pb, dw, transposed ? nullptr : dbias, nch, ntl, g->Ci / 16, g->Co / 16, 16, 16, g->Co, g->Ci, taps);
      if (!deferred) if (int e = n3d_wgrad_finalize_batch(&job, 1, stream)) return e;
      N3D_LAUNCH_CHECK();
      return N3D_OK;
    }
  }
  const bool x_planar = !transposed && xld < g->Ci;      // node-planar x (include/n3d.h): the 1x1x1 streaming weight gradient only
  if (x_planar && !(xld >= 4 && xld % 4 == 0 && g->Ci % xld == 0 && !in_gate && k1_wgrad_shape_ok(g) && dyld % 4 == 0 && aligned_quad(x, sb16) && aligned_quad(dy, db16)))
    N3D_UNSUPPORTED("conv_bwd_weight: a node-planar x (pitch < channels) is taken by the 1x1x1 streaming weight gradient only");
  if (!transposed && !in_gate && k1_wgrad_shape_ok(g) && xld % 4 == 0 && dyld % 4 == 0 && aligned_quad(x, sb16) && aligned_quad(dy, db16)) {
    // 1x1x1 streaming weight gradient (large levels, few channels)
    const int64_t total = (int64_t)g->B * No;
    const size_t nslab = (size_t)g->Ci * g->Co;
    // 2048 voxels per workgroup, more when the slab workspace (sized for the tiled kernel) or 1024 workgroups would be exceeded
    int64_t maxc = (int64_t)(avail / (nslab + g->Co));
    if (maxc > 1024) maxc = 1024;
    // (down to 512 on the small levels: 65 536 voxels in 2048-voxel chunks are 32 workgroups walking four trips each)
    int64_t chunk = K1W_CHUNK;
    while (chunk > 512 && cdiv(total, chunk) < 256) chunk >>= 1;
    if (maxc >= 1 && cdiv(total, chunk) > maxc) chunk = (cdiv(total, maxc) + 255) / 256 * 256;
    const int nchunks = (int)cdiv(total, chunk);
    if (maxc >= 64 && (size_t)nchunks * (nslab + g->Co) <= avail) {
      K1WgArgs q;
      q.x = x; q.xld = xld; q.dy = dy; q.dyld = dyld; q.partial = wsf; q.pbias = wsf + (size_t)nchunks * nslab; q.total = total;
      q.chunk = chunk; q.flags = flags; q.stride = g->stride; q.Wo = g->Wo; q.Ho = g->Ho; q.Wi = g->Wi; q.Hi = g->Hi; q.No = No;
      q.Ni = (int64_t)g->Di * g->Hi * g->Wi; q.x_node_stride = x_planar ? (int64_t)g->B * q.Ni * xld : 0; q.fNo = FastDiv((uint32_t)No); q.fWo = FastDiv((uint32_t)g->Wo); q.fHo = FastDiv((uint32_t)g->Ho);
      switch (g->Ci / 4) {
        case 1: launch_k1_wgrad_c<1>(q, g->Co, nchunks, s); break;
        case 2: launch_k1_wgrad_c<2>(q, g->Co, nchunks, s); break;
        case 3: launch_k1_wgrad_c<3>(q, g->Co, nchunks, s); break;
        case 4: launch_k1_wgrad_c<4>(q, g->Co, nchunks, s); break;
        case 5: launch_k1_wgrad_c<5>(q, g->Co, nchunks, s); break;
        default: launch_k1_wgrad_c<6>(q, g->Co, nchunks, s); break;
      }
      // one "tile" holding the whole [Ci][Co] slab: ci_t = Ci, co_t = Co
      n3d_final_job job;
      fill_job(deferred ? deferred : &job, q.partial, q.pbias, dw, dbias, nchunks, 1, 1, 1, g->Ci, g->Co, g->Co, g->Ci, 1);
      if (!deferred) if (int e = n3d_wgrad_finalize_batch(&job, 1, stream)) return e;
      N3D_LAUNCH_CHECK();
      return N3D_OK;
    }
  }
  WgradPlan p = wgrad_plan(g->B, No, g->Ci, g->Co, taps);
  if (p.partial_floats + p.pbias_floats > avail) { set_error("conv_bwd_weight: workspace too small"); return N3D_ERR_WORKSPACE; }
  WgradArgs a;
  a.x = x; a.xld = xld; a.Di = g->Di; a.Hi = g->Hi; a.Wi = g->Wi; a.Ci = g->Ci;
  a.dy = dy; a.dyld = dyld; a.Do = g->Do; a.Ho = g->Ho; a.Wo = g->Wo; a.Co = g->Co;
  a.B = g->B; a.k = g->k; a.stride = g->stride; a.dil = g->dil; a.pad = g->pad; a.flags = flags; a.in_gate = in_gate;
  a.partial = wsf; a.pbias = wsf + p.partial_floats; a.tci = p.tci; a.tco = p.tco; a.chunk = p.chunk;
  a.fNo = FastDiv((uint32_t)No); a.fWo = FastDiv((uint32_t)g->Wo); a.fHo = FastDiv((uint32_t)g->Ho);
  N3D_CHECK_ARG((int64_t)g->B * No < (1ll << 31), "conv_bwd_weight: tensor too large for 32-bit voxel indexing");
  N3D_CHECK_ARG(g->Ci % 4 == 0 && xld % 4 == 0 && aligned_quad(x, sb16), "conv_bwd_weight: i-side tensor needs C %% 4 == 0 and quad alignment");
  if (p.ci_t == 8 && p.co_t == 16) launch_wgrad_t<8, 16>(a, p, s);
  else if (p.ci_t == 8 && p.co_t == 8) launch_wgrad_t<8, 8>(a, p, s);
  else if (p.ci_t == 8 && p.co_t == 4) launch_wgrad_t<8, 4>(a, p, s);
  else if (p.ci_t == 4 && p.co_t == 16) launch_wgrad_t<4, 16>(a, p, s);
  else if (p.ci_t == 4 && p.co_t == 8) launch_wgrad_t<4, 8>(a, p, s);
  else launch_wgrad_t<4, 4>(a, p, s);
  n3d_final_job job;
  fill_job(deferred ? deferred : &job, a.partial, a.pbias, dw, transposed ? nullptr : dbias, p.nchunks, p.ntiles, p.tci, p.tco, p.ci_t, p.co_t, g->Co, g->Ci, taps);
  if (!deferred) if (int e = n3d_wgrad_finalize_batch(&job, 1, stream)) return e;
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_conv_bwd_weight(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, float* dw, float* dbias, int flags,
                        const float* in_gate, void* ws, size_t ws_bytes, n3d_final_job* deferred, void* stream) {
  if (int e = check_geom(g, "conv_bwd_weight")) return e;
  N3D_CHECK_ARG(x && dy && (dw || dbias), "conv_bwd_weight: bad pointers");
  return run_wgrad(g, x, xld, dy, dyld, dw, dbias, flags, in_gate, ws, ws_bytes, stream, false, deferred);
}

int n3d_conv_bwd_both(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, const float* w, float* dx,
                      int64_t dxld, int flags_data, const float* relu_src, int64_t rld, const float* out_gate, void* ws_data,
                      size_t ws_data_bytes, float* dw, float* dbias, int flags_weight, const float* in_gate, void* ws_weight,
                      size_t ws_weight_bytes, n3d_final_job* deferred, void* stream) {
  if (int e = check_geom(g, "conv_bwd_both")) return e;
  N3D_CHECK_ARG(x && dy && w && dx && (dw || dbias) && dyld >= g->Co && dxld >= g->Ci, "conv_bwd_both: bad pointers/pitches");
  DualReq rq = {w, dx, dxld, flags_data & ~N3D_RELU_IN, relu_src, rld, out_gate, ws_data, ws_data_bytes, false};
  if (int e = run_wgrad(g, x, xld, dy, dyld, dw, dbias, flags_weight, in_gate, ws_weight, ws_weight_bytes, stream, false, deferred, &rq)) return e;
  if (rq.done) return N3D_OK;
  return run_gather(g, true, dy, dyld, w, nullptr, dx, dxld, flags_data & ~N3D_RELU_IN, nullptr, relu_src, rld, out_gate, nullptr, ws_data,
                    ws_data_bytes, stream);
}

int n3d_convT_bwd_both(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, const float* w, float* dx,
                       int64_t dxld, int flags_data, void* ws_data, size_t ws_data_bytes, float* dw, int flags_weight, void* ws_weight,
                       size_t ws_weight_bytes, n3d_final_job* deferred, void* stream) {
  if (int e = check_geom(g, "convT_bwd_both")) return e;
  N3D_CHECK_ARG(x && dy && w && dx && dw && xld >= g->Co && dyld >= g->Ci && dxld >= g->Co, "convT_bwd_both: bad pointers/pitches");
  DualReq rq = {w, dx, dxld, flags_data & ~N3D_RELU_IN, nullptr, 0, nullptr, ws_data, ws_data_bytes, false};
  if (int e = run_wgrad(g, dy, dyld, x, xld, dw, nullptr, flags_weight & ~N3D_RELU_IN, nullptr, ws_weight, ws_weight_bytes, stream, true,
                        deferred, &rq))
    return e;
  if (rq.done) return N3D_OK;
  return run_gather(g, false, dy, dyld, w, nullptr, dx, dxld, flags_data & ~N3D_RELU_IN, nullptr, nullptr, 0, nullptr, nullptr, ws_data,
                    ws_data_bytes, stream);
}

int n3d_dwconv_batch(const n3d_dw_job* jobs, int n, void* stream) {
  N3D_CHECK_ARG(jobs && n >= 1 && n <= N3D_MAX_GROUP_TERMS, "dwconv_batch: 1..8 jobs");
  DwArgsN js;
  int64_t Nd0 = 0;
  int C0 = 0, B0 = 0;
  for (int i = 0; i < 8; ++i) {
    const n3d_dw_job* q = &jobs[i < n ? i : 0];
    N3D_CHECK_ARG(q->g && q->src && q->w && q->dst, "dwconv_batch: null pointers");
    if (int e = check_geom(q->g, "dwconv_batch")) return e;
    N3D_CHECK_ARG(q->g->depthwise, "dwconv_batch: depthwise geometries only");
    N3D_CHECK_ARG(q->sld % 4 == 0 && q->dld % 4 == 0 && aligned16(q->src) && aligned16(q->dst), "dwconv_batch: needs 16-byte aligned pitched rows");
    js.j[i] = dw_args(q->g, q->data_grad != 0, q->src, q->sld, q->w, q->bias, q->dst, q->dld, q->flags);
    const int64_t Nd = (int64_t)js.j[i].Dd * js.j[i].Hd * js.j[i].Wd;
    if (i == 0) { Nd0 = Nd; C0 = js.j[i].C; B0 = q->g->B; }
    N3D_CHECK_ARG(Nd == Nd0 && js.j[i].C == C0 && q->g->B == B0, "dwconv_batch: the jobs must share output shape, channel count and batch");
  }
  hipLaunchKernelGGL(dw_gatherN_kernel, dim3((unsigned)cdiv(Nd0 * (C0 / 4), 256), B0, n), dim3(256), (size_t)(C0 / 4) * 27 * sizeof(float4),
                     (hipStream_t)stream, js);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_conv_fwd2(const n3d_conv_fwd_call* c0, const n3d_conv_fwd_call* c1, void* stream) {
  N3D_CHECK_ARG(c0 && c1 && c0->g && c1->g, "conv_fwd2: bad args");
  const n3d_conv_fwd_call* cs[2] = {c0, c1};
  for (int i = 0; i < 2; ++i) {
    if (int e = check_geom(cs[i]->g, "conv_fwd2")) return e;
    N3D_CHECK_ARG(cs[i]->x && cs[i]->w && cs[i]->y, "conv_fwd2: null pointers");
  }
  if (!((c0->flags | c1->flags) & N3D_NO_MFMA) && !c0->g->depthwise && !c1->g->depthwise) {
    {
      const VoxCall vc[2] = {vox_call_fwd(c0), vox_call_fwd(c1)};
      const int r = mfma_vox_multi_try(2, vc, (hipStream_t)stream);
      if (r < 0) return r;
      if (r == 1) return N3D_OK;
    }
    // forward conv = gather with data_grad=false; transposed forward = gather with data_grad=true (run_gather convention)
    const int r = mfma_conv_pair_try(c0->g, c0->transposed != 0, c0->x, c0->xld, c0->w, c0->bias, c0->y, c0->yld, c0->flags, c0->in_gate,
                                     c0->stats, c0->ws, c0->ws_bytes, c1->g, c1->transposed != 0, c1->x, c1->xld, c1->w, c1->bias, c1->y,
                                     c1->yld, c1->flags, c1->in_gate, c1->stats, c1->ws, c1->ws_bytes, (hipStream_t)stream, nullptr, nullptr);
    if (r < 0) return r;
    if (r == 1) return N3D_OK;
  }
  for (int i = 0; i < 2; ++i) {
    const n3d_conv_fwd_call* c = cs[i];
    const int e = c->transposed ? n3d_convT_fwd(c->g, c->x, c->xld, c->w, c->bias, c->y, c->yld, c->flags, c->in_gate, c->stats, c->ws, c->ws_bytes, stream)
                                : n3d_conv_fwd(c->g, c->x, c->xld, c->w, c->bias, c->y, c->yld, c->flags, c->in_gate, c->stats, c->ws, c->ws_bytes, stream);
    if (e) return e;
  }
  return N3D_OK;
}

int n3d_conv_fwdN(const n3d_conv_fwd_call* calls, int n, void* stream) {
  N3D_CHECK_ARG(calls && n >= 1 && n <= 4, "conv_fwdN: 1..4 calls");
  if (n >= 3) {
    const n3d_conv_geom* g[4]; bool dg[4]; const float* src[4]; int64_t sld[4]; const float* w[4]; const float* bias[4]; float* dst[4];
    int64_t dld[4]; int flags[4]; const float* gate[4]; double* stats[4]; void* ws[4]; size_t wsb[4];
    bool mf = true;
    for (int i = 0; i < n; ++i) {
      const n3d_conv_fwd_call* c = &calls[i];
      N3D_CHECK_ARG(c->g && c->x && c->w && c->y, "conv_fwdN: null pointers");
      if (int e = check_geom(c->g, "conv_fwdN")) return e;
      mf = mf && !(c->flags & N3D_NO_MFMA) && !c->g->depthwise;
      g[i] = c->g; dg[i] = c->transposed != 0; src[i] = c->x; sld[i] = c->xld; w[i] = c->w; bias[i] = c->bias; dst[i] = c->y; dld[i] = c->yld;
      flags[i] = c->flags; gate[i] = c->in_gate; stats[i] = c->stats; ws[i] = c->ws; wsb[i] = c->ws_bytes;
    }
    if (mf) {
      VoxCall vc[4];
      for (int i = 0; i < n; ++i) vc[i] = vox_call_fwd(&calls[i]);
      int r = mfma_vox_multi_try(n, vc, (hipStream_t)stream);
      if (r < 0) return r;
      if (r == 1) return N3D_OK;
      r = mfma_conv_multi_try(n, g, dg, src, sld, w, bias, dst, dld, flags, gate, stats, ws, wsb, (hipStream_t)stream);
      if (r < 0) return r;
      if (r == 1) return N3D_OK;
    }
  }
  // two at a time (one launch each where foldable), a leftover alone
  int i = 0;
  for (; i + 1 < n; i += 2)
    if (int e = n3d_conv_fwd2(&calls[i], &calls[i + 1], stream)) return e;
  if (i < n) {
    const n3d_conv_fwd_call* c = &calls[i];
    const int e = c->transposed ? n3d_convT_fwd(c->g, c->x, c->xld, c->w, c->bias, c->y, c->yld, c->flags, c->in_gate, c->stats, c->ws, c->ws_bytes, stream)
                                : n3d_conv_fwd(c->g, c->x, c->xld, c->w, c->bias, c->y, c->yld, c->flags, c->in_gate, c->stats, c->ws, c->ws_bytes, stream);
    if (e) return e;
  }
  return N3D_OK;
}

int n3d_conv_bwd_data2(const n3d_conv_bwd_call* c0, const n3d_conv_bwd_call* c1, void* stream) {
  N3D_CHECK_ARG(c0 && c1 && c0->g && c1->g, "conv_bwd_data2: bad args");
  const n3d_conv_bwd_call* cs[2] = {c0, c1};
  bool pair = c0->dx != c1->dx;  // both may accumulate into one input gradient: then they must run one after the other
  for (int i = 0; i < 2; ++i) {
    const n3d_conv_bwd_call* c = cs[i];
    if (int e = check_geom(c->g, "conv_bwd_data2")) return e;
    N3D_CHECK_ARG(c->dy && c->w && c->dx, "conv_bwd_data2: null pointers");
    if (c->transposed && (c->relu_src || c->out_gate)) N3D_UNSUPPORTED("conv_bwd_data2: transposed conv with relu / gate extras");
    pair = pair && !(c->flags_data & N3D_NO_MFMA) && !c->g->depthwise;
  }
  if (pair) {
    // data gradient of a conv = gather with data_grad=true, of a transposed conv = gather with data_grad=false (run_gather convention)
    {
      VoxCall vc[2];
      for (int i = 0; i < 2; ++i) {
        const n3d_conv_bwd_call* c = cs[i];
        vc[i] = VoxCall{c->g, !c->transposed, c->dy, c->dyld, c->w, nullptr, c->dx, c->dxld, c->flags_data & ~N3D_RELU_IN, nullptr, c->relu_src,
                        c->out_gate, nullptr, c->ws_data, c->ws_data_bytes};
      }
      const int r = mfma_vox_multi_try(2, vc, (hipStream_t)stream);
      if (r < 0) return r;
      if (r == 1) return N3D_OK;
    }
    const PairExtras x0{c0->relu_src, c0->rld, c0->out_gate}, x1{c1->relu_src, c1->rld, c1->out_gate};
    const int r = mfma_conv_pair_try(c0->g, !c0->transposed, c0->dy, c0->dyld, c0->w, nullptr, c0->dx, c0->dxld, c0->flags_data & ~N3D_RELU_IN, nullptr,
                                     nullptr, c0->ws_data, c0->ws_data_bytes, c1->g, !c1->transposed, c1->dy, c1->dyld, c1->w, nullptr, c1->dx,
                                     c1->dxld, c1->flags_data & ~N3D_RELU_IN, nullptr, nullptr, c1->ws_data, c1->ws_data_bytes, (hipStream_t)stream,
                                     &x0, &x1);
    if (r < 0) return r;
    if (r == 1) return N3D_OK;
  }
  for (int i = 0; i < 2; ++i) {
    const n3d_conv_bwd_call* c = cs[i];
    const int e = c->transposed ? n3d_convT_bwd_data(c->g, c->dy, c->dyld, c->w, c->dx, c->dxld, c->flags_data, c->ws_data, c->ws_data_bytes, stream)
                                : n3d_conv_bwd_data(c->g, c->dy, c->dyld, c->w, c->dx, c->dxld, c->flags_data, c->relu_src, c->rld, c->out_gate,
                                                    c->ws_data, c->ws_data_bytes, stream);
    if (e) return e;
  }
  return N3D_OK;
}

int n3d_conv_bwd_both2(const n3d_conv_bwd_call* c0, const n3d_conv_bwd_call* c1, void* stream) {
  N3D_CHECK_ARG(c0 && c1 && c0->g && c1->g, "conv_bwd_both2: bad args");
  const n3d_conv_bwd_call* cs[2] = {c0, c1};
  hipStream_t s = (hipStream_t)stream;
  bool quad = true;
  for (int i = 0; i < 2; ++i) {
    const n3d_conv_bwd_call* c = cs[i];
    if (int e = check_geom(c->g, "conv_bwd_both2")) return e;
    N3D_CHECK_ARG(c->x && c->dy && c->w && c->dx && c->dw, "conv_bwd_both2: null pointers");
    if (c->transposed && (c->dbias || c->relu_src || c->out_gate || c->in_gate)) N3D_UNSUPPORTED("conv_bwd_both2: transposed conv with bias / relu / gate extras");
    quad = quad && !((c->flags_data | c->flags_weight) & N3D_NO_MFMA) && c->xld % 4 == 0 && c->dyld % 4 == 0 && c->ws_data && c->ws_weight;
  }
  // the two data gradients must not touch the same memory (both may accumulate into one input gradient)
  quad = quad && c0->dx != c1->dx && mfma_bwd_quad_ok(c0->g, c0->transposed != 0, c1->g, c1->transposed != 0);
  BwdOne b[2];
  if (quad) {
    for (int i = 0; i < 2 && quad; ++i) {
      const n3d_conv_bwd_call* c = cs[i];
      const n3d_conv_geom* g = c->g;
      const int taps = g->k * g->k * g->k;
      const size_t skip = align_up(packed_floats(g) * 4, 256);
      if (c->ws_weight_bytes <= skip || c->ws_data_bytes < (size_t)taps * g->Ci * g->Co * 4) { quad = false; break; }
      float* wsf = (float*)((char*)c->ws_weight + skip);
      const size_t avail = (c->ws_weight_bytes - skip) / 4;
      const size_t nt16 = (size_t)taps * (g->Ci / 16) * (g->Co / 16);
      if ((N3D_WG16_SLABS + nt16) * (256 + 16) > avail) { quad = false; break; }
      b[i] = BwdOne{g, c->transposed != 0, c->dy, c->dyld, (const float*)c->ws_data, c->dx, c->dxld, c->flags_data & ~N3D_RELU_IN,
                    c->relu_src, c->rld, c->out_gate, c->x, c->xld, c->transposed ? (c->flags_weight & ~N3D_RELU_IN) : c->flags_weight,
                    c->in_gate, wsf, wsf + (N3D_WG16_SLABS + nt16) * 256, (N3D_WG16_SLABS + nt16) * 256, 0, 0};
    }
  }
  if (quad) {
    for (int i = 0; i < 2; ++i) {
      const n3d_conv_bwd_call* c = cs[i];
      if (!(c->flags_data & N3D_PREPACKED))
        mfma_pack16(c->w, (float*)c->ws_data, c->g->Co, c->g->Ci, c->g->k * c->g->k * c->g->k, c->transposed ? 0 : 1, s);
    }
    const int r = mfma_bwd_quad_try(&b[0], &b[1], s);
    if (r < 0) return r;
    if (r == 1) {
      for (int i = 0; i < 2; ++i) {
        const n3d_conv_bwd_call* c = cs[i];
        const n3d_conv_geom* g = c->g;
        const int taps = g->k * g->k * g->k;
        n3d_final_job job;
        fill_job(&job, b[i].partial, b[i].pbias, c->dw, c->transposed ? nullptr : c->dbias, b[i].nch, b[i].ntl, g->Ci / 16, g->Co / 16, 16, 16, g->Co,
                 g->Ci, taps);
        if (c->deferred) *c->deferred = job;
        else if (int e = n3d_wgrad_finalize_batch(&job, 1, stream)) return e;
      }
      return N3D_OK;
    }
  }
  for (int i = 0; i < 2; ++i) {
    const n3d_conv_bwd_call* c = cs[i];
    const int e = c->transposed
                      ? n3d_convT_bwd_both(c->g, c->x, c->xld, c->dy, c->dyld, c->w, c->dx, c->dxld, c->flags_data, c->ws_data, c->ws_data_bytes, c->dw,
                                           c->flags_weight, c->ws_weight, c->ws_weight_bytes, c->deferred, stream)
                      : n3d_conv_bwd_both(c->g, c->x, c->xld, c->dy, c->dyld, c->w, c->dx, c->dxld, c->flags_data, c->relu_src, c->rld, c->out_gate,
                                          c->ws_data, c->ws_data_bytes, c->dw, c->dbias, c->flags_weight, c->in_gate, c->ws_weight,
                                          c->ws_weight_bytes, c->deferred, stream);
    if (e) return e;
  }
  return N3D_OK;
}

int n3d_conv_pack_info(const n3d_conv_geom* g, int data_grad, int flags, int32_t* layout, int32_t* cdp, int64_t* floats) {
  if (int e = check_geom(g, "conv_pack_info")) return e;
  N3D_CHECK_ARG(layout && cdp && floats, "conv_pack_info: null outputs");
  const int taps = g->k * g->k * g->k;
  const int Cs = data_grad ? g->Co : g->Ci, Cd = data_grad ? g->Ci : g->Co;
  if (g->depthwise) { *layout = -1; *cdp = 0; *floats = 0; return N3D_OK; }  // depthwise kernels read native weights
  if (const int l16 = vox16_layout(g, data_grad != 0, flags)) { *layout = l16; *cdp = Cd; *floats = ((int64_t)27 * Cd * Cs + 1) / 2; return N3D_OK; }   // bf16 [27][cd][cs]
  const int ml = mfma_pack_layout(g, data_grad != 0, flags);
  if (ml == 2 || ml == 3) { *layout = ml; *cdp = Cd; *floats = (int64_t)27 * Cd * Cs; return N3D_OK; }
  if (ml == 1) { *layout = 1; *cdp = Cd; *floats = (int64_t)taps * Cs * Cd; return N3D_OK; }
  const int cot = pick_cot(Cd);
  *layout = 0; *cdp = (int)align_up(Cd, cot); *floats = (int64_t)taps * Cs * (*cdp);
  return N3D_OK;
}

int n3d_pack_batch(const n3d_pack_job* jobs, int njobs, void* stream) {
  N3D_CHECK_ARG(jobs && njobs >= 0, "pack_batch: bad args");
  for (int i = 0; i < njobs; ++i) {
    const n3d_pack_job& q = jobs[i];
    N3D_CHECK_ARG(q.w && q.dst && q.Co > 0 && q.Ci > 0 && q.Co < 65536 && q.Ci < 65536 && q.cdp >= 0 && q.cdp < 65536 && q.taps > 0 && q.taps < 256 &&
                  q.layout >= 0 && q.layout <= 5 && ((uintptr_t)q.w & 3) == 0 && ((uintptr_t)q.dst & 3) == 0, "pack_batch: bad job");
  }
  int base = 0;
  while (base < njobs) {
    SegTable sp;
    int n = seg_group<2>(jobs + base, njobs - base, N3D_PACK_JOBS, sp, [](const n3d_pack_job& q, const void** o) { o[0] = q.w; o[1] = q.dst; });
    PackJobs pj;
    pj.seg = sp.sb;
    int blocks[N3D_PACK_JOBS], nblk = 0;
    uint32_t start[N3D_PACK_JOBS];
    for (int i = 0; i < n; ++i) {
      const n3d_pack_job& q = jobs[base + i];
      const int Cs = q.data_grad ? q.Co : q.Ci, Cd = q.data_grad ? q.Ci : q.Co;
      blocks[i] = pack_job_blocks(q.layout, Cs, Cd, q.cdp, q.Co, q.taps);
    }
    n = unit_map(blocks, n, N3D_PACK_GSHIFT, pj.unit, start, &nblk);
    for (int i = 0; i < n; ++i) {
      const n3d_pack_job& q = jobs[base + i];
      PackJobD& d = pj.j[i];
      d.w = sp.enc(q.w); d.dst = sp.enc(q.dst); d.start = start[i];
      d.Co = (uint16_t)q.Co; d.Ci = (uint16_t)q.Ci; d.cdp = (uint16_t)q.cdp; d.taps = (uint8_t)q.taps;
      d.mode = (uint8_t)(q.layout | (q.data_grad ? 16 : 0));
    }
    for (int i = n; i < N3D_PACK_JOBS; ++i) pj.j[i] = pj.j[0];
    if (nblk > 0) hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, pj);
    base += n;
  }
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_wgrad_finalize_batch(const n3d_final_job* jobs, int njobs, void* stream) {
  N3D_CHECK_ARG(jobs && njobs >= 0, "wgrad_finalize_batch: bad args");
  for (int i = 0; i < njobs; ++i) {
    const n3d_final_job& q = jobs[i];
    if (q.nchunks == 0) continue;   // nothing deferred
    N3D_CHECK_ARG(q.nchunks > 0 && q.ntiles > 0 && q.ntiles < 65536 && q.Co > 0 && q.Ci > 0 && q.Co < 65536 && q.Ci < 65536 && q.tci > 0 &&
                  q.tci < 256 && q.tco > 0 && q.tco < 256 && q.ci_t > 0 && q.ci_t < 65536 && q.co_t > 0 && q.co_t < 65536 && q.taps > 0 &&
                  q.taps < 256 && q.ntiles == q.taps * q.tci * q.tco && q.partial, "wgrad_finalize_batch: bad job");
  }
  int base = 0;
  while (base < njobs) {
    SegTable sp;
    int n = seg_group<4>(jobs + base, njobs - base, N3D_FINAL_JOBS, sp, [](const n3d_final_job& q, const void** o) {
      o[0] = q.partial; o[1] = q.pbias; o[2] = q.dw; o[3] = q.dbias; });
    FinalJobs fj;
    fj.seg = sp.sb;
    int blocks[N3D_FINAL_JOBS], nblk = 0;
    uint32_t start[N3D_FINAL_JOBS];
    for (int i = 0; i < n; ++i) {
      const n3d_final_job& q = jobs[base + i];
      blocks[i] = q.nchunks > 0 ? final_job_blocks(q.nchunks, q.ntiles, q.tci, q.tco, q.ci_t, q.co_t, q.taps) : 0;
    }
    n = unit_map(blocks, n, N3D_FINAL_GSHIFT, fj.unit, start, &nblk);
    for (int i = 0; i < n; ++i) {
      const n3d_final_job& q = jobs[base + i];
      FinalJobD& d = fj.j[i];
      d.partial = sp.enc(q.partial); d.pbias = sp.enc(q.pbias); d.dw = sp.enc(q.dw); d.dbias = sp.enc(q.dbias); d.start = start[i];
      d.nchunks = q.nchunks; d.ntiles = (uint16_t)q.ntiles; d.Co = (uint16_t)q.Co; d.Ci = (uint16_t)q.Ci;
      d.tci = (uint8_t)q.tci; d.tco = (uint8_t)q.tco; d.ci_t = (uint16_t)q.ci_t; d.co_t = (uint16_t)q.co_t; d.taps = (uint8_t)q.taps; d.pad_ = 0;
    }
    for (int i = n; i < N3D_FINAL_JOBS; ++i) fj.j[i] = fj.j[0];
    if (nblk > 0) hipLaunchKernelGGL(wgrad_final_batch_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, fj);
    base += n;
  }
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

// host-only self-test of the job-table pointer encoding (tests/test_host_cpu.py): scattered fake addresses are grouped and
// every encoded pointer must decode to itself; returns the number of launches the 300 jobs would take, < 0 on a mismatch
int n3d_selftest_job_tables(void) {
  std::vector<n3d_final_job> jobs(300);
  uint64_t r = 0x9e3779b97f4a7c15ull;
  auto next = [&]() { r ^= r << 13; r ^= r >> 7; r ^= r << 17; return r; };
  for (size_t i = 0; i < jobs.size(); ++i) {
    n3d_final_job& q = jobs[i];
    const uintptr_t region[4] = {0x700000000000ull, 0x700000000000ull, 0x7f0000000000ull, 0x7f0000000000ull};
    uintptr_t a[4];
    // the first 100 jobs live in two compact allocations, the next 100 are spread over 64 GB, the last 100 over 4 TB
    for (int k = 0; k < 4; ++k) a[k] = region[k] + (next() % (i < 100 ? (1ull << 28) : (i < 200 ? (1ull << 36) : (1ull << 42)))) / 4 * 4;
    q.partial = (const float*)a[0]; q.pbias = (i & 1) ? (const float*)a[1] : nullptr; q.dw = (float*)a[2]; q.dbias = (i & 2) ? (float*)a[3] : nullptr;
  }
  int launches = 0;
  size_t base = 0;
  while (base < jobs.size()) {
    SegTable st;
    const int n = seg_group<4>(jobs.data() + base, (int)(jobs.size() - base), N3D_FINAL_JOBS, st, [](const n3d_final_job& q, const void** o) {
      o[0] = q.partial; o[1] = q.pbias; o[2] = q.dw; o[3] = q.dbias; });
    if (n < 1) return -1;
    for (int i = 0; i < n; ++i) {
      const n3d_final_job& q = jobs[base + i];
      const void* ptrs[4] = {q.partial, q.pbias, q.dw, q.dbias};
      for (int k = 0; k < 4; ++k) {
        const uint32_t e = st.enc(ptrs[k]);
        if (!ptrs[k]) { if (e != N3D_NO_OFF) return -2; continue; }
        if (e == N3D_NO_OFF) return -3;
        if (st.sb.b[e >> 29] + (uintptr_t)(e & 0x1fffffffu) * sizeof(float) != (uintptr_t)ptrs[k]) return -4;
      }
    }
    base += n;
    ++launches;
  }
  if (base != jobs.size()) return -5;
  return launches;
}

int n3d_convT_bwd_weight(const n3d_conv_geom* g, const float* x, int64_t xld, const float* dy, int64_t dyld, float* dw, float* dbias, int flags,
                         void* ws, size_t ws_bytes, n3d_final_job* deferred, void* stream) {
  if (int e = check_geom(g, "convT_bwd_weight")) return e;
  N3D_CHECK_ARG(x && dy && (dw || dbias) && xld >= g->Co && dyld >= g->Ci, "convT_bwd_weight: bad pointers/pitches");
  // kernel roles: i-side tensor = dy (Ci channels), o-side tensor = x (Co channels); see run_wgrad
  if (dw) {
    // the storage flags follow the tensors into their swapped roles
    const int rf = (flags & ~(N3D_RELU_IN | N3D_SRC_BF16 | N3D_DST_BF16)) | ((flags & N3D_SRC_BF16) ? N3D_DST_BF16 : 0) | ((flags & N3D_DST_BF16) ? N3D_SRC_BF16 : 0);
    int e = run_wgrad(g, dy, dyld, x, xld, dw, nullptr, rf, nullptr, ws, ws_bytes, stream, true, deferred);
    if (e) return e;
  } else if (deferred) {
    deferred->nchunks = 0;
  }
  if (dbias) {
    // bias gradient = per-channel sum of dy over the i side
    hipStream_t s = (hipStream_t)stream;
    N3D_CHECK_ARG(g->Ci % 4 == 0 && dyld % 4 == 0 && aligned_quad(dy, flags & N3D_DST_BF16) && g->Ci <= 256, "convT_bwd_weight: dy needs C %% 4 == 0");
    const size_t skip = align_up(packed_floats(g) * 4, 256);
    float* wsf = (float*)((char*)ws + skip);
    const int64_t total = (int64_t)g->B * g->Di * g->Hi * g->Wi;
    int64_t nblk = cdiv(total, 4096);
    if (nblk > 256) nblk = 256;
    const int64_t chunk = cdiv(total, nblk);
    nblk = cdiv(total, chunk);
    if (!ws || ws_bytes < skip + (size_t)nblk * g->Ci * 4) { set_error("convT_bwd_weight: workspace too small"); return N3D_ERR_WORKSPACE; }
    if (deferred && deferred->nchunks > 0) {
      // the weight-gradient slabs of this call stay in `ws` until the caller's batched finalize: the channel sums must not
      // land on them.  They go behind the slabs if there is room, otherwise the slabs are finalized now.
      const float* end = deferred->partial + (size_t)deferred->nchunks * deferred->ntiles * deferred->ci_t * deferred->co_t;
      if (deferred->pbias) {
        const float* e2 = deferred->pbias + (size_t)deferred->nchunks * deferred->tco * deferred->co_t;
        if (e2 > end) end = e2;
      }
      float* behind = (float*)(((uintptr_t)end + 255) & ~(uintptr_t)255);
      if ((char*)behind + (size_t)nblk * g->Ci * 4 <= (char*)ws + ws_bytes && (char*)behind >= (char*)wsf) {
        wsf = behind;
      } else {
        if (int e = n3d_wgrad_finalize_batch(deferred, 1, stream)) return e;
        deferred->nchunks = 0;
      }
    }
    // (not deferred: the slab region is free again once run_wgrad's final kernel has been enqueued -- stream order)
    // dy is the SECOND activation tensor of a transposed weight-gradient call... but the kernel roles are swapped (see above):
    // the caller's dy (Ci channels, i side) carries the N3D_DST_BF16 flag
    if (flags & N3D_DST_BF16)
      hipLaunchKernelGGL(channel_sum_kernel<bf16_t>, dim3((unsigned)nblk), dim3(256), (size_t)4 * (g->Ci / 4) * 4 * sizeof(float), s,
                         reinterpret_cast<const bf16_t*>(dy), dyld, total, g->Ci, chunk, wsf);
    else
      hipLaunchKernelGGL(channel_sum_kernel<float>, dim3((unsigned)nblk), dim3(256), (size_t)4 * (g->Ci / 4) * 4 * sizeof(float), s, dy, dyld, total, g->Ci,
                         chunk, wsf);
    hipLaunchKernelGGL(channel_sum_final_kernel, dim3((unsigned)cdiv(g->Ci, 256)), dim3(256), 0, s, wsf, (int)nblk, g->Ci, dbias);
    N3D_LAUNCH_CHECK();
  }
  return N3D_OK;
}

}  // extern "C"
