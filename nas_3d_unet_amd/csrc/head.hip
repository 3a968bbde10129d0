// Fused network head (nas.py:50-52, searched.py:91-93) and its Dice loss (loss.py:12-14):
//   p = sigmoid(conv1x1x1(Dropout3d(x)) + bias)            one pass over x, one write of p
//   Dice partial sums (sum p*t, sum p, sum t per (b, c))     in the same pass when the target is given
// and the whole backward -- d loss / d p from the Dice sums, sigmoid', data gradient, weight and bias gradient -- in ONE
// pass over (x, t) that writes dx.  The reference runs Dropout3d, Conv3d, Sigmoid, four reductions and their autograd
// counterparts as separate full-tensor passes; here the head is HBM-bound streaming work: forward reads Ci and writes
// Co channels per voxel, backward reads Ci (+ Co target channels) and writes Ci.
// Dropout3d(p) zeroes whole channels per sample and rescales by 1/(1-p) (prim_ops.py:66,72-73): a (B, Ci) gate applied
// to x on load.  The gate is either supplied or drawn on the device by n3d_dropout3d_gate from a counter-based generator
// (seed + step counter in device memory, so a captured HIP graph draws a fresh mask on every replay).
#include "n3d_common.h"

namespace n3d {

constexpr int HEAD_CHUNK = 1024;   // voxels per workgroup (4 per thread; 2048: head_fwd 17.6 / head_bwd 21.8 us at (2,12,64^3), 1024: 15.4 / 17.6, 512: 16.4 / 21.1)
constexpr int HEAD_COMAX = 4;      // output channels computed per voxel (weights zero-padded)

// splitmix64 of (seed, counter, element): uniform in [0,1) with 24 bits
__host__ __device__ inline float head_uniform(uint64_t seed, uint32_t counter, uint32_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (((uint64_t)counter << 20) + idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

__global__ void dropout3d_gate_kernel(uint32_t* __restrict__ state, float p, int n, float* __restrict__ gate) {
  __shared__ uint32_t st[3];
  if (threadIdx.x < 3) st[threadIdx.x] = state[threadIdx.x];
  __syncthreads();
  const uint64_t seed = ((uint64_t)st[1] << 32) | st[0];
  const float keep = 1.0f / (1.0f - p);
  for (int i = threadIdx.x; i < n; i += blockDim.x) gate[i] = head_uniform(seed, st[2], (uint32_t)i) >= p ? keep : 0.f;
  if (threadIdx.x == 0) state[2] = st[2] + 1;
}

// element offset of channel quad q of a voxel record: q * 4 in the ordinary pitched layout; node-planar (include/n3d.h, n3d_head):
// node q / QN starts node_stride elements further, the quad sits at (q % QN) * 4 inside the node's record
__device__ __forceinline__ int64_t head_quad_off(int q, int qn, int64_t node_stride) {
  return qn ? (int64_t)(q / qn) * node_stride + (q % qn) * 4 : (int64_t)q * 4;
}

// Byte targets of a workgroup's 1024-voxel chunk: one 4-byte load per lane and channel (voxels 4 * tid .. 4 * tid + 3) instead of four
// 1-byte loads (a byte load moves 64 B per wave instruction: the byte form of the per-voxel loads was SLOWER than the float form),
// handed to the lanes that own the voxels (tid + 256 k) through LDS.
template <int NCO>
__device__ __forceinline__ void head_target_quads_issue(const void* t, int64_t base, int64_t tsc, int tid, uint32_t (&q)[NCO]) {
#pragma unroll
  for (int co = 0; co < NCO; ++co) q[co] = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(t) + base + co * tsc)[tid];
}
template <int NCO, int NK>
__device__ __forceinline__ void head_target_quads_take(const uint32_t (&q)[NCO], uint32_t (*lds)[256], int tid, float (&tv)[NK][NCO]) {
#pragma unroll
  for (int co = 0; co < NCO; ++co) lds[co][tid] = q[co];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int co = 0; co < NCO; ++co) tv[k][co] = (float)((lds[co][(tid >> 2) + 64 * k] >> (8 * (tid & 3))) & 0xffu);
}

struct HeadFwdArgs {
  const void* x; int64_t xld; int64_t N; int64_t xns; int qn;   // xns / qn: node stride and quads per node (qn == 0: pitched layout)
  const float* w; const float* bias; const float* gate;
  float* p; int64_t psb, psc, psv; float* logits;
  const void* t; int64_t tsb, tsc, tsv;      // targets: float, or bytes {0, 1} (TT = uint8_t; strides in elements)
  double* partial; int rows; int Ci, Co;
  int t_quad;   // byte targets, dense and 4-aligned, whole chunks: a lane fetches FOUR voxels' bytes (head_target_quads)
};

// thread = voxel; lanes walk consecutive voxels (x: Ci * sizeof(TX) contiguous bytes per voxel, p / t: strided per channel)
// NCO: output channels computed per voxel -- 3 = exactly three (the reference's out_channels, config.yml: no fourth channel of zero
// weights, no run-time channel tests), HEAD_COMAX = any count up to four
// TT: storage of the targets -- float, or uint8_t holding exactly {0, 1} (generator.py:230-248 yields booleans): a quarter of the
// target bytes, the same sums bit for bit
template <typename TX, int CIQ, int NCO, typename TT = float>
__global__ __launch_bounds__(256) void head_fwd_kernel(HeadFwdArgs a) {
  N3D_CHAIN_PRIO();
  constexpr int CI = CIQ * 4;
  constexpr bool ALL = NCO != HEAD_COMAX;     // every computed channel is a real one
  __shared__ float wsm[HEAD_COMAX][CI];
  __shared__ double red[HEAD_COMAX * 3][4];
  const int b = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < HEAD_COMAX * CI; i += 256) {
    const int co = i / CI, ci = i - co * CI;
    wsm[co][ci] = co < a.Co ? a.w[co * CI + ci] * (a.gate ? a.gate[b * CI + ci] : 1.f) : 0.f;
  }
  __syncthreads();
  float bias[NCO];
#pragma unroll
  for (int co = 0; co < NCO; ++co) bias[co] = (ALL || co < a.Co) ? a.bias[co] : 0.f;
  const TX* xb = reinterpret_cast<const TX*>(a.x) + (int64_t)b * a.N * a.xld;
  int64_t qoff[CIQ];
#pragma unroll
  for (int q = 0; q < CIQ; ++q) qoff[q] = head_quad_off(q, a.qn, a.xns);
  float spt[HEAD_COMAX], sp[HEAD_COMAX], st[HEAD_COMAX];
#pragma unroll
  for (int co = 0; co < HEAD_COMAX; ++co) spt[co] = sp[co] = st[co] = 0.f;     // (a channel that is not computed stays 0)
  const int64_t v0 = (int64_t)blockIdx.x * HEAD_CHUNK;
  // every operand of the workgroup's four voxel rounds is requested before the first use (predicated, no early exit: a `break` in
  // the loop kept each round's loads behind the previous round's stores -- 2.6 TB/s on a 37 MB pass)
  constexpr int NK = HEAD_CHUNK / 256;
  constexpr bool TU8 = sizeof(TT) == 1;
  __shared__ uint32_t tql[TU8 ? NCO : 1][256];
  const bool tquad = TU8 && a.t_quad;     // (uniform)
  float4 xqs[NK][CIQ];
  float tvs[NK][NCO];
  uint32_t tq[NCO];
  if (tquad) head_target_quads_issue<NCO>(a.t, b * a.tsb + v0, a.tsc, tid, tq);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int64_t v = v0 + tid + k * 256;
    const int64_t vc = v < a.N ? v : v0;
#pragma unroll
    for (int q = 0; q < CIQ; ++q) xqs[k][q] = ld4(xb + vc * a.xld + qoff[q]);
    if (!tquad) {
#pragma unroll
      for (int co = 0; co < NCO; ++co)
        tvs[k][co] = (a.t && (ALL || co < a.Co)) ? (float)reinterpret_cast<const TT*>(a.t)[b * a.tsb + co * a.tsc + vc * a.tsv] : 0.f;
    }
  }
  if (tquad) head_target_quads_take<NCO, NK>(tq, tql, tid, tvs);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int64_t v = v0 + tid + k * 256;
    if (v >= a.N) break;
    const float4 (&xq)[CIQ] = xqs[k];
    const float (&tv)[NCO] = tvs[k];
    float z[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
      float s = bias[co];
#pragma unroll
      for (int q = 0; q < CIQ; ++q) {
        s = fmaf(wsm[co][q * 4 + 0], xq[q].x, s); s = fmaf(wsm[co][q * 4 + 1], xq[q].y, s);
        s = fmaf(wsm[co][q * 4 + 2], xq[q].z, s); s = fmaf(wsm[co][q * 4 + 3], xq[q].w, s);
      }
      z[co] = s;
    }
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
      if (ALL || co < a.Co) {
        const float pr = 1.0f / (1.0f + expf(-z[co]));
        if (a.p) a.p[b * a.psb + co * a.psc + v * a.psv] = pr;
        if (a.logits) a.logits[b * a.psb + co * a.psc + v * a.psv] = z[co];
        spt[co] = fmaf(pr, tv[co], spt[co]); sp[co] += pr; st[co] += tv[co];
      }
    }
  }
  if (!a.partial) return;
  const int wave = tid >> 6, lane = tid & 63;
  {
    // the four channels of each of the three sums together (wave_sum4_d): row r of the wave ends up with channel {0, 2, 1, 3}[r]
    static_assert(HEAD_COMAX == 4, "head_fwd: packed wave sums");
    const double d0 = wave_sum4_d(spt[0], spt[1], spt[2], spt[3]), d1 = wave_sum4_d(sp[0], sp[1], sp[2], sp[3]),
                 d2 = wave_sum4_d(st[0], st[1], st[2], st[3]);
    const int co = classsum4_sel(lane);
    if ((lane & 15) == 0) { red[co * 3][wave] = d0; red[co * 3 + 1][wave] = d1; red[co * 3 + 2][wave] = d2; }
  }
  __syncthreads();
  if (tid < a.Co * 3) {
    const int co = tid / 3, k = tid - co * 3;
    a.partial[(((int64_t)b * a.Co + co) * a.rows + blockIdx.x) * 3 + k] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
  }
}

// The Dice sums of dice_finalize_kernel (elementwise.hip) from the partial rows of head_fwd_kernel, fixed-order sums.
// rows % 256 == 0 (every training size: rows = voxels / 1024) and BC <= 64: the [BC][rows][3] array is walked FLAT -- thread t takes
// the records t, t + 256, ... , eight of them requested before the first add -- and a (b, c) pair ends every rows / 256 records
// (uniform over the workgroup).  The one-wave-per-pair walk below paid one memory latency per 64 rows: 19 us at 128^3 (2048 rows),
// 4.6 us at 64^3, between the head's forward and backward passes.
__global__ __launch_bounds__(256) void head_dice_finalize_kernel(const double* __restrict__ partial, int rows, int BC, double smooth,
                                                                 double* __restrict__ sums, float* __restrict__ loss) {
  __shared__ double ratio[256];
  __shared__ double red[64][3][4];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  if ((rows & 255) == 0 && BC <= 64) {
    const int per = rows >> 8, M = BC * per;
    double s[3] = {0, 0, 0};
    int left = per, pair = 0;
    for (int m0 = 0; m0 < M; m0 += 8) {
      double v[8][3];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ok = m0 + u < M;
        const double* rec = partial + ((int64_t)(ok ? m0 + u : 0) * 256 + t) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) v[u][k] = rec[k];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (m0 + u >= M) break;
#pragma unroll
        for (int k = 0; k < 3; ++k) s[k] += v[u][k];
        if (--left == 0) {
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double w = wave_sum_d(s[k]);
            if (lane == 0) red[pair][k][wave] = w;
            s[k] = 0;
          }
          left = per; ++pair;
        }
      }
    }
    __syncthreads();
    double r = 0;
    if (t < BC) {
      double q[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) { q[k] = (red[t][k][0] + red[t][k][1]) + (red[t][k][2] + red[t][k][3]); sums[t * 3 + k] = q[k]; }
      r = (2.0 * q[0] + smooth) / (q[1] + q[2] + smooth);
    }
    ratio[t] = r;
    __syncthreads();
    if (t == 0) {
      double a = 0;
      for (int i = 0; i < BC; ++i) a += ratio[i];
      *loss = (float)(1.0 - a / BC);
    }
    return;
  }
  double acc = 0;
  for (int i = wave; i < BC; i += 4) {
    double s[3] = {0, 0, 0};
    for (int r = lane; r < rows; r += 64)
#pragma unroll
      for (int k = 0; k < 3; ++k) s[k] += partial[((int64_t)i * rows + r) * 3 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) s[k] = wave_sum_d(s[k]);
    if (lane == 0) {
      sums[i * 3] = s[0]; sums[i * 3 + 1] = s[1]; sums[i * 3 + 2] = s[2];
      acc += (2.0 * s[0] + smooth) / (s[1] + s[2] + smooth);
    }
  }
  ratio[t] = acc;
  __syncthreads();
  if (t == 0) {
    // (only lane 0 of each wave holds a value: the four adds below are the 256-entry sum without its zeros, bit for bit)
    double s = 0;
    for (int i = 0; i < 256; i += 64) s += ratio[i];
    *loss = (float)(1.0 - s / BC);
  }
}

struct HeadBwdArgs {
  const void* x; int64_t xld; int64_t N; int64_t xns, dxns; int qn;
  const float* w; const float* bias; const float* gate;
  const float* dp; int64_t dsb, dsc, dsv;
  const void* t; int64_t tsb, tsc, tsv; int t_quad;
  const double* sums; const float* dloss; double smooth; int BC;
  void* dx; int64_t dxld; int accumulate;
  float* partial; float* pbias; int chunks_per_sample; int Ci, Co;
};

// thread = voxel.  d logit = dp * p * (1 - p) with p recomputed from x (no saved activations are read);
// dx[ci] = gate[ci] * sum_co W[co][ci] * dlogit[co];  dW[co][ci] = gate[ci] * sum_v dlogit[co] * x[ci];  dbias[co] = sum_v dlogit[co]
template <typename TX, typename TD, int CIQ, int NCO, typename TT = float>     // NCO, TT: as head_fwd_kernel
__global__ __launch_bounds__(256) void head_bwd_kernel(HeadBwdArgs a) {
  N3D_CHAIN_PRIO();
  constexpr int CI = CIQ * 4, NV = HEAD_COMAX * CI + HEAD_COMAX;
  constexpr bool ALL = NCO != HEAD_COMAX;
  __shared__ float wsm[HEAD_COMAX][CI];
  __shared__ float gsm[CI];
  __shared__ float red[4][NV];
  const int b = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < HEAD_COMAX * CI; i += 256) {
    const int co = i / CI, ci = i - co * CI;
    wsm[co][ci] = co < a.Co ? a.w[co * CI + ci] * (a.gate ? a.gate[b * CI + ci] : 1.f) : 0.f;
  }
  if (tid < CI) gsm[tid] = a.gate ? a.gate[b * CI + tid] : 1.f;
  __syncthreads();
  float bias[NCO], k2[NCO], k0[NCO];
#pragma unroll
  for (int co = 0; co < NCO; ++co) {
    bias[co] = (ALL || co < a.Co) ? a.bias[co] : 0.f;
    k2[co] = k0[co] = 0.f;
    if (a.sums && (ALL || co < a.Co)) {
      // d loss / d p = -(1/BC) * (2 t den - num) / den^2   (loss.py:13-14), as n3d_dice_bwd
      const int i = b * a.Co + co;
      const double num = 2.0 * a.sums[i * 3] + a.smooth, den = a.sums[i * 3 + 1] + a.sums[i * 3 + 2] + a.smooth;
      const double gl = a.dloss ? (double)*a.dloss : 1.0;
      k2[co] = (float)(-gl / a.BC * 2.0 / den);
      k0[co] = (float)(gl / a.BC * num / (den * den));
    }
  }
  const TX* xb = reinterpret_cast<const TX*>(a.x) + (int64_t)b * a.N * a.xld;
  TD* dxb = reinterpret_cast<TD*>(a.dx) + (int64_t)b * a.N * a.dxld;
  int64_t qoff[CIQ], dqoff[CIQ];
#pragma unroll
  for (int q = 0; q < CIQ; ++q) { qoff[q] = head_quad_off(q, a.qn, a.xns); dqoff[q] = head_quad_off(q, a.qn, a.dxns); }
  float acc[HEAD_COMAX][CI], accb[HEAD_COMAX];     // (a channel that is not computed stays 0 and folds away)
#pragma unroll
  for (int co = 0; co < HEAD_COMAX; ++co) {
    accb[co] = 0.f;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) acc[co][ci] = 0.f;
  }
  const int64_t v0 = (int64_t)blockIdx.x * HEAD_CHUNK;
  // the four voxel rounds' input operands are requested before the first use (as head_fwd_kernel)
  constexpr int NK = HEAD_CHUNK / 256;
  constexpr bool TU8 = sizeof(TT) == 1;
  __shared__ uint32_t tql[TU8 ? NCO : 1][256];
  const bool tquad = TU8 && a.t_quad && a.sums;     // (uniform)
  float4 xqs[NK][CIQ];
  float gin[NK][NCO];
  uint32_t tq[NCO];
  if (tquad) head_target_quads_issue<NCO>(a.t, b * a.tsb + v0, a.tsc, tid, tq);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int64_t v = v0 + tid + k * 256;
    const int64_t vc = v < a.N ? v : v0;
#pragma unroll
    for (int q = 0; q < CIQ; ++q) xqs[k][q] = ld4(xb + vc * a.xld + qoff[q]);
    if (!tquad) {
#pragma unroll
      for (int co = 0; co < NCO; ++co) {
        gin[k][co] = 0.f;
        if (ALL || co < a.Co)
          gin[k][co] = a.sums ? (float)reinterpret_cast<const TT*>(a.t)[b * a.tsb + co * a.tsc + vc * a.tsv] : a.dp[b * a.dsb + co * a.dsc + vc * a.dsv];
      }
    }
  }
  if (tquad) head_target_quads_take<NCO, NK>(tq, tql, tid, gin);
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int64_t v = v0 + tid + k * 256;
    if (v >= a.N) break;
    const float4 (&xq)[CIQ] = xqs[k];
    float gp[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
      gp[co] = 0.f;
      if (ALL || co < a.Co) gp[co] = a.sums ? fmaf(k2[co], gin[k][co], k0[co]) : gin[k][co];
    }
    float4 prevq[CIQ];
    if (a.accumulate) {
#pragma unroll
      for (int q = 0; q < CIQ; ++q) prevq[q] = ld4(dxb + v * a.dxld + dqoff[q]);
    }
    float dl[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
      float s = bias[co];
#pragma unroll
      for (int q = 0; q < CIQ; ++q) {
        s = fmaf(wsm[co][q * 4 + 0], xq[q].x, s); s = fmaf(wsm[co][q * 4 + 1], xq[q].y, s);
        s = fmaf(wsm[co][q * 4 + 2], xq[q].z, s); s = fmaf(wsm[co][q * 4 + 3], xq[q].w, s);
      }
      const float pr = 1.0f / (1.0f + expf(-s));
      dl[co] = gp[co] * pr * (1.0f - pr);
      accb[co] += dl[co];
    }
#pragma unroll
    for (int q = 0; q < CIQ; ++q) {
      float o[4] = {0.f, 0.f, 0.f, 0.f};
      const float xe[4] = {xq[q].x, xq[q].y, xq[q].z, xq[q].w};
#pragma unroll
      for (int co = 0; co < NCO; ++co) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = fmaf(wsm[co][q * 4 + e], dl[co], o[e]);      // wsm already carries the gate
          acc[co][q * 4 + e] = fmaf(dl[co], xe[e], acc[co][q * 4 + e]);
        }
      }
      float4 ov = make_float4(o[0], o[1], o[2], o[3]);
      if (a.accumulate) { ov.x += prevq[q].x; ov.y += prevq[q].y; ov.z += prevq[q].z; ov.w += prevq[q].w; }
      st4(dxb + v * a.dxld + dqoff[q], ov);
    }
  }
  if (!a.partial) return;
  // block sums -> one partial slab [ci][co] (+ bias row) per workgroup; the common fixed-order finalize adds the slabs
  const int wave = tid >> 6, lane = tid & 63;
  // (four accumulators per wave sum, wave_classsum4_f: row r of the wave ends up with the total of value {0, 2, 1, 3}[r])
  static_assert(CI % 4 == 0 && HEAD_COMAX == 4, "head_bwd: packed wave sums");
  const int sel = classsum4_sel(lane);
  const bool wr = (lane & 15) == 0;
#pragma unroll
  for (int co = 0; co < NCO; ++co) {
#pragma unroll
    for (int ci = 0; ci < CI; ci += 4) {
      const float s = wave_classsum4_f<1>(acc[co][ci], acc[co][ci + 1], acc[co][ci + 2], acc[co][ci + 3]);
      if (wr) red[wave][(ci + sel) * HEAD_COMAX + co] = s;
    }
  }
  {
    const float sb = wave_classsum4_f<1>(accb[0], accb[1], accb[2], accb[3]);
    if (wr) red[wave][HEAD_COMAX * CI + sel] = sb;
  }
  __syncthreads();
  const int chunk = b * a.chunks_per_sample + blockIdx.x;
  for (int i = tid; i < NV; i += 256) {
    const float s = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    if (i < HEAD_COMAX * CI) {
      const int ci = i / HEAD_COMAX, co = i - ci * HEAD_COMAX;
      if (co < a.Co) a.partial[(int64_t)chunk * (CI * a.Co) + ci * a.Co + co] = s * gsm[ci];
    } else {
      const int co = i - HEAD_COMAX * CI;
      if (co < a.Co) a.pbias[(int64_t)chunk * a.Co + co] = s;
    }
  }
}

// (byte targets: the three-channel form only -- the reference's out_channels, the generator's three boolean maps)
template <typename TX>
static bool launch_head_fwd(const HeadFwdArgs& a, int B, bool t_u8, hipStream_t s) {
  const dim3 grid((unsigned)a.rows, (unsigned)B), blk(256);
  const bool three = a.Co == 3;
  if (t_u8 && !three) return false;
#define N3D_HEAD_FWD(Q_)                                                                                                     \
  case Q_: if (t_u8) hipLaunchKernelGGL((head_fwd_kernel<TX, Q_, 3, uint8_t>), grid, blk, 0, s, a);                           \
           else if (three) hipLaunchKernelGGL((head_fwd_kernel<TX, Q_, 3>), grid, blk, 0, s, a);                              \
           else hipLaunchKernelGGL((head_fwd_kernel<TX, Q_, HEAD_COMAX>), grid, blk, 0, s, a);                                \
           break;
  switch (a.Ci / 4) {
    N3D_HEAD_FWD(1) N3D_HEAD_FWD(2) N3D_HEAD_FWD(3) N3D_HEAD_FWD(4) N3D_HEAD_FWD(6) N3D_HEAD_FWD(8)
    default: return false;
  }
#undef N3D_HEAD_FWD
  return true;
}

template <typename TX, typename TD>
static bool launch_head_bwd(const HeadBwdArgs& a, int B, bool t_u8, hipStream_t s) {
  const dim3 grid((unsigned)a.chunks_per_sample, (unsigned)B), blk(256);
  const bool three = a.Co == 3;
  if (t_u8 && !three) return false;
#define N3D_HEAD_BWD(Q_)                                                                                                     \
  case Q_: if (t_u8) hipLaunchKernelGGL((head_bwd_kernel<TX, TD, Q_, 3, uint8_t>), grid, blk, 0, s, a);                       \
           else if (three) hipLaunchKernelGGL((head_bwd_kernel<TX, TD, Q_, 3>), grid, blk, 0, s, a);                          \
           else hipLaunchKernelGGL((head_bwd_kernel<TX, TD, Q_, HEAD_COMAX>), grid, blk, 0, s, a);                            \
           break;
  switch (a.Ci / 4) {
    N3D_HEAD_BWD(1) N3D_HEAD_BWD(2) N3D_HEAD_BWD(3) N3D_HEAD_BWD(4) N3D_HEAD_BWD(6) N3D_HEAD_BWD(8)
    default: return false;
  }
#undef N3D_HEAD_BWD
  return true;
}

static int check_head(const n3d_head* h, const char* what) {
  N3D_CHECK_ARG(h && h->x && h->w && h->bias && h->B > 0 && h->N > 0, "%s: bad args", what);
  const int q = h->Ci / 4;
  if (h->Ci % 4 != 0 || !(q == 1 || q == 2 || q == 3 || q == 4 || q == 6 || q == 8) || h->Co < 1 || h->Co > HEAD_COMAX)
    N3D_UNSUPPORTED("%s: Ci in {4,8,12,16,24,32} and Co <= %d are built (Ci=%d Co=%d)", what, HEAD_COMAX, h->Ci, h->Co);
  N3D_CHECK_ARG(h->x_dtype == N3D_F32 || h->x_dtype == N3D_BF16, "%s: unknown dtype %d", what, h->x_dtype);
  N3D_CHECK_ARG(h->t_dtype == N3D_F32 || h->t_dtype == N3D_U8, "%s: targets are N3D_F32 or N3D_U8 (t_dtype=%d)", what, h->t_dtype);
  if (h->t_dtype == N3D_U8 && h->Co != 3) N3D_UNSUPPORTED("%s: byte targets are built for Co = 3 (Co=%d)", what, h->Co);
  const int esz = h->x_dtype == N3D_BF16 ? 2 : 4;
  if (h->node_c) {
    N3D_CHECK_ARG(h->node_c > 0 && h->node_c % 4 == 0 && h->Ci % h->node_c == 0 && h->xld >= h->node_c && h->xld % 4 == 0 &&
                  h->x_node_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(h->x) % (4 * esz)) == 0,
                  "%s: node-planar x needs node_c %% 4 == 0, Ci %% node_c == 0, ld >= node_c, quad-aligned node stride", what);
    return 0;
  }
  N3D_CHECK_ARG(h->xld >= h->Ci && h->xld % 4 == 0 && (reinterpret_cast<uintptr_t>(h->x) % (4 * esz)) == 0, "%s: x needs ld %% 4 == 0 and quad alignment", what);
  return 0;
}

}  // namespace n3d

using namespace n3d;

extern "C" {

float n3d_dropout3d_uniform(uint64_t seed, uint32_t counter, uint32_t index) { return head_uniform(seed, counter, index); }

int n3d_dropout3d_gate(uint32_t* state, float p, int B, int C, float* gate, void* stream) {
  N3D_CHECK_ARG(state && gate && B > 0 && C > 0 && p >= 0.f && p < 1.f, "dropout3d_gate: bad args");
  hipLaunchKernelGGL(dropout3d_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, p, B * C, gate);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_head_rows(int64_t N) { return (int)cdiv(N, HEAD_CHUNK); }

size_t n3d_head_workspace_bytes(const n3d_head* h) {
  if (!h || h->N <= 0 || h->B <= 0) return 0;
  return (size_t)h->B * cdiv(h->N, HEAD_CHUNK) * ((size_t)h->Ci * h->Co + h->Co) * sizeof(float);
}

int n3d_head_fwd(const n3d_head* h, float* p, int64_t psb, int64_t psc, int64_t psv, float* logits, const void* t, int64_t tsb,
                 int64_t tsc, int64_t tsv, float smooth, double* partial, double* sums, float* loss, void* stream) {
  if (int e = check_head(h, "head_fwd")) return e;
  N3D_CHECK_ARG(p || t, "head_fwd: no output (p may be NULL only in the Dice mode: the trainers' step needs the loss alone)");
  N3D_CHECK_ARG(!t || (partial && sums && loss), "head_fwd: the Dice mode needs partial / sums / loss");
  HeadFwdArgs a;
  a.x = h->x; a.xld = h->xld; a.N = h->N; a.w = h->w; a.bias = h->bias; a.gate = h->gate;
  a.qn = h->node_c / 4; a.xns = h->x_node_stride;
  a.p = p; a.psb = psb; a.psc = psc; a.psv = psv; a.logits = logits;
  a.t = t; a.tsb = tsb; a.tsc = tsc; a.tsv = tsv; a.partial = t ? partial : nullptr; a.rows = (int)cdiv(h->N, HEAD_CHUNK);
  a.Ci = h->Ci; a.Co = h->Co;
  hipStream_t s = (hipStream_t)stream;
  const bool u8 = t && h->t_dtype == N3D_U8;
  a.t_quad = u8 && tsv == 1 && h->N % HEAD_CHUNK == 0 && tsb % 4 == 0 && tsc % 4 == 0 && reinterpret_cast<uintptr_t>(t) % 4 == 0;
  const bool ok = h->x_dtype == N3D_BF16 ? launch_head_fwd<bf16_t>(a, h->B, u8, s) : launch_head_fwd<float>(a, h->B, u8, s);
  if (!ok) N3D_UNSUPPORTED("head_fwd: Ci=%d", h->Ci);
  if (t) hipLaunchKernelGGL(head_dice_finalize_kernel, dim3(1), dim3(256), 0, s, partial, a.rows, h->B * h->Co, (double)smooth, sums, loss);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_head_bwd(const n3d_head* h, const float* dp, int64_t dsb, int64_t dsc, int64_t dsv, const void* t, int64_t tsb, int64_t tsc,
                 int64_t tsv, float smooth, const double* sums, const float* dloss, void* dx, int64_t dxld, int dx_dtype, int flags,
                 float* dw, float* dbias, void* ws, size_t ws_bytes, n3d_final_job* deferred, void* stream) {
  if (deferred) deferred->nchunks = 0;
  if (int e = check_head(h, "head_bwd")) return e;
  N3D_CHECK_ARG(dx && dxld >= (h->node_c ? h->node_c : h->Ci) && dxld % 4 == 0 && (!h->node_c || h->dx_node_stride % 4 == 0), "head_bwd: bad dx");
  N3D_CHECK_ARG((dp != nullptr) != (t != nullptr && sums != nullptr), "head_bwd: give either dp or (t, sums)");
  N3D_CHECK_ARG(dx_dtype == N3D_F32 || dx_dtype == N3D_BF16, "head_bwd: unknown dx dtype");
  const bool want_w = dw || dbias;
  const int cps = (int)cdiv(h->N, HEAD_CHUNK);
  const size_t need = n3d_head_workspace_bytes(h);
  if (want_w && (!ws || ws_bytes < need)) { set_error("head_bwd: workspace too small (%zu < %zu)", ws_bytes, need); return N3D_ERR_WORKSPACE; }
  HeadBwdArgs a;
  a.x = h->x; a.xld = h->xld; a.N = h->N; a.w = h->w; a.bias = h->bias; a.gate = h->gate;
  a.qn = h->node_c / 4; a.xns = h->x_node_stride; a.dxns = h->dx_node_stride;
  a.dp = dp; a.dsb = dsb; a.dsc = dsc; a.dsv = dsv; a.t = t; a.tsb = tsb; a.tsc = tsc; a.tsv = tsv;
  a.sums = dp ? nullptr : sums; a.dloss = dloss; a.smooth = (double)smooth; a.BC = h->B * h->Co;
  a.dx = dx; a.dxld = dxld; a.accumulate = (flags & N3D_ACCUMULATE) ? 1 : 0;
  const int nchunks = h->B * cps;
  a.partial = want_w ? (float*)ws : nullptr;
  a.pbias = want_w ? (float*)ws + (size_t)nchunks * h->Ci * h->Co : nullptr;
  a.chunks_per_sample = cps; a.Ci = h->Ci; a.Co = h->Co;
  hipStream_t s = (hipStream_t)stream;
  bool ok;
  const bool u8 = !dp && h->t_dtype == N3D_U8;
  a.t_quad = u8 && tsv == 1 && h->N % HEAD_CHUNK == 0 && tsb % 4 == 0 && tsc % 4 == 0 && reinterpret_cast<uintptr_t>(t) % 4 == 0;
  if (h->x_dtype == N3D_BF16) ok = dx_dtype == N3D_BF16 ? launch_head_bwd<bf16_t, bf16_t>(a, h->B, u8, s) : launch_head_bwd<bf16_t, float>(a, h->B, u8, s);
  else ok = dx_dtype == N3D_BF16 ? launch_head_bwd<float, bf16_t>(a, h->B, u8, s) : launch_head_bwd<float, float>(a, h->B, u8, s);
  if (!ok) N3D_UNSUPPORTED("head_bwd: Ci=%d", h->Ci);
  N3D_LAUNCH_CHECK();
  if (want_w) {
    // one "tile" holding the whole [Ci][Co] slab per chunk (layout of n3d_final_job: position = ci * co_t + co)
    n3d_final_job job;
    job.partial = a.partial; job.pbias = a.pbias; job.dw = dw; job.dbias = dbias; job.nchunks = nchunks; job.ntiles = 1; job.tci = 1; job.tco = 1;
    job.ci_t = h->Ci; job.co_t = h->Co; job.Co = h->Co; job.Ci = h->Ci; job.taps = 1; job.pad_ = 0;
    if (deferred) *deferred = job;
    else if (int e = n3d_wgrad_finalize_batch(&job, 1, stream)) return e;
  }
  return N3D_OK;
}

}  // extern "C"
