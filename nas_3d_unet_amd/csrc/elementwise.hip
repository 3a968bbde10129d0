// Bandwidth-bound kernels of the hot path: per-channel statistics, the normalise/activate/
// weighted-sum epilogue and its backward, SE gate, pooling, Dice, layout, Adam.
// All are HBM-bound streaming passes: 16-byte accesses, lane-consecutive addresses, >= 256 blocks
// where the tensor is large enough, partial rows + a tiny finalize kernel instead of atomics.
#include <type_traits>

#include "n3d_common.h"

namespace n3d {

// ------------------------------------------------------------------------------------------------
// block-level reduction of NV4 = 4*NV doubles per thread over the threads that share a channel quad;
// writes row[c*NV + k] for the block's sample.  vals are laid out [k][j] (k value kind, j channel in quad)
// ------------------------------------------------------------------------------------------------
template <int NV>
__device__ __forceinline__ void block_reduce_to_row(double (&vals)[NV * 4], int cpb, double* __restrict__ row,
                                                    double* lds /* [4 waves][cpb max 64][NV*4] */, bool wave_done = false) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  if (!wave_done) {
#pragma unroll
    for (int q = 0; q < NV * 4; ++q) vals[q] = wave_sum_strided(vals[q], cpb);
  }
  // the class of lane l (< cpb) in this wave is (wave*64 + l) % cpb
  if (lane < cpb) {
    const int c4 = (wave * 64 + lane) % cpb;
#pragma unroll
    for (int q = 0; q < NV * 4; ++q) lds[(wave * 64 + c4) * (NV * 4) + q] = vals[q];
  }
  __syncthreads();
  const int nq = cpb * NV * 4;
  for (int i = threadIdx.x; i < nq; i += blockDim.x) {
    const int c4 = i / (NV * 4), q = i % (NV * 4);
    double s = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += lds[(w * 64 + c4) * (NV * 4) + q];
    const int k = q / 4, j = q % 4;
    row[(c4 * 4 + j) * NV + k] = s;
  }
}

// The same for cpb a power of two <= 16 with the wave stage packed four values to a register (wave_classsum4_f): NT terms of
// NK kinds (s1, s2, sz / sum, sum of squares) x 4 channels each, X[NK t + kind]; ONE LDS round and one barrier for all terms.  The
// wave stage is fp32, the cross-wave sum fp64 as in block_reduce_to_row.  rows[t][(c4 * 4 + j) * NK + kind].
template <int NT, int NK, int cpb>
__device__ __forceinline__ void block_reduce_packed_rows(const float (&X)[NT * NK], double* const (&rows)[NT],
                                                         float* ldsf /* [4 waves][16][NT*NK][4] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15;
  if (l16 < cpb) {
    float* o = ldsf + ((wave * 16 + l16) * (NT * NK)) * 4 + classsum4_sel(lane);
#pragma unroll
    for (int q = 0; q < NT * NK; ++q) o[q * 4] = X[q];
  }
  __syncthreads();
  const int nq = cpb * NT * NK * 4;
  for (int i = threadIdx.x; i < nq; i += blockDim.x) {
    const int c4 = i / (NT * NK * 4), q = i % (NT * NK * 4);     // q = (t * NK + kind) * 4 + j
    double s = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += (double)ldsf[(w * 16 + c4) * (NT * NK * 4) + q];
    const int t = q / (NK * 4), kind = (q % (NK * 4)) / 4, j = q % 4;
    rows[t][(c4 * 4 + j) * NK + kind] = s;
  }
}

// NOTE on wave classes: when 64 % cpb != 0 the lanes < cpb of different waves hold different
// classes; the LDS slot is indexed by the class, and every class is present in every wave
// (cpb <= 64), so each (wave, class) slot is written exactly once.

// ------------------------------------------------------------------------------------------------
// channel statistics
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void channel_stats_body(const T* __restrict__ x, int64_t ld, int64_t N, int C, const EwMap& m,
                                                   double* __restrict__ stats) {
  __shared__ double lds[4 * 64 * 8];
  const int b = blockIdx.y;
  const int t = threadIdx.x;
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  const bool active = vl < m.vpb;
  const int64_t v0 = (int64_t)blockIdx.x * m.vpc;
  float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  if (active) {
    const T* xb = x + (int64_t)b * N * ld + c4 * 4;
    for (int it = 0; it < m.iters; ++it) {
      const int64_t v = v0 + (int64_t)it * m.vpb + vl;
      if (v < N) {
        const float4 q = ld4(xb + v * ld);
        s[0] += q.x; s[1] += q.y; s[2] += q.z; s[3] += q.w;
        ss[0] += q.x * q.x; ss[1] += q.y * q.y; ss[2] += q.z * q.z; ss[3] += q.w * q.w;
      }
    }
  }
  double* row = stats + ((int64_t)b * gridDim.x + blockIdx.x) * C * 2;
  if (is_pow2(m.cpb) && m.cpb <= 16) {
    class_dispatch16(m.cpb, [&](auto cc) {
      constexpr int CPB = decltype(cc)::value;
      const float X[2] = {wave_classsum4_f<CPB>(s[0], s[1], s[2], s[3]), wave_classsum4_f<CPB>(ss[0], ss[1], ss[2], ss[3])};
      double* const rows[1] = {row};
      block_reduce_packed_rows<1, 2, CPB>(X, rows, reinterpret_cast<float*>(lds));
    });
    return;
  }
  double vals[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) { vals[j] = s[j]; vals[4 + j] = ss[j]; }
  block_reduce_to_row<2>(vals, m.cpb, row, lds);
}
template <typename T>
__global__ __launch_bounds__(256) void channel_stats_kernel(const T* __restrict__ x, int64_t ld, int64_t N, int C,
                                                            EwMap m, double* __restrict__ stats) { N3D_CHAIN_PRIO();
  channel_stats_body(x, ld, N, C, m, stats);
}
// up to 8 tensors of one shape in one launch (grid.z = tensor): the inputs of a supernet node's primitives
struct StatsJobN { const float* x[8]; int64_t ld[8]; double* stats[8]; };
__global__ __launch_bounds__(256) void channel_statsN_kernel(StatsJobN js, int64_t N, int C, EwMap m) { N3D_CHAIN_PRIO();
  const float* x; int64_t ld; double* st;
  switch (blockIdx.z) {
    case 0: x = js.x[0]; ld = js.ld[0]; st = js.stats[0]; break; case 1: x = js.x[1]; ld = js.ld[1]; st = js.stats[1]; break;
    case 2: x = js.x[2]; ld = js.ld[2]; st = js.stats[2]; break; case 3: x = js.x[3]; ld = js.ld[3]; st = js.stats[3]; break;
    case 4: x = js.x[4]; ld = js.ld[4]; st = js.stats[4]; break; case 5: x = js.x[5]; ld = js.ld[5]; st = js.stats[5]; break;
    case 6: x = js.x[6]; ld = js.ld[6]; st = js.stats[6]; break; default: x = js.x[7]; ld = js.ld[7]; st = js.stats[7]; break;
  }
  channel_stats_body<float>(x, ld, N, C, m, st);
}

// sum partial rows: out[q] for q < ncol, executed by a whole 256-thread block; result in lds_out
__device__ __forceinline__ void reduce_rows(const double* __restrict__ rows, int nrows, int ncol, double* lds_part /*[256]*/,
                                            double* lds_out /*[ncol]*/) {
  const int t = threadIdx.x;
  const int nrl = 256 / ncol > 0 ? 256 / ncol : 1;
  // ncol <= 256 guaranteed by callers (C <= 64, NV <= 3 -> 192)
  const int q = t % ncol, rl = t / ncol;
  double s = 0;
  if (rl < nrl) {
    // a rolled "load; add" loop pays one memory latency per row: keep 8 independent loads in flight instead
    int r = rl;
    for (; r + 7 * nrl < nrows; r += 8 * nrl) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = rows[(int64_t)(r + u * nrl) * ncol + q];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < nrows; r += nrl) s += rows[(int64_t)r * ncol + q];
  }
  lds_part[t] = s;
  __syncthreads();
  if (t < ncol) {
    double a = 0;
    for (int r = 0; r < nrl; ++r) a += lds_part[r * ncol + t];
    lds_out[t] = a;
  }
  __syncthreads();
}

struct GnCoefArgs { const double* stats; int rows; const float* gamma; const float* beta; float* a; float* bb; float* mean_rstd; double* sumraw; };

__device__ __forceinline__ void gn_coeffs_body(const GnCoefArgs q, int C, int G, double count, float eps) {
  __shared__ double part[256];
  __shared__ double tot[192];
  __shared__ double mr[64 * 2];
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const float gam_t = (t < C) ? q.gamma[t] : 0.f, bet_t = (t < C) ? q.beta[t] : 0.f;  // issued before the row loads
  reduce_rows(q.stats + (int64_t)b * q.rows * C * 2, q.rows, C * 2, part, tot);
  const int cg = C / G;
  if (t < G) {
    double s = 0, ss = 0;
    for (int c = t * cg; c < (t + 1) * cg; ++c) { s += tot[c * 2]; ss += tot[c * 2 + 1]; }
    const double n = count * cg;
    const double mean = s / n;
    double var = ss / n - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    mr[t * 2] = mean; mr[t * 2 + 1] = rstd;
    if (q.mean_rstd) { q.mean_rstd[(b * G + t) * 2] = (float)mean; q.mean_rstd[(b * G + t) * 2 + 1] = (float)rstd; }
  }
  __syncthreads();
  if (t < C) {
    const int g = t / cg;
    const float rstd = (float)mr[g * 2 + 1], mean = (float)mr[g * 2];
    const float av = gam_t * rstd;
    q.a[b * C + t] = av;
    q.bb[b * C + t] = bet_t - mean * av;
    if (q.sumraw) q.sumraw[b * C + t] = tot[t * 2];
  }
}
// grid (B, terms): blockIdx.y selects the op (two ops of a searched-cell node share one launch)
__global__ __launch_bounds__(256) void gn_coeffs_kernel(GnCoefArgs q0, GnCoefArgs q1, int C, int G, double count, float eps) {
  N3D_CHAIN_PRIO();
  gn_coeffs_body(blockIdx.y ? q1 : q0, C, G, count, eps);
}
// up to N3D_MAX_GROUP_TERMS ops of a supernet node in one launch: grid (B, terms).  The descriptors live in the kernel
// arguments; the switch keeps every access to them statically indexed (a dynamic index would turn into dependent scalar loads)
#define N3D_PICK8(arr, i, dst)                                                                                     \
  switch (i) { case 0: dst = arr[0]; break; case 1: dst = arr[1]; break; case 2: dst = arr[2]; break; case 3: dst = arr[3]; break; \
               case 4: dst = arr[4]; break; case 5: dst = arr[5]; break; case 6: dst = arr[6]; break; default: dst = arr[7]; break; }
#define N3D_PICK16(arr, i, dst)                                                                                    \
  switch (i) { case 0: dst = arr[0]; break; case 1: dst = arr[1]; break; case 2: dst = arr[2]; break; case 3: dst = arr[3]; break; \
               case 4: dst = arr[4]; break; case 5: dst = arr[5]; break; case 6: dst = arr[6]; break; case 7: dst = arr[7]; break; \
               case 8: dst = arr[8]; break; case 9: dst = arr[9]; break; case 10: dst = arr[10]; break; case 11: dst = arr[11]; break; \
               case 12: dst = arr[12]; break; case 13: dst = arr[13]; break; case 14: dst = arr[14]; break; default: dst = arr[15]; break; }
struct GnCoefArgsN { GnCoefArgs q[8]; };
__global__ __launch_bounds__(256) void gn_coeffsN_kernel(GnCoefArgsN qs, int C, int G, double count, float eps) { N3D_CHAIN_PRIO();
  GnCoefArgs q;
  N3D_PICK8(qs.q, blockIdx.y, q);
  gn_coeffs_body(q, C, G, count, eps);
}

// ------------------------------------------------------------------------------------------------
// epilogue forward: out (+)= w * act(a*raw + b)
// ------------------------------------------------------------------------------------------------
template <bool RELU, bool ACC, typename T = float>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ raw, int64_t rld, const float* __restrict__ a,
                                                         const float* __restrict__ bb, const float* __restrict__ wptr,
                                                         T* __restrict__ out, int64_t old_, int64_t N, int C, EwMap m) {
  N3D_CHAIN_PRIO();
  const int b = blockIdx.y;
  const int t = threadIdx.x;
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  if (vl >= m.vpb) return;
  float4 av = make_float4(1, 1, 1, 1), bv = make_float4(0, 0, 0, 0);
  if (a) av = ld4(a + b * C + c4 * 4);
  if (bb) bv = ld4(bb + b * C + c4 * 4);
  const float w = wptr ? *wptr : 1.0f;
  const T* rb = raw + (int64_t)b * N * rld + c4 * 4;
  T* ob = out + (int64_t)b * N * old_ + c4 * 4;
  const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
#pragma unroll 4
  for (int it = 0; it < m.iters; ++it) {
    const int64_t v = v0 + (int64_t)it * m.vpb;
    if (v >= N) break;
    const float4 q = ld4(rb + v * rld);
    float4 z;
    z.x = fmaf(av.x, q.x, bv.x); z.y = fmaf(av.y, q.y, bv.y); z.z = fmaf(av.z, q.z, bv.z); z.w = fmaf(av.w, q.w, bv.w);
    if (RELU) { z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f); z.z = fmaxf(z.z, 0.f); z.w = fmaxf(z.w, 0.f); }
    T* op = ob + v * old_;
    if (ACC) {
      float4 o = ld4(op);
      o.x = fmaf(w, z.x, o.x); o.y = fmaf(w, z.y, o.y); o.z = fmaf(w, z.z, o.z); o.w = fmaf(w, z.w, o.w);
      st4(op, o);
    } else {
      z.x *= w; z.y *= w; z.z *= w; z.w *= w;
      st4(op, z);
    }
  }
}

// backward pass 1
template <bool RELU, typename T = float>
__global__ __launch_bounds__(256) void affine_bwd_reduce_kernel(const T* __restrict__ dout, int64_t dld, const T* __restrict__ raw,
                                                                int64_t rld, const float* __restrict__ a, const float* __restrict__ bb,
                                                                int64_t N, int C, EwMap m, double* __restrict__ sums) {
  N3D_CHAIN_PRIO();
  __shared__ double lds[4 * 64 * 12];
  const int b = blockIdx.y;
  const int t = threadIdx.x;
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  const bool active = vl < m.vpb;
  float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, sz[4] = {0, 0, 0, 0};
  if (active) {
    float av[4] = {1, 1, 1, 1}, bv[4] = {0, 0, 0, 0};
    if (a) { const float4 q = ld4(a + b * C + c4 * 4); av[0] = q.x; av[1] = q.y; av[2] = q.z; av[3] = q.w; }
    if (bb) { const float4 q = ld4(bb + b * C + c4 * 4); bv[0] = q.x; bv[1] = q.y; bv[2] = q.z; bv[3] = q.w; }
    const T* db = dout + (int64_t)b * N * dld + c4 * 4;
    const T* rb = raw + (int64_t)b * N * rld + c4 * 4;
    const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
    for (int it0 = 0; it0 < m.iters; it0 += 4) {
      float4 dq[4], rq[4];
      bool ok[4];
      // request up to 8 independent 16-byte loads before the first use
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t v = v0 + (int64_t)(it0 + u) * m.vpb;
        ok[u] = (it0 + u < m.iters) && v < N;
        const int64_t vc = ok[u] ? v : 0;
        dq[u] = ld4(db + vc * dld);
        rq[u] = ld4(rb + vc * rld);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
        const float d[4] = {dq[u].x, dq[u].y, dq[u].z, dq[u].w};
        const float r[4] = {rq[u].x, rq[u].y, rq[u].z, rq[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float z = fmaf(av[j], r[j], bv[j]);
          float g = d[j];
          if (RELU) { g = z > 0.f ? g : 0.f; z = fmaxf(z, 0.f); }
          s1[j] += g;
          s2[j] = fmaf(g, r[j], s2[j]);
          sz[j] = fmaf(d[j], z, sz[j]);
        }
      }
    }
  }
  // the wave-level stage runs in fp32 (<= 64 lanes x <= 4 partials each), the cross-wave / cross-row stages in fp64
  double* row = sums + ((int64_t)b * gridDim.x + blockIdx.x) * C * 3;
  if (is_pow2(m.cpb) && m.cpb <= 16) {
    class_dispatch16(m.cpb, [&](auto cc) {
      constexpr int CPB = decltype(cc)::value;
      const float X[3] = {wave_classsum4_f<CPB>(s1[0], s1[1], s1[2], s1[3]), wave_classsum4_f<CPB>(s2[0], s2[1], s2[2], s2[3]),
                          wave_classsum4_f<CPB>(sz[0], sz[1], sz[2], sz[3])};
      double* const rows[1] = {row};
      block_reduce_packed_rows<1, 3, CPB>(X, rows, reinterpret_cast<float*>(lds));
    });
    return;
  }
  double vals[12];
  if (is_pow2(m.cpb)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vals[j] = wave_classsum_f(s1[j], m.cpb); vals[4 + j] = wave_classsum_f(s2[j], m.cpb); vals[8 + j] = wave_classsum_f(sz[j], m.cpb);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) { vals[j] = s1[j]; vals[4 + j] = s2[j]; vals[8 + j] = sz[j]; }
  }
  block_reduce_to_row<3>(vals, m.cpb, row, lds, is_pow2(m.cpb));
}

// GroupNorm backward coefficients.  One workgroup of 256*BP threads: thread (bl = tid/256, t = tid%256) works on
// sample b = b0 + bl, so up to BP = 4 samples go through the dependent-load chain (rows -> group sums ->
// coefficients) side by side instead of one after the other; per-channel sums over samples meet in LDS.
#define GNB_BP 4
__device__ __forceinline__ void reduce_rows_b(const double* __restrict__ rows, int nrows, int ncol, double* lds_part /*[256]*/,
                                              double* lds_out /*[ncol]*/, int t, bool active) {
  const int nrl = 256 / ncol > 0 ? 256 / ncol : 1;
  const int q = t % ncol, rl = t / ncol;
  double s = 0;
  if (active && rl < nrl) {
    int r = rl;
    for (; r + 7 * nrl < nrows; r += 8 * nrl) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = rows[(int64_t)(r + u * nrl) * ncol + q];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < nrows; r += nrl) s += rows[(int64_t)r * ncol + q];
  }
  lds_part[t] = s;
  __syncthreads();
  if (t < ncol) {
    double a = 0;
    for (int r = 0; r < nrl; ++r) a += lds_part[r * ncol + t];
    lds_out[t] = a;
  }
  __syncthreads();
}

struct GnBwdCoefArgs {
  const double* sums; int rows; const float* gamma; const float* mean_rstd; const float* wptr; float* dgamma; float* dbeta; float* dalpha;
  float* A; float* Bc; float* Cc; const double* sumraw; float* dbias_conv;
};

// BP: samples side by side = 256-thread slices of the workgroup (2 for the batch-2 steps of the benchmark configurations: half the
// waves to launch and to meet at the barriers; 4 for larger batches)
template <int BP>
__device__ __forceinline__ void gn_bwd_coeffs_body(const GnBwdCoefArgs q, int B, int C, int G, double count) {
  const double* __restrict__ sums = q.sums; const int rows = q.rows; const float* __restrict__ gamma = q.gamma;
  const float* __restrict__ mean_rstd = q.mean_rstd; const float* __restrict__ wptr = q.wptr;
  float* __restrict__ dgamma = q.dgamma; float* __restrict__ dbeta = q.dbeta; float* __restrict__ dalpha = q.dalpha;
  float* __restrict__ A = q.A; float* __restrict__ Bc = q.Bc; float* __restrict__ Cc = q.Cc;
  const double* __restrict__ sumraw = q.sumraw; float* __restrict__ dbias_conv = q.dbias_conv;
  __shared__ double part[BP][256];
  __shared__ double tot[BP][192];
  __shared__ double gc[BP][64 * 2];
  __shared__ double acc4[BP][4][64];  // per-sample dgamma, dbeta, dz, dbias contributions
  const int bl = threadIdx.x >> 8, t = threadIdx.x & 255;
  const int cg = C / G;
  const double w = wptr ? (double)*wptr : 1.0;
  // independent loads first: they overlap the row reductions
  const double gam = (t < C) ? (double)gamma[t] : 0.0;
  double dg = 0, db = 0, dz = 0, dbc = 0;
  for (int b0 = 0; b0 < B; b0 += BP) {
    const int b = b0 + bl;
    const bool act = b < B;
    const int bb = act ? b : B - 1;
    const int gq = (t < C) ? t / cg : 0;
    const double mean_c = mean_rstd[(bb * G + gq) * 2], rstd_c = mean_rstd[(bb * G + gq) * 2 + 1];
    const double sraw = (dbias_conv && t < C) ? sumraw[bb * C + t] : 0.0;
    reduce_rows_b(sums + (int64_t)bb * rows * C * 3, rows, C * 3, part[bl], tot[bl], t, act);
    if (t < G) {
      const double mean = mean_rstd[(bb * G + t) * 2], rstd = mean_rstd[(bb * G + t) * 2 + 1];
      double c1 = 0, c2 = 0;
      for (int c = t * cg; c < (t + 1) * cg; ++c) {
        const double S1 = tot[bl][c * 3], S2 = tot[bl][c * 3 + 1];
        const double gm = (double)gamma[c];
        c1 += gm * w * S1;
        c2 += gm * w * rstd * (S2 - mean * S1);
      }
      const double n = count * cg;
      gc[bl][t * 2] = c1 / n; gc[bl][t * 2 + 1] = c2 / n;
    }
    __syncthreads();
    if (t < C && act) {
      const double S1 = tot[bl][t * 3], S2 = tot[bl][t * 3 + 1], Sz = tot[bl][t * 3 + 2];
      dg += w * rstd_c * (S2 - mean_c * S1);
      db += w * S1;
      dz += Sz;
      const double c1 = gc[bl][gq * 2], c2 = gc[bl][gq * 2 + 1];
      const double Av = rstd_c * gam * w, Bv = -rstd_c * c1 + rstd_c * rstd_c * c2 * mean_c, Cv = -rstd_c * rstd_c * c2;
      A[b * C + t] = (float)Av;
      Bc[b * C + t] = (float)Bv;
      Cc[b * C + t] = (float)Cv;
      if (dbias_conv) dbc += Av * S1 + count * Bv + Cv * sraw;
    }
    __syncthreads();
  }
  if (t < 64) { acc4[bl][0][t] = (t < C) ? dg : 0.0; acc4[bl][1][t] = (t < C) ? db : 0.0; acc4[bl][2][t] = (t < C) ? dz : 0.0; acc4[bl][3][t] = (t < C) ? dbc : 0.0; }
  __syncthreads();
  if (bl == 0 && t < C) {
    double a0 = 0, a1 = 0, a3 = 0;
    for (int k = 0; k < BP; ++k) { a0 += acc4[k][0][t]; a1 += acc4[k][1][t]; a3 += acc4[k][3][t]; }
    if (dgamma) dgamma[t] = (float)a0;
    if (dbeta) dbeta[t] = (float)a1;
    if (dbias_conv) dbias_conv[t] = (float)a3;
  }
  if (dalpha && threadIdx.x < 64) {
    // (one wave: per-channel totals over the samples, then a butterfly over the 64 lanes -- a fixed tree instead of 64 BP serial adds)
    double sdz = 0;
    for (int k = 0; k < BP; ++k) sdz += acc4[k][2][threadIdx.x];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sdz += __shfl_xor(sdz, o, 64);
    if (threadIdx.x == 0) *dalpha = (float)sdz;
  }
}
// grid (terms): blockIdx.x selects the op
template <int BP>
__global__ __launch_bounds__(256 * BP) void gn_bwd_coeffs_kernel(GnBwdCoefArgs q0, GnBwdCoefArgs q1, int B, int C, int G, double count) {
  N3D_CHAIN_PRIO();
  gn_bwd_coeffs_body<BP>(blockIdx.x ? q1 : q0, B, C, G, count);
}
struct GnBwdCoefArgsN { GnBwdCoefArgs q[8]; };
template <int BP>
__global__ __launch_bounds__(256 * BP) void gn_bwd_coeffsN_kernel(GnBwdCoefArgsN qs, int B, int C, int G, double count) { N3D_CHAIN_PRIO();
  GnBwdCoefArgs q;
  N3D_PICK8(qs.q, blockIdx.x, q);
  gn_bwd_coeffs_body<BP>(q, B, C, G, count);
}

__device__ __forceinline__ void plain_bwd_coeffs_body(const double* __restrict__ sums, int rows, const float* __restrict__ wptr,
                                                      int B, int C, float* __restrict__ dalpha, float* __restrict__ A) {
  __shared__ double part[256];
  __shared__ double tot[192];
  __shared__ double zred[64];
  const int t = threadIdx.x;
  const float w = wptr ? *wptr : 1.0f;
  double dz = 0;
  for (int b = 0; b < B; ++b) {
    if (dalpha) {
      reduce_rows(sums + (int64_t)b * rows * C * 3, rows, C * 3, part, tot);
      if (t < C) dz += tot[t * 3 + 2];
      __syncthreads();
    }
    if (t < C && A) A[b * C + t] = w;
  }
  if (dalpha) {
    if (t < 64) zred[t] = (t < C) ? dz : 0.0;
    __syncthreads();
    if (t < 64) {
      double s = zred[t];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
      if (t == 0) *dalpha = (float)s;
    }
  }
}
__global__ __launch_bounds__(256) void plain_bwd_coeffs_kernel(const double* __restrict__ sums, int rows, const float* __restrict__ wptr,
                                                               int B, int C, float* __restrict__ dalpha, float* __restrict__ A) {
  plain_bwd_coeffs_body(sums, rows, wptr, B, C, dalpha, A);
}
struct PlainCoefArgs { const double* sums; int rows; const float* wptr; float* dalpha; float* A; };
struct PlainCoefArgsN { PlainCoefArgs q[8]; };
__global__ __launch_bounds__(256) void plain_bwd_coeffsN_kernel(PlainCoefArgsN qs, int B, int C) {
  PlainCoefArgs q;
  N3D_PICK8(qs.q, blockIdx.x, q);
  plain_bwd_coeffs_body(q.sums, q.rows, q.wptr, B, C, q.dalpha, q.A);
}

// backward pass 2
template <bool RELU, bool ACC, typename T = float>
__global__ __launch_bounds__(256) void affine_bwd_apply_kernel(const T* __restrict__ dout, int64_t dld, const T* __restrict__ raw,
                                                               int64_t rld, const float* __restrict__ a, const float* __restrict__ bb,
                                                               const float* __restrict__ A, const float* __restrict__ Bc,
                                                               const float* __restrict__ Cc, T* __restrict__ draw, int64_t drld,
                                                               int64_t N, int C, EwMap m) {
  N3D_CHAIN_PRIO();
  const int b = blockIdx.y;
  const int t = threadIdx.x;
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  if (vl >= m.vpb) return;
  float av[4] = {1, 1, 1, 1}, bv[4] = {0, 0, 0, 0}, Av[4] = {1, 1, 1, 1}, Bv[4] = {0, 0, 0, 0}, Cv[4] = {0, 0, 0, 0};
  const int co = b * C + c4 * 4;
  if (a) { const float4 q = ld4(a + co); av[0] = q.x; av[1] = q.y; av[2] = q.z; av[3] = q.w; }
  if (bb) { const float4 q = ld4(bb + co); bv[0] = q.x; bv[1] = q.y; bv[2] = q.z; bv[3] = q.w; }
  if (A) { const float4 q = ld4(A + co); Av[0] = q.x; Av[1] = q.y; Av[2] = q.z; Av[3] = q.w; }
  if (Bc) { const float4 q = ld4(Bc + co); Bv[0] = q.x; Bv[1] = q.y; Bv[2] = q.z; Bv[3] = q.w; }
  if (Cc) { const float4 q = ld4(Cc + co); Cv[0] = q.x; Cv[1] = q.y; Cv[2] = q.z; Cv[3] = q.w; }
  const T* db = dout + (int64_t)b * N * dld + c4 * 4;
  const T* rb = raw + (int64_t)b * N * rld + c4 * 4;
  T* ob = draw + (int64_t)b * N * drld + c4 * 4;
  const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
#pragma unroll 2
  for (int it = 0; it < m.iters; ++it) {
    const int64_t v = v0 + (int64_t)it * m.vpb;
    if (v >= N) break;
    const float4 dq = ld4(db + v * dld);
    const float4 rq = ld4(rb + v * rld);
    const float d[4] = {dq.x, dq.y, dq.z, dq.w};
    const float r[4] = {rq.x, rq.y, rq.z, rq.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g = d[j];
      if (RELU) { const float z = fmaf(av[j], r[j], bv[j]); g = z > 0.f ? g : 0.f; }
      o[j] = fmaf(Av[j], g, fmaf(Cv[j], r[j], Bv[j]));
    }
    T* op = ob + v * drld;
    if (ACC) { const float4 p = ld4(op); o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w; }
    st4(op, make_float4(o[0], o[1], o[2], o[3]));
  }
}


// ------------------------------------------------------------------------------------------------
// Fused-prologue variants for small tensors (the deep U-net levels, where a step is launch-latency bound):
// every workgroup recomputes its sample's GroupNorm coefficients from the partial-statistics rows in its
// prologue (a few hundred doubles), so the separate coefficient kernel -- one launch and one dependent
// memory round trip per op -- disappears.  Workgroup (0, b) also stores the coefficients for the backward pass.
// ------------------------------------------------------------------------------------------------
template <bool RELU, bool ACC, typename T = float>
__global__ __launch_bounds__(256) void affine_act_gn_kernel(const T* __restrict__ raw, int64_t rld, const double* __restrict__ stats, int rows,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, int G, double count,
                                                            float eps, const float* __restrict__ wptr, T* __restrict__ out, int64_t old_,
                                                            int64_t N, int C, EwMap m, float* __restrict__ a_out, float* __restrict__ b_out,
                                                            float* __restrict__ mr_out, double* __restrict__ sumraw) {
  N3D_CHAIN_PRIO();
  __shared__ double part[256];
  __shared__ double tot[128];
  __shared__ float ab[2][64];
  __shared__ __attribute__((aligned(16))) float abw[4][2][64];
  const int b = blockIdx.y;
  const int t = threadIdx.x;
  const float gam_t = (t < C) ? gamma[t] : 0.f, bet_t = (t < C) ? beta[t] : 0.f;
  const float w = wptr ? *wptr : 1.0f;
  // the first iteration's operands are requested before the prologue: its memory round trip overlaps the
  // coefficient computation instead of following it
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  const T* rb = raw + (int64_t)b * N * rld + c4 * 4;
  T* ob = out + (int64_t)b * N * old_ + c4 * 4;
  const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
  const bool act0 = vl < m.vpb && v0 < N;
  float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), o0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (act0) {
    q0 = ld4(rb + v0 * rld);
    if (ACC) o0 = ld4(ob + v0 * old_);
  }
  const int cg = C / G;
  float4 av, bv;
  if (is_pow2(C) && cg <= 16) {
    // wave-level prologue, no workgroup barrier: each wave redundantly sums the partial rows (lane = row slot x
    // channel), folds them with DPP / permlane sums, derives its group's mean and rstd, and hands the per-channel
    // coefficients to its own lanes through a private LDS strip (LDS operations of one wave execute in order).
    const int lane = t & 63, wave = t >> 6;
    const int c = lane & (C - 1), rs = lane / C, nslots = 64 / C;
    const float gam = gamma[c], bet = beta[c];
    const double2* st2 = reinterpret_cast<const double2*>(stats + (int64_t)b * rows * C * 2);
    double s = 0, ss = 0;
    {
      int r = rs;
      for (; r + 3 * nslots < rows; r += 4 * nslots) {
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = st2[(int64_t)(r + u * nslots) * C + c];
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += v[u].x; ss += v[u].y; }
      }
      for (; r < rows; r += nslots) { const double2 v = st2[(int64_t)r * C + c]; s += v.x; ss += v.y; }
    }
    double gv[2] = {s, ss};
    wave_classsum_dn<2>(gv, C);
    s = gv[0];
    wave_groupsum_dn<2>(gv, cg);
    const double gs = gv[0], gss = gv[1];
    const double n = count * cg;
    const double mean = gs / n;
    double var = gss / n - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const float a1 = gam * (float)rstd;
    const float b1 = bet - (float)mean * a1;
    if (lane < C) { abw[wave][0][lane] = a1; abw[wave][1][lane] = b1; }
    if (blockIdx.x == 0 && wave == 0 && lane < C) {
      a_out[b * C + c] = a1; b_out[b * C + c] = b1;
      if (sumraw) sumraw[b * C + c] = s;
      if (c % cg == 0) { const int g = c / cg; mr_out[(b * G + g) * 2] = (float)mean; mr_out[(b * G + g) * 2 + 1] = (float)rstd; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!act0) return;
    av = ld4(&abw[wave][0][c4 * 4]);
    bv = ld4(&abw[wave][1][c4 * 4]);
  } else {
    reduce_rows(stats + (int64_t)b * rows * C * 2, rows, C * 2, part, tot);
    if (t < C) {
      const int g = t / cg;
      double s = 0, ss = 0;
      for (int c = g * cg; c < (g + 1) * cg; ++c) { s += tot[c * 2]; ss += tot[c * 2 + 1]; }
      const double n = count * cg;
      const double mean = s / n;
      double var = ss / n - mean * mean;
      if (var < 0) var = 0;
      const double rstd = 1.0 / sqrt(var + (double)eps);
      const float a1 = gam_t * (float)rstd;
      const float b1 = bet_t - (float)mean * a1;
      ab[0][t] = a1; ab[1][t] = b1;
      if (blockIdx.x == 0) {
        a_out[b * C + t] = a1; b_out[b * C + t] = b1;
        if (sumraw) sumraw[b * C + t] = tot[t * 2];
        if (t % cg == 0) { mr_out[(b * G + g) * 2] = (float)mean; mr_out[(b * G + g) * 2 + 1] = (float)rstd; }
      }
    }
    __syncthreads();
    if (!act0) return;
    av = make_float4(ab[0][c4 * 4], ab[0][c4 * 4 + 1], ab[0][c4 * 4 + 2], ab[0][c4 * 4 + 3]);
    bv = make_float4(ab[1][c4 * 4], ab[1][c4 * 4 + 1], ab[1][c4 * 4 + 2], ab[1][c4 * 4 + 3]);
  }
  // iterations 1.. are requested before iteration 0 is finished (loads first, then math and stores)
  constexpr int PF = 3;
  float4 qn[PF], on[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int64_t v = v0 + (int64_t)(i + 1) * m.vpb;
    const bool ok = (i + 1) < m.iters && v < N;
    const int64_t vc = ok ? v : v0;
    qn[i] = ld4(rb + vc * rld);
    if (ACC) on[i] = ld4(ob + vc * old_);
  }
  auto emit = [&](int64_t v, const float4 q, float4 o) {
    float4 z;
    z.x = fmaf(av.x, q.x, bv.x); z.y = fmaf(av.y, q.y, bv.y); z.z = fmaf(av.z, q.z, bv.z); z.w = fmaf(av.w, q.w, bv.w);
    if (RELU) { z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f); z.z = fmaxf(z.z, 0.f); z.w = fmaxf(z.w, 0.f); }
    T* op = ob + v * old_;
    if (ACC) {
      o.x = fmaf(w, z.x, o.x); o.y = fmaf(w, z.y, o.y); o.z = fmaf(w, z.z, o.z); o.w = fmaf(w, z.w, o.w);
      st4(op, o);
    } else {
      z.x *= w; z.y *= w; z.z *= w; z.w *= w;
      st4(op, z);
    }
  };
  emit(v0, q0, o0);
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int64_t v = v0 + (int64_t)(i + 1) * m.vpb;
    if ((i + 1) < m.iters && v < N) emit(v, qn[i], on[i]);
  }
  // further iterations (tensors beyond 1024 rows x 4 iterations per sample) go four at a time, loads first
  for (int it = PF + 1; it < m.iters; it += 4) {
    float4 q[4], o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t v = v0 + (int64_t)(it + j) * m.vpb;
      const int64_t vc = ((it + j) < m.iters && v < N) ? v : v0;
      q[j] = ld4(rb + vc * rld);
      o[j] = ACC ? ld4(ob + vc * old_) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t v = v0 + (int64_t)(it + j) * m.vpb;
      if ((it + j) < m.iters && v < N) emit(v, q[j], o[j]);
    }
  }
}

// backward pass 2 with the GroupNorm-backward coefficients computed in the prologue (per workgroup, for its sample);
// workgroup (0,0) additionally covers all samples (B <= 4) to emit dgamma / dbeta / dalpha / conv-bias gradient.
// All global loads of the prologue (parameter vectors and every partial row this workgroup needs) are issued
// before the first use, so the prologue costs ONE memory round trip; everything after it is LDS work.
#define GNF_MAXB 4
struct GnBwdTerm {
  const float* raw; int64_t rld; const float* a; const float* b; const double* sums; int rows; const float* gamma; const float* mean_rstd;
  const float* wptr; const double* sumraw; float* draw; int64_t drld; float* dgamma; float* dbeta; float* dalpha; float* dbias_conv; int relu;
  const float* cA; const float* cB; const float* cC;   // PRE variant: coefficients from n3d_gn_bwd_coeffs2
};

// GroupNorm-backward coefficients of BOTH terms of a pair, wave level (see affine_bwd_apply_gn_kernel); strip[k][0..2][c] = A, B, C.
// Everything a wave needs from memory -- the partial rows of the two terms (same row count: they come out of one reduce2 launch),
// statistics, parameters -- is requested before any of the arithmetic.  That matters most for the LEADER wave (workgroup (0,0),
// wave 0), which also forms the parameter gradients and therefore needs the rows of ALL samples: these tensors were written by the
// previous launch on other XCDs, a round trip is ~2 us, the leader used to make one per sample, and the launch is over when its
// slowest workgroup is.  NB = samples this wave handles (1, or GNF_MAXB for the leader with the unused ones predicated off), DEPTH =
// rows per lane requested together.
template <int NT, int NB, int DEPTH>
__device__ __forceinline__ void gn_bwd_prologue_impl(const GnBwdTerm& t0, const GnBwdTerm& t1, const int B, const int C, const int G,
                                                     const double count, float (*strip)[3][64]) {
  constexpr bool LEAD = NB > 1;
  const int lane = threadIdx.x & 63;
  const int cg = C / G;
  const int c = lane & (C - 1), rs = lane / C, nslots = 64 / C;
  const int gq = c / cg;
  const GnBwdTerm* ts[2] = {&t0, &t1};
  const int nbw = LEAD ? B : 1;
  const int rows = t0.rows;           // == t1.rows (NT == 1: t1 is t0)
  double w[NT], gam[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) { w[k] = ts[k]->wptr ? (double)*ts[k]->wptr : 1.0; gam[k] = (double)ts[k]->gamma[c]; }
  int bs[NB];
  double mn[NB][NT], rsd[NB][NT], pf[NB][NT], acc[NB][NT][3];
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    bs[kb] = LEAD ? ((int)blockIdx.y + 1 + (kb < nbw ? kb : 0)) % B : (int)blockIdx.y;   // own sample last (kb == nbw - 1)
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      mn[kb][k] = ts[k]->mean_rstd[(bs[kb] * G + gq) * 2]; rsd[kb][k] = ts[k]->mean_rstd[(bs[kb] * G + gq) * 2 + 1];
      pf[kb][k] = (LEAD && ts[k]->dbias_conv) ? ts[k]->sumraw[bs[kb] * C + c] : 0.0;
      acc[kb][k][0] = acc[kb][k][1] = acc[kb][k][2] = 0.0;
    }
  }
  for (int r = rs; r < rows; r += DEPTH * nslots) {
    double v[NB][NT][DEPTH][3];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
          const int rr = r + u * nslots;
          const double* p = ts[k]->sums + (((int64_t)bs[kb] * rows + (rr < rows ? rr : rs)) * C + c) * 3;
#pragma unroll
          for (int q = 0; q < 3; ++q) v[kb][k][u][q] = p[q];
        }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
          if (r + u * nslots < rows) {
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[kb][k][q] += v[kb][k][u][q];
          }
  }
  double dg[NT], db[NT], dz[NT], dbc[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) dg[k] = db[k] = dz[k] = dbc[k] = 0.0;
#pragma unroll
  for (int kb = 0; kb < NB; ++kb) {
    if (kb < nbw) {
      double cs[NT * 2], gsum[NT * 2];
#pragma unroll
      for (int k = 0; k < NT; ++k) { cs[2 * k] = acc[kb][k][0]; cs[2 * k + 1] = acc[kb][k][1]; }
      wave_classsum_dn<NT * 2>(cs, C);
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        gsum[2 * k] = gam[k] * w[k] * cs[2 * k];
        gsum[2 * k + 1] = gam[k] * w[k] * rsd[kb][k] * (cs[2 * k + 1] - mn[kb][k] * cs[2 * k]);
      }
      wave_groupsum_dn<NT * 2>(gsum, cg);
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const GnBwdTerm& t = *ts[k];
        const double S1 = cs[2 * k], S2 = cs[2 * k + 1];
        double Sz = acc[kb][k][2];
        if (LEAD && t.dalpha) Sz = wave_classsum_d(Sz, C);
        const double n = count * cg;
        const double c1 = gsum[2 * k] / n;
        const double c2 = gsum[2 * k + 1] / n;
        const double A1 = rsd[kb][k] * gam[k] * w[k], B1 = -rsd[kb][k] * c1 + rsd[kb][k] * rsd[kb][k] * c2 * mn[kb][k],
                     C1 = -rsd[kb][k] * rsd[kb][k] * c2;
        if (kb == nbw - 1 && lane < C) { strip[k][0][lane] = (float)A1; strip[k][1][lane] = (float)B1; strip[k][2][lane] = (float)C1; }
        if (LEAD) {
          dg[k] += w[k] * rsd[kb][k] * (S2 - mn[kb][k] * S1);
          db[k] += w[k] * S1;
          dz[k] += Sz;
          if (t.dbias_conv) dbc[k] += A1 * S1 + count * B1 + C1 * pf[kb][k];
        }
      }
    }
  }
  if (LEAD) {
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const GnBwdTerm& t = *ts[k];
      if (lane < C) {
        if (t.dgamma) t.dgamma[c] = (float)dg[k];
        if (t.dbeta) t.dbeta[c] = (float)db[k];
        if (t.dbias_conv) t.dbias_conv[c] = (float)dbc[k];
      }
      if (t.dalpha) {
        const double sdz = wave_sum_d(lane < C ? dz[k] : 0.0);
        if (lane == 0) *t.dalpha = (float)sdz;
      }
    }
  }
}
__device__ __forceinline__ void gn_bwd_prologue_wave2(const GnBwdTerm& t0, const GnBwdTerm& t1, const int B, const int C, const int G,
                                                      const double count, const bool lead_w, float (*strip)[3][64]) {
  if (lead_w) {
    if (B <= 2) gn_bwd_prologue_impl<2, 2, 4>(t0, t1, B, C, G, count, strip);   // (no slots predicated off: half the requests)
    else gn_bwd_prologue_impl<2, GNF_MAXB, 2>(t0, t1, B, C, G, count, strip);
  } else gn_bwd_prologue_impl<2, 1, 4>(t0, t1, B, C, G, count, strip);
}

template <bool RELU, bool ACC, typename T = float>
__global__ __launch_bounds__(256) void affine_bwd_apply_gn_kernel(const T* __restrict__ dout, int64_t dld, const T* __restrict__ raw,
                                                                  int64_t rld, const float* __restrict__ a, const float* __restrict__ bb,
                                                                  const double* __restrict__ sums, int rows, const float* __restrict__ gamma,
                                                                  const float* __restrict__ mean_rstd, const float* __restrict__ wptr,
                                                                  const double* __restrict__ sumraw, int B, int G, double count,
                                                                  T* __restrict__ draw, int64_t drld, int64_t N, int C, EwMap m,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dalpha,
                                                                  float* __restrict__ dbias_conv) {
  N3D_CHAIN_PRIO();
  __shared__ double part[256];
  __shared__ double tot[192];
  __shared__ double gc[64 * 2];
  __shared__ double gsh[64];
  __shared__ double zred[64];
  __shared__ float coef[3][64];
  __shared__ __attribute__((aligned(16))) float coefw[4][3][64];
  const int t = threadIdx.x;
  const int cg = C / G;
  const bool wave_path = is_pow2(C) && cg <= 16;
  if (wave_path && blockIdx.x == gridDim.x - 1) {
    // the extra workgroup column (grid.x = rows + 1, see affine_bwd_apply_gn2_kernel): the parameter gradients
    if (blockIdx.y == 0 && t < 64) {
      const GnBwdTerm tm = {nullptr, 0, a, bb, sums, rows, gamma, mean_rstd, wptr, sumraw, nullptr, 0, dgamma, dbeta, dalpha, dbias_conv,
                            RELU ? 1 : 0, nullptr, nullptr, nullptr};
      if (B <= 2) gn_bwd_prologue_impl<1, 2, 4>(tm, tm, B, C, G, count, &coefw[0]);
      else gn_bwd_prologue_impl<1, GNF_MAXB, 2>(tm, tm, B, C, G, count, &coefw[0]);
    }
    return;
  }
  const bool leader = !wave_path && (blockIdx.x == 0 && blockIdx.y == 0);
  const int nb = leader ? B : 1;
  // ---- load phase (no dependent loads): the first iteration's tensor operands and forward coefficients, then
  // parameters, group statistics and this thread's share of the partial rows
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  const T* dbp = dout + (int64_t)blockIdx.y * N * dld + c4 * 4;
  const T* rb = raw + (int64_t)blockIdx.y * N * rld + c4 * 4;
  T* ob = draw + (int64_t)blockIdx.y * N * drld + c4 * 4;
  const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
  const bool act0 = vl < m.vpb && v0 < N;
  float4 dq0 = make_float4(0.f, 0.f, 0.f, 0.f), rq0 = dq0, pq0 = dq0, aq = make_float4(1.f, 1.f, 1.f, 1.f), bq = dq0;
  if (act0) {
    dq0 = ld4(dbp + v0 * dld);
    rq0 = ld4(rb + v0 * rld);
    if (ACC) pq0 = ld4(ob + v0 * drld);
    const int co = (int)blockIdx.y * C + c4 * 4;
    if (a) aq = ld4(a + co);
    if (bb) bq = ld4(bb + co);
  }
  float Av[4], Bv[4], Cv[4];
  if (is_pow2(C) && cg <= 16) {
    // wave-level prologue (no workgroup barrier): lane = row slot x channel; partial rows -> class sums over the
    // row slots -> DPP group sums -> GroupNorm-backward coefficients, redundantly in every wave.  Wave 0 of the
    // leader workgroup walks all B samples to emit dgamma / dbeta / dalpha / conv-bias gradient.
    const int wave = t >> 6;
    // (the pair kernels' prologue with one term: all operands of the wave in one go)
    const GnBwdTerm tm = {nullptr, 0, a, bb, sums, rows, gamma, mean_rstd, wptr, sumraw, nullptr, 0, dgamma, dbeta, dalpha, dbias_conv,
                          RELU ? 1 : 0, nullptr, nullptr, nullptr};
    gn_bwd_prologue_impl<1, 1, 4>(tm, tm, B, C, G, count, &coefw[wave]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!act0) return;
    const float4 qa = ld4(&coefw[wave][0][c4 * 4]);
    const float4 qb = ld4(&coefw[wave][1][c4 * 4]);
    const float4 qc = ld4(&coefw[wave][2][c4 * 4]);
    Av[0] = qa.x; Av[1] = qa.y; Av[2] = qa.z; Av[3] = qa.w;
    Bv[0] = qb.x; Bv[1] = qb.y; Bv[2] = qb.z; Bv[3] = qb.w;
    Cv[0] = qc.x; Cv[1] = qc.y; Cv[2] = qc.z; Cv[3] = qc.w;
  } else {
    const double w = wptr ? (double)*wptr : 1.0;
    const double gam = (t < C) ? (double)gamma[t] : 0.0;
    const int gq = (t < C) ? t / cg : 0;
    double mean_c[GNF_MAXB], rstd_c[GNF_MAXB], ps[GNF_MAXB], pf[GNF_MAXB];
    const int ncol3 = C * 3;
    const int nrl3 = 256 / ncol3 > 0 ? 256 / ncol3 : 1;
    const int q3 = t % ncol3, rl3 = t / ncol3;
    const bool want_f = leader && dbias_conv;
  #pragma unroll
    for (int k = 0; k < GNF_MAXB; ++k) {
      ps[k] = 0; pf[k] = 0; mean_c[k] = 0; rstd_c[k] = 1;
      if (k < nb) {
        const int b = leader ? ((int)blockIdx.y + 1 + k) % B : (int)blockIdx.y;
        mean_c[k] = mean_rstd[(b * G + gq) * 2]; rstd_c[k] = mean_rstd[(b * G + gq) * 2 + 1];
        if (want_f && t < C) pf[k] = sumraw[b * C + t];
      }
    }
    if (rl3 < nrl3)
      for (int r = rl3; r < rows; r += nrl3) {
  #pragma unroll
        for (int k = 0; k < GNF_MAXB; ++k)
          if (k < nb) {
            const int b = leader ? ((int)blockIdx.y + 1 + k) % B : (int)blockIdx.y;
            ps[k] += sums[((int64_t)b * rows + r) * ncol3 + q3];
          }
      }
    if (t < C) gsh[t] = gam;
    // ---- LDS phase, sample by sample (the leader's own sample comes last so that coef[] ends up its own)
    double dg = 0, db = 0, dz = 0, dbc = 0;
  #pragma unroll
    for (int k = 0; k < GNF_MAXB; ++k) {
      if (k < nb) {
        part[t] = ps[k];
        __syncthreads();
        if (t < ncol3) {
          double acc = 0;
          for (int r = 0; r < nrl3; ++r) acc += part[r * ncol3 + t];
          tot[t] = acc;
        }
        __syncthreads();
        if (t < C && (t % cg) == 0) {
          // one thread per group: mean_c / rstd_c of this thread ARE the group's
          double c1 = 0, c2 = 0;
          for (int c = t; c < t + cg; ++c) {
            const double S1 = tot[c * 3], S2 = tot[c * 3 + 1];
            c1 += gsh[c] * w * S1;
            c2 += gsh[c] * w * rstd_c[k] * (S2 - mean_c[k] * S1);
          }
          const double n = count * cg;
          gc[gq * 2] = c1 / n; gc[gq * 2 + 1] = c2 / n;
        }
        __syncthreads();
        if (t < C) {
          const double S1 = tot[t * 3], S2 = tot[t * 3 + 1], Sz = tot[t * 3 + 2];
          const double c1 = gc[gq * 2], c2 = gc[gq * 2 + 1];
          const double rs = rstd_c[k], mn = mean_c[k];
          const double Av = rs * gam * w, Bv = -rs * c1 + rs * rs * c2 * mn, Cv = -rs * rs * c2;
          coef[0][t] = (float)Av; coef[1][t] = (float)Bv; coef[2][t] = (float)Cv;
          if (leader) {
            dg += w * rs * (S2 - mn * S1);
            db += w * S1;
            dz += Sz;
            if (dbias_conv) dbc += Av * S1 + count * Bv + Cv * pf[k];
          }
        }
        __syncthreads();
      }
    }
    if (leader) {
      if (t < C) {
        if (dgamma) dgamma[t] = (float)dg;
        if (dbeta) dbeta[t] = (float)db;
        if (dbias_conv) dbias_conv[t] = (float)dbc;
      }
      if (dalpha) {
        if (t < 64) zred[t] = (t < C) ? dz : 0.0;
        __syncthreads();
        if (t == 0) {
          double sdz = 0;
          for (int i = 0; i < 64; ++i) sdz += zred[i];
          *dalpha = (float)sdz;
        }
      }
    }
    if (!act0) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) { Av[j] = coef[0][c4 * 4 + j]; Bv[j] = coef[1][c4 * 4 + j]; Cv[j] = coef[2][c4 * 4 + j]; }
  }
  const float av[4] = {aq.x, aq.y, aq.z, aq.w}, bv[4] = {bq.x, bq.y, bq.z, bq.w};
  // iterations 1.. are requested before iteration 0 is finished (loads first, then math and stores)
  constexpr int PF = 3;
  float4 dn[PF], rn[PF], pn[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int64_t v = v0 + (int64_t)(i + 1) * m.vpb;
    const bool ok = (i + 1) < m.iters && v < N;
    const int64_t vc = ok ? v : v0;
    dn[i] = ld4(dbp + vc * dld);
    rn[i] = ld4(rb + vc * rld);
    if (ACC) pn[i] = ld4(ob + vc * drld);
  }
  auto emit = [&](int64_t v, const float4 dq, const float4 rq, const float4 pq) {
    const float d[4] = {dq.x, dq.y, dq.z, dq.w};
    const float r[4] = {rq.x, rq.y, rq.z, rq.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g = d[j];
      if (RELU) { const float z = fmaf(av[j], r[j], bv[j]); g = z > 0.f ? g : 0.f; }
      o[j] = fmaf(Av[j], g, fmaf(Cv[j], r[j], Bv[j]));
    }
    if (ACC) { o[0] += pq.x; o[1] += pq.y; o[2] += pq.z; o[3] += pq.w; }
    st4(ob + v * drld, make_float4(o[0], o[1], o[2], o[3]));
  };
  emit(v0, dq0, rq0, pq0);
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int64_t v = v0 + (int64_t)(i + 1) * m.vpb;
    if ((i + 1) < m.iters && v < N) emit(v, dn[i], rn[i], pn[i]);
  }
  for (int it = PF + 1; it < m.iters; it += 4) {   // four iterations at a time, loads first
    float4 dq[4], rq[4], pq[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t v = v0 + (int64_t)(it + j) * m.vpb;
      const int64_t vc = ((it + j) < m.iters && v < N) ? v : v0;
      dq[j] = ld4(dbp + vc * dld);
      rq[j] = ld4(rb + vc * rld);
      pq[j] = ACC ? ld4(ob + vc * drld) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t v = v0 + (int64_t)(it + j) * m.vpb;
      if ((it + j) < m.iters && v < N) emit(v, dq[j], rq[j], pq[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Node-level pair kernels.  A searched-cell node is  op_a(x_i) + op_b(x_j)  (searched.py:45-50): both ops end in the
// same GroupNorm -> ReLU -> weighted-sum epilogue on tensors of one shape, and in backward both consume the same
// node gradient.  One launch handles both terms: forward out = term0 + term1 written once (no read-modify-write
// of the node buffer), backward reads the node gradient once for both reductions / both gradient tensors.
// Wave-level prologues only (C a power of two, group width <= 16); the host falls back to two single launches.
// ------------------------------------------------------------------------------------------------
struct GnFwdTerm {
  const float* raw; int64_t rld; const double* stats; int rows; const float* gamma; const float* beta; const float* wptr;
  float* a_out; float* b_out; float* mr_out; double* sumraw; int relu;
};

// forward GroupNorm coefficients of sample b, computed redundantly by one wave; the wave's lanes < C publish them
// in its private LDS strip (strip[0][c] = a, strip[1][c] = b)
__device__ __forceinline__ void gn_fwd_prologue_wave(const GnFwdTerm& t, const int b, const int C, const int G, const double count, const float eps,
                                                     const bool store, float (*strip)[64]) {
  const int lane = threadIdx.x & 63;
  const int cg = C / G;
  const int c = lane & (C - 1), rs = lane / C, nslots = 64 / C;
  const float gam = t.gamma[c], bet = t.beta[c];
  const double2* st2 = reinterpret_cast<const double2*>(t.stats + (int64_t)b * t.rows * C * 2);
  double s = 0, ss = 0;
  int r = rs;
  for (; r + 3 * nslots < t.rows; r += 4 * nslots) {
    double2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = st2[(int64_t)(r + u * nslots) * C + c];
#pragma unroll
    for (int u = 0; u < 4; ++u) { s += v[u].x; ss += v[u].y; }
  }
  for (; r < t.rows; r += nslots) { const double2 v = st2[(int64_t)r * C + c]; s += v.x; ss += v.y; }
  double gv[2] = {s, ss};
  wave_classsum_dn<2>(gv, C);
  s = gv[0];
  wave_groupsum_dn<2>(gv, cg);
  const double gs = gv[0], gss = gv[1];
  const double n = count * cg;
  const double mean = gs / n;
  double var = gss / n - mean * mean;
  if (var < 0) var = 0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const float a1 = gam * (float)rstd;
  const float b1 = bet - (float)mean * a1;
  if (lane < C) { strip[0][lane] = a1; strip[1][lane] = b1; }
  if (store && lane < C) {
    t.a_out[b * C + c] = a1; t.b_out[b * C + c] = b1;
    if (t.sumraw) t.sumraw[b * C + c] = s;
    if (c % cg == 0) { const int g = c / cg; t.mr_out[(b * G + g) * 2] = (float)mean; t.mr_out[(b * G + g) * 2 + 1] = (float)rstd; }
  }
}

// PRE: the GroupNorm coefficients were computed by n3d_gn_coeffs2 (large tensors): a_out / b_out are inputs, no prologue
template <bool ACC, bool PRE, typename T = float>
__global__ __launch_bounds__(256) void affine_act_gn2_kernel(GnFwdTerm t0, GnFwdTerm t1, int G, double count, float eps, T* __restrict__ out,
                                                             int64_t old_, T* __restrict__ out1, int64_t old1, int64_t N, int C, EwMap m) {
  N3D_CHAIN_PRIO();
  __shared__ __attribute__((aligned(16))) float abw[4][2][2][64];  // [wave][term][a|b][channel]
  const int b = blockIdx.y, t = threadIdx.x, wave = t >> 6;
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  const T* r0 = reinterpret_cast<const T*>(t0.raw) + (int64_t)b * N * t0.rld + c4 * 4;
  const T* r1 = reinterpret_cast<const T*>(t1.raw) + (int64_t)b * N * t1.rld + c4 * 4;
  T* ob = out + (int64_t)b * N * old_ + c4 * 4;
  const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
  const bool act0 = vl < m.vpb && v0 < N;
  constexpr int PF = 4;   // iterations requested up front (the first one ahead of the prologues)
  float4 q0[PF], q1[PF], on[PF];
  const float w0 = t0.wptr ? *t0.wptr : 1.0f, w1 = t1.wptr ? *t1.wptr : 1.0f;
  if (act0) {
    q0[0] = ld4(r0 + v0 * t0.rld);
    q1[0] = ld4(r1 + v0 * t1.rld);
    if (ACC) on[0] = ld4(ob + v0 * old_);
  }
  float4 a0, b0, a1, b1;
  if (PRE) {
    if (!act0) return;
    const int co = b * C + c4 * 4;
    a0 = ld4(t0.a_out + co); b0 = ld4(t0.b_out + co);
    a1 = ld4(t1.a_out + co); b1 = ld4(t1.b_out + co);
  } else {
    const bool store = blockIdx.x == 0 && wave == 0;
    gn_fwd_prologue_wave(t0, b, C, G, count, eps, store, abw[wave][0]);
    gn_fwd_prologue_wave(t1, b, C, G, count, eps, store, abw[wave][1]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!act0) return;
    a0 = ld4(&abw[wave][0][0][c4 * 4]); b0 = ld4(&abw[wave][0][1][c4 * 4]);
    a1 = ld4(&abw[wave][1][0][c4 * 4]); b1 = ld4(&abw[wave][1][1][c4 * 4]);
  }
  const float f0 = t0.relu ? 0.f : -INFINITY, f1 = t1.relu ? 0.f : -INFINITY;
#pragma unroll
  for (int i = 1; i < PF; ++i) {
    const int64_t v = v0 + (int64_t)i * m.vpb;
    const int64_t vc = (i < m.iters && v < N) ? v : v0;
    q0[i] = ld4(r0 + vc * t0.rld);
    q1[i] = ld4(r1 + vc * t1.rld);
    if (ACC) on[i] = ld4(ob + vc * old_);
  }
  auto emit = [&](int64_t v, const float4 x0, const float4 x1, const float4 o) {
    float4 z;
    // term0 first, then term1, then (accumulate mode) the previous content: the reference's left-to-right sum
    z.x = w0 * fmaxf(fmaf(a0.x, x0.x, b0.x), f0); z.y = w0 * fmaxf(fmaf(a0.y, x0.y, b0.y), f0);
    z.z = w0 * fmaxf(fmaf(a0.z, x0.z, b0.z), f0); z.w = w0 * fmaxf(fmaf(a0.w, x0.w, b0.w), f0);
    if (ACC) { z.x += o.x; z.y += o.y; z.z += o.z; z.w += o.w; }
    if (out1) {
      // two independent outputs (the two preprocess ops of a cell): term 1 goes to its own tensor
      st4(ob + v * old_, z);
      float4 y;
      y.x = w1 * fmaxf(fmaf(a1.x, x1.x, b1.x), f1); y.y = w1 * fmaxf(fmaf(a1.y, x1.y, b1.y), f1);
      y.z = w1 * fmaxf(fmaf(a1.z, x1.z, b1.z), f1); y.w = w1 * fmaxf(fmaf(a1.w, x1.w, b1.w), f1);
      st4(out1 + (int64_t)b * N * old1 + c4 * 4 + v * old1, y);
      return;
    }
    z.x = fmaf(w1, fmaxf(fmaf(a1.x, x1.x, b1.x), f1), z.x); z.y = fmaf(w1, fmaxf(fmaf(a1.y, x1.y, b1.y), f1), z.y);
    z.z = fmaf(w1, fmaxf(fmaf(a1.z, x1.z, b1.z), f1), z.z); z.w = fmaf(w1, fmaxf(fmaf(a1.w, x1.w, b1.w), f1), z.w);
    st4(ob + v * old_, z);
  };
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int64_t v = v0 + (int64_t)i * m.vpb;
    if (i < m.iters && v < N) emit(v, q0[i], q1[i], on[i]);
  }
  // further iterations (tensors beyond 1024 rows x 4 iterations per sample: the 128^3 level) go four at a time, loads first
  // (one at a time, load -> math -> store, this pass ran at 2.2 TB/s at (2,4,128^3) against 3.5-3.9 for the backward passes)
  for (int it = PF; it < m.iters; it += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int64_t v = v0 + (int64_t)(it + j) * m.vpb;
      const int64_t vc = ((it + j) < m.iters && v < N) ? v : v0;
      q0[j] = ld4(r0 + vc * t0.rld);
      q1[j] = ld4(r1 + vc * t1.rld);
      if (ACC) on[j] = ld4(ob + vc * old_);
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int64_t v = v0 + (int64_t)(it + j) * m.vpb;
      if ((it + j) < m.iters && v < N) emit(v, q0[j], q1[j], on[j]);
    }
  }
}

// backward pass 1 for two ops that share the node gradient dout: sums0 / sums1 rows as affine_bwd_reduce_kernel
struct BwdRedTerm { const float* raw; int64_t rld; const float* a; const float* b; double* sums; int relu; };

// TWO: the second op has its own output gradient (independent outputs); otherwise both share dout (a node)
template <bool TWO, typename T = float>
__global__ __launch_bounds__(256) void affine_bwd_reduce2_kernel(const T* __restrict__ dout, int64_t dld, const T* __restrict__ dout1,
                                                                 int64_t dld1, BwdRedTerm t0, BwdRedTerm t1, int64_t N, int C, EwMap m) {
  N3D_CHAIN_PRIO();
  __shared__ double lds[4 * 64 * 12];
  const int b = blockIdx.y;
  const int t = threadIdx.x;
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  const bool active = vl < m.vpb;
  float s1[2][4], s2[2][4], sz[2][4];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) s1[k][j] = s2[k][j] = sz[k][j] = 0.f;
  if (active) {
    float4 av[2], bv[2];
    av[0] = ld4(t0.a + b * C + c4 * 4); bv[0] = ld4(t0.b + b * C + c4 * 4);
    av[1] = ld4(t1.a + b * C + c4 * 4); bv[1] = ld4(t1.b + b * C + c4 * 4);
    const float thr[2] = {t0.relu ? 0.f : -INFINITY, t1.relu ? 0.f : -INFINITY};
    const T* db = dout + (int64_t)b * N * dld + c4 * 4;
    const T* db1 = TWO ? dout1 + (int64_t)b * N * dld1 + c4 * 4 : nullptr;
    const T* rb0 = reinterpret_cast<const T*>(t0.raw) + (int64_t)b * N * t0.rld + c4 * 4;
    const T* rb1 = reinterpret_cast<const T*>(t1.raw) + (int64_t)b * N * t1.rld + c4 * 4;
    const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
    for (int it0 = 0; it0 < m.iters; it0 += 4) {
      float4 dq[4], dq1[TWO ? 4 : 1], rq[2][4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t v = v0 + (int64_t)(it0 + u) * m.vpb;
        ok[u] = (it0 + u < m.iters) && v < N;
        const int64_t vc = ok[u] ? v : 0;
        dq[u] = ld4(db + vc * dld);
        if (TWO) dq1[TWO ? u : 0] = ld4(db1 + vc * dld1);
        rq[0][u] = ld4(rb0 + vc * t0.rld);
        rq[1][u] = ld4(rb1 + vc * t1.rld);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float4 dk = (TWO && k) ? dq1[TWO ? u : 0] : dq[u];
          const float d[4] = {dk.x, dk.y, dk.z, dk.w};
          const float r[4] = {rq[k][u].x, rq[k][u].y, rq[k][u].z, rq[k][u].w};
          const float a4[4] = {av[k].x, av[k].y, av[k].z, av[k].w}, b4[4] = {bv[k].x, bv[k].y, bv[k].z, bv[k].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float z = fmaf(a4[j], r[j], b4[j]);
            const float g = z > thr[k] ? d[j] : 0.f;
            z = fmaxf(z, thr[k]);
            s1[k][j] += g;
            s2[k][j] = fmaf(g, r[j], s2[k][j]);
            sz[k][j] = fmaf(d[j], z, sz[k][j]);
          }
        }
      }
    }
  }
  if (m.cpb <= 16) {
    const int64_t ro = ((int64_t)b * gridDim.x + blockIdx.x) * C * 3;
    double* const rows[2] = {t0.sums + ro, t1.sums + ro};
    class_dispatch16(m.cpb, [&](auto cc) {
      constexpr int CPB = decltype(cc)::value;
      float X[6];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        X[3 * k + 0] = wave_classsum4_f<CPB>(s1[k][0], s1[k][1], s1[k][2], s1[k][3]);
        X[3 * k + 1] = wave_classsum4_f<CPB>(s2[k][0], s2[k][1], s2[k][2], s2[k][3]);
        X[3 * k + 2] = wave_classsum4_f<CPB>(sz[k][0], sz[k][1], sz[k][2], sz[k][3]);
      }
      block_reduce_packed_rows<2, 3, CPB>(X, rows, reinterpret_cast<float*>(lds));
    });
    return;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    double vals[12];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vals[j] = wave_classsum_f(s1[k][j], m.cpb); vals[4 + j] = wave_classsum_f(s2[k][j], m.cpb); vals[8 + j] = wave_classsum_f(sz[k][j], m.cpb);
    }
    double* row = (k == 0 ? t0.sums : t1.sums) + ((int64_t)b * gridDim.x + blockIdx.x) * C * 3;
    if (k == 1) __syncthreads();
    block_reduce_to_row<3>(vals, m.cpb, row, lds, true);
  }
}



template <bool PRE, bool TWO, typename T = float>
__global__ __launch_bounds__(256) void affine_bwd_apply_gn2_kernel(const T* __restrict__ dout, int64_t dld, const T* __restrict__ dout1,
                                                                   int64_t dld1, GnBwdTerm t0, GnBwdTerm t1, int B, int G, double count, int64_t N,
                                                                   int C, EwMap m) {
  N3D_CHAIN_PRIO();
  __shared__ __attribute__((aligned(16))) float cw[4][2][3][64];  // [wave][term][A|B|C][channel]
  const int t = threadIdx.x, wave = t >> 6;
  if (!PRE && blockIdx.x == gridDim.x - 1) {
    // the extra workgroup of the launch (grid.x = rows + 1): the parameter gradients, which need the partial rows of EVERY sample.
    // They used to be the job of wave 0 of workgroup (0, 0) in front of its share of the apply work, and a launch lasts as long
    // as its slowest workgroup; here they run beside the apply workgroups, whose waves all handle one sample
    if (blockIdx.y == 0 && wave == 0) gn_bwd_prologue_wave2(t0, t1, B, C, G, count, true, cw[0]);
    return;
  }
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  const int by = blockIdx.y;
  const T* dbp = dout + (int64_t)by * N * dld + c4 * 4;
  const T* dbp1 = TWO ? dout1 + (int64_t)by * N * dld1 + c4 * 4 : nullptr;  // TWO: second op has its own output gradient
  const T* rb0 = reinterpret_cast<const T*>(t0.raw) + (int64_t)by * N * t0.rld + c4 * 4;
  const T* rb1 = reinterpret_cast<const T*>(t1.raw) + (int64_t)by * N * t1.rld + c4 * 4;
  T* o0 = reinterpret_cast<T*>(t0.draw) + (int64_t)by * N * t0.drld + c4 * 4;
  T* o1 = reinterpret_cast<T*>(t1.draw) + (int64_t)by * N * t1.drld + c4 * 4;
  const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
  const bool act0 = vl < m.vpb && v0 < N;
  constexpr int PF = 4;
  float4 dq[PF], dq1[TWO ? PF : 1], r0[PF], r1[PF];
  float4 fa[2], fb[2];
  fa[0] = fa[1] = make_float4(1.f, 1.f, 1.f, 1.f); fb[0] = fb[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (act0) {
    dq[0] = ld4(dbp + v0 * dld);
    if (TWO) dq1[0] = ld4(dbp1 + v0 * dld1);
    r0[0] = ld4(rb0 + v0 * t0.rld);
    r1[0] = ld4(rb1 + v0 * t1.rld);
    const int co = by * C + c4 * 4;
    fa[0] = ld4(t0.a + co); fb[0] = ld4(t0.b + co);
    fa[1] = ld4(t1.a + co); fb[1] = ld4(t1.b + co);
  }
  float4 cA[2], cB[2], cC[2];
  if (PRE) {
    if (!act0) return;
    const int co = by * C + c4 * 4;
    cA[0] = ld4(t0.cA + co); cB[0] = ld4(t0.cB + co); cC[0] = ld4(t0.cC + co);
    cA[1] = ld4(t1.cA + co); cB[1] = ld4(t1.cB + co); cC[1] = ld4(t1.cC + co);
  } else {
    gn_bwd_prologue_wave2(t0, t1, B, C, G, count, false, cw[wave]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!act0) return;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      cA[k] = ld4(&cw[wave][k][0][c4 * 4]);
      cB[k] = ld4(&cw[wave][k][1][c4 * 4]);
      cC[k] = ld4(&cw[wave][k][2][c4 * 4]);
    }
  }
  const float thr[2] = {t0.relu ? 0.f : -INFINITY, t1.relu ? 0.f : -INFINITY};
#pragma unroll
  for (int i = 1; i < PF; ++i) {
    const int64_t v = v0 + (int64_t)i * m.vpb;
    const int64_t vc = (i < m.iters && v < N) ? v : v0;
    dq[i] = ld4(dbp + vc * dld);
    if (TWO) dq1[TWO ? i : 0] = ld4(dbp1 + vc * dld1);
    r0[i] = ld4(rb0 + vc * t0.rld);
    r1[i] = ld4(rb1 + vc * t1.rld);
  }
  auto one = [&](const int k, const float4 d4, const float4 r4, T* op) {
    const float d[4] = {d4.x, d4.y, d4.z, d4.w}, r[4] = {r4.x, r4.y, r4.z, r4.w};
    const float a4[4] = {fa[k].x, fa[k].y, fa[k].z, fa[k].w}, b4[4] = {fb[k].x, fb[k].y, fb[k].z, fb[k].w};
    const float A4[4] = {cA[k].x, cA[k].y, cA[k].z, cA[k].w}, B4[4] = {cB[k].x, cB[k].y, cB[k].z, cB[k].w}, C4[4] = {cC[k].x, cC[k].y, cC[k].z, cC[k].w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float z = fmaf(a4[j], r[j], b4[j]);
      const float g = z > thr[k] ? d[j] : 0.f;
      o[j] = fmaf(A4[j], g, fmaf(C4[j], r[j], B4[j]));
    }
    st4(op, make_float4(o[0], o[1], o[2], o[3]));
  };
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int64_t v = v0 + (int64_t)i * m.vpb;
    if (i < m.iters && v < N) { one(0, dq[i], r0[i], o0 + v * t0.drld); one(1, TWO ? dq1[TWO ? i : 0] : dq[i], r1[i], o1 + v * t1.drld); }
  }
  for (int it = PF; it < m.iters; it += PF) {   // four iterations at a time, loads first
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int64_t v = v0 + (int64_t)(it + j) * m.vpb;
      const int64_t vc = ((it + j) < m.iters && v < N) ? v : v0;
      dq[j] = ld4(dbp + vc * dld);
      if (TWO) dq1[TWO ? j : 0] = ld4(dbp1 + vc * dld1);
      r0[j] = ld4(rb0 + vc * t0.rld);
      r1[j] = ld4(rb1 + vc * t1.rld);
    }
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int64_t v = v0 + (int64_t)(it + j) * m.vpb;
      if ((it + j) < m.iters && v < N) { one(0, dq[j], r0[j], o0 + v * t0.drld); one(1, TWO ? dq1[TWO ? j : 0] : dq[j], r1[j], o1 + v * t1.drld); }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Whole GroupNorm-epilogue backward of a searched-cell node on the small levels (<= 2048 channel quads per group) in ONE
// launch: reduction pass, coefficient math and d(raw) pass of BOTH terms, one workgroup per GroupNorm group, every element
// held in registers between the passes (the reduce2 + apply_gn2 pair re-reads all three tensors and pays two launches).
// Thread t owns the quads e = t + i*1024 (i < QPT) of the group's [B][N][cg/4] quads; samples start on wave boundaries
// (padded with idle lanes where N * cg/4 is not a multiple of 64), so a wave-level class sum never mixes samples.  Sums: fp32 per lane and per wave (DPP tree),
// fp64 across waves in a fixed order, coefficients in fp64 as the two-launch path does.
// ------------------------------------------------------------------------------------------------
// TWO: term 1 has its own output gradient dout1 (the two preprocess ops of a cell, independent outputs of one shape)
// SPLIT: one workgroup per (group, SAMPLE) -- blockIdx.y = sample -- for tensors whose B samples do not fit one workgroup (the 8^3
//   level at 16 channels per group: 2048 quads per sample).  Everything up to d(raw) is per sample; the parameter gradients sum
//   over the samples, so each workgroup stores its contribution to `scratch` and the one that draws the last ticket of the group
//   (atomicInc wrapping at B - 1: the counter is back at 0 for the next launch / graph replay) adds them in sample order.
// NT: 2 = a node's two terms; 1 = a single epilogue (a preprocess op whose partner has another shape)
template <int QPT, bool TWO, bool SPLIT, int NT, typename T = float>
__global__ __launch_bounds__(1024) void gn_bwd_small2_kernel(const T* __restrict__ dout, int64_t dld, const T* __restrict__ dout1,
                                                            int64_t dld1, GnBwdTerm t0, GnBwdTerm t1, int B, int N, int C, int G, double count,
                                                            double* __restrict__ scratch, unsigned* __restrict__ tickets) {
  N3D_CHAIN_PRIO();
  constexpr int MB = SPLIT ? 1 : 4;
  __shared__ float red[QPT * 16][2][32];   // [slot = i*16 + wave][term][(S1 | S2) x 16 channels]
  __shared__ double tot[MB][2][32];        // [b][term][(S1 | S2) x 16 channels]
  __shared__ float fco[MB][2][2][16];      // forward a | b of (sample, term, channel): staged once, read twice per element
  __shared__ float coef[MB][2][3][16];     // [b][term][A | B | C][channel]
  __shared__ double contrib[MB][2][3][16]; // [b][term][dgamma | dbeta | dbias][channel]
  __shared__ unsigned last_one;
  const int g = blockIdx.x, cg = C / G, cpg4 = cg >> 2;
  const int b0 = SPLIT ? (int)blockIdx.y : 0, Bl = SPLIT ? 1 : B;   // first sample / number of samples of this workgroup
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  // a sample's quads occupy a whole number of waves (padded with idle lanes when N * cg/4 is not a multiple of 64: the 2^3 level)
  const int real_b = N * cpg4, per_b = (real_b + 63) & ~63, total = Bl * per_b;
  const int q = t % cpg4;                  // the channel quad of every element of this thread
  const int c0 = g * cg + q * 4;
  const float thr0 = t0.relu ? 0.f : -INFINITY, thr1 = t1.relu ? 0.f : -INFINITY;
  if (t < Bl * 2 * 2 * 16) {
    const int b = t >> 6, k = (t >> 5) & 1, ab = (t >> 4) & 1, c = t & 15;
    const GnBwdTerm& tm = k ? t1 : t0;
    fco[b][k][ab][c] = (c < cg && k < NT) ? (ab ? tm.b : tm.a)[(b0 + b) * C + g * cg + c] : 0.f;
  }
  float4 d4[QPT], e4[TWO ? QPT : 1], r0[QPT], r1[NT == 2 ? QPT : 1];
  bool ok[QPT];
  int bi[QPT];
  int64_t vox[QPT];
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int e = t + i * 1024;
    const int eb = e / per_b, er = e - eb * per_b;
    ok[i] = e < total && er < real_b;
    bi[i] = ok[i] ? eb : 0;
    vox[i] = (int64_t)(b0 + bi[i]) * N + (ok[i] ? er : 0) / cpg4;
    d4[i] = ld4(dout + vox[i] * dld + c0);
    if (TWO) e4[TWO ? i : 0] = ld4(dout1 + vox[i] * dld1 + c0);
    r0[i] = ld4(reinterpret_cast<const T*>(t0.raw) + vox[i] * t0.rld + c0);
    if (NT == 2) r1[NT == 2 ? i : 0] = ld4(reinterpret_cast<const T*>(t1.raw) + vox[i] * t1.rld + c0);
  }
  __syncthreads();
  // ---- pass 1: S1 = sum g, S2 = sum g * raw per (sample, channel), g = dout behind the term's ReLU mask
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const float4 e4i = TWO ? e4[TWO ? i : 0] : d4[i];
    const float4 r1i = NT == 2 ? r1[NT == 2 ? i : 0] : r0[i];
    const float dd[2][4] = {{d4[i].x, d4[i].y, d4[i].z, d4[i].w}, {e4i.x, e4i.y, e4i.z, e4i.w}};
    const float ra[2][4] = {{r0[i].x, r0[i].y, r0[i].z, r0[i].w}, {r1i.x, r1i.y, r1i.z, r1i.w}};
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const float thr = k ? thr1 : thr0;
      const float4 fa = ld4(&fco[bi[i]][k][0][q * 4]), fb = ld4(&fco[bi[i]][k][1][q * 4]);
      const float aa[4] = {fa.x, fa.y, fa.z, fa.w}, bb[4] = {fb.x, fb.y, fb.z, fb.w};
      float gmv[4], grv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z = fmaf(aa[j], ra[k][j], bb[j]);
        gmv[j] = (ok[i] && z > thr) ? dd[k][j] : 0.f;
        grv[j] = gmv[j] * ra[k][j];
      }
      // the four channels of the quad reduced together (wave_classsum4_f): row r of the wave ends up with channel {0,2,1,3}[r]
      class_dispatch16(cpg4, [&](auto cc) {
        constexpr int CPB = decltype(cc)::value;
        const float s1 = wave_classsum4_f<CPB>(gmv[0], gmv[1], gmv[2], gmv[3]), s2 = wave_classsum4_f<CPB>(grv[0], grv[1], grv[2], grv[3]);
        const int l16 = lane & 15, j = classsum4_sel(lane);
        if (l16 < CPB) { red[i * 16 + wave][k][l16 * 4 + j] = s1; red[i * 16 + wave][k][16 + l16 * 4 + j] = s2; }
      });
    }
  }
  __syncthreads();
  // fixed-order sum of the wave slots of each sample: slot s covers the quads [64 s, 64 s + 64), a sample per_b / 64 of them
  if (t < Bl * 2 * 32) {
    const int b = t >> 6, k = (t >> 5) & 1, idx = t & 31;
    double s = 0;
    if ((idx & 15) < cg && k < NT) {
      const int spb = per_b >> 6, s0 = b * spb;
      for (int sl = s0; sl < s0 + spb; ++sl) s += (double)red[sl][k][idx];
    }
    tot[b][k][idx] = s;
  }
  __syncthreads();
  // ---- coefficients of d(raw) = A * g + B + C * raw and the per-sample parameter-gradient contributions
  if (t < Bl * 2 * 16) {
    const int b = t >> 5, k = (t >> 4) & 1, c = t & 15;
    if (c < cg && k < NT) {
      const GnBwdTerm& tm = k ? t1 : t0;
      const int gb = b0 + b;
      const double w = tm.wptr ? (double)*tm.wptr : 1.0;
      const double mn = tm.mean_rstd[(gb * G + g) * 2], rsd = tm.mean_rstd[(gb * G + g) * 2 + 1];
      double c1 = 0, c2 = 0;
      for (int cc = 0; cc < cg; ++cc) {
        const double gm = (double)tm.gamma[g * cg + cc];
        const double S1 = tot[b][k][cc], S2 = tot[b][k][16 + cc];
        c1 += gm * w * S1;
        c2 += gm * w * rsd * (S2 - mn * S1);
      }
      const double n = count * cg;
      c1 /= n; c2 /= n;
      const double gam = (double)tm.gamma[g * cg + c];
      const double S1 = tot[b][k][c], S2 = tot[b][k][16 + c];
      const double Av = rsd * gam * w, Bv = -rsd * c1 + rsd * rsd * c2 * mn, Cv = -rsd * rsd * c2;
      coef[b][k][0][c] = (float)Av; coef[b][k][1][c] = (float)Bv; coef[b][k][2][c] = (float)Cv;
      const double cdg = w * rsd * (S2 - mn * S1), cdb = w * S1;
      const double cdc = tm.dbias_conv ? Av * S1 + count * Bv + Cv * tm.sumraw[gb * C + g * cg + c] : 0.0;
      contrib[b][k][0][c] = cdg; contrib[b][k][1][c] = cdb; contrib[b][k][2][c] = cdc;
      if (SPLIT) {
        double* sc = scratch + ((((int64_t)g * B + gb) * 2 + k) * 3) * 16 + c;
        __hip_atomic_store(sc, cdg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sc + 16, cdb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sc + 32, cdc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
      }
    }
  }
  __syncthreads();
  if (SPLIT && t == 0) {
    // the ticket goes out before the d(raw) pass: by the time that pass is through, the last sample's workgroup has its answer
    last_one = atomicInc(&tickets[g], (unsigned)(B - 1)) == (unsigned)(B - 1);
  }
  // ---- pass 2: d(raw) of both terms from the registers
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    if (!ok[i]) continue;
    const float4 e4i = TWO ? e4[TWO ? i : 0] : d4[i];
    const float4 r1i = NT == 2 ? r1[NT == 2 ? i : 0] : r0[i];
    const float dd[2][4] = {{d4[i].x, d4[i].y, d4[i].z, d4[i].w}, {e4i.x, e4i.y, e4i.z, e4i.w}};
    const float ra[2][4] = {{r0[i].x, r0[i].y, r0[i].z, r0[i].w}, {r1i.x, r1i.y, r1i.z, r1i.w}};
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      const float thr = k ? thr1 : thr0;
      const float4 fa = ld4(&fco[bi[i]][k][0][q * 4]), fb = ld4(&fco[bi[i]][k][1][q * 4]);
      const float4 qA = ld4(&coef[bi[i]][k][0][q * 4]), qB = ld4(&coef[bi[i]][k][1][q * 4]),
                   qC = ld4(&coef[bi[i]][k][2][q * 4]);
      const float aa[4] = {fa.x, fa.y, fa.z, fa.w}, bb[4] = {fb.x, fb.y, fb.z, fb.w};
      const float A4[4] = {qA.x, qA.y, qA.z, qA.w}, B4[4] = {qB.x, qB.y, qB.z, qB.w}, C4[4] = {qC.x, qC.y, qC.z, qC.w};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float z = fmaf(aa[j], ra[k][j], bb[j]);
        const float gm = z > thr ? dd[k][j] : 0.f;
        o[j] = fmaf(A4[j], gm, fmaf(C4[j], ra[k][j], B4[j]));
      }
      T* op = (k ? reinterpret_cast<T*>(t1.draw) + vox[i] * t1.drld : reinterpret_cast<T*>(t0.draw) + vox[i] * t0.drld) + c0;
      st4(op, make_float4(o[0], o[1], o[2], o[3]));
    }
  }
  // ---- parameter gradients: sums over the samples in sample order
  if (SPLIT) {
    __syncthreads();
    if (!last_one) return;
    __threadfence();
  }
  if (t < 2 * 16) {
    const int k = t >> 4, c = t & 15;
    if (c < cg && k < NT) {
      const GnBwdTerm& tm = k ? t1 : t0;
      double dg = 0, db = 0, dbc = 0;
      for (int b = 0; b < B; ++b) {
        if (SPLIT) {
          const double* sc = scratch + ((((int64_t)g * B + b) * 2 + k) * 3) * 16 + c;
          dg += __hip_atomic_load(sc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          db += __hip_atomic_load(sc + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          dbc += __hip_atomic_load(sc + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          dg += contrib[b][k][0][c]; db += contrib[b][k][1][c]; dbc += contrib[b][k][2][c];
        }
      }
      if (tm.dgamma) tm.dgamma[g * cg + c] = (float)dg;
      if (tm.dbeta) tm.dbeta[g * cg + c] = (float)db;
      if (tm.dbias_conv) tm.dbias_conv[g * cg + c] = (float)dbc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// N-term epilogues of a supernet node (cell.py:76-81: a node sums 10-22 weighted primitives; 8-16 of them end in a GroupNorm).
// Coefficients come from gn_coeffsN / gn_bwd_coeffsN.  Forward: ONE pass writes (or accumulates into) the node buffer for up
// to 8 terms, in term order.  Backward: the per-term reductions and d(raw) passes are independent, so the launch is simply
// batched over blockIdx.z.
// ------------------------------------------------------------------------------------------------
struct FwdTermN { const float* raw[8]; int64_t rld[8]; const float* a[8]; const float* b[8]; const float* wptr[8]; int relu[8]; int n; };

template <bool ACC>
__global__ __launch_bounds__(256) void affine_actN_kernel(FwdTermN ts, float* __restrict__ out, int64_t old_, int64_t N, int C, EwMap m) { N3D_CHAIN_PRIO();
  const int b = blockIdx.y, t = threadIdx.x;
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  if (vl >= m.vpb) return;
  const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
  if (v0 >= N) return;
  const int co = b * C + c4 * 4;
  float4 av[8], bv[8];
  float w[8], fl[8];
  const float* rb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (k < ts.n) {
      av[k] = ts.a[k] ? ld4(ts.a[k] + co) : make_float4(1.f, 1.f, 1.f, 1.f);
      bv[k] = ts.b[k] ? ld4(ts.b[k] + co) : make_float4(0.f, 0.f, 0.f, 0.f);
      w[k] = ts.wptr[k] ? *ts.wptr[k] : 1.0f;
      fl[k] = ts.relu[k] ? 0.f : -INFINITY;
      rb[k] = ts.raw[k] + (int64_t)b * N * ts.rld[k] + c4 * 4;
    }
  }
  float* ob = out + (int64_t)b * N * old_ + c4 * 4;
  for (int it = 0; it < m.iters; ++it) {
    const int64_t v = v0 + (int64_t)it * m.vpb;
    if (v >= N) break;
    float4 q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < ts.n) q[k] = ld4(rb[k] + v * ts.rld[k]);
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ACC) z = ld4(ob + v * old_);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < ts.n) {
        z.x = fmaf(w[k], fmaxf(fmaf(av[k].x, q[k].x, bv[k].x), fl[k]), z.x); z.y = fmaf(w[k], fmaxf(fmaf(av[k].y, q[k].y, bv[k].y), fl[k]), z.y);
        z.z = fmaf(w[k], fmaxf(fmaf(av[k].z, q[k].z, bv[k].z), fl[k]), z.z); z.w = fmaf(w[k], fmaxf(fmaf(av[k].w, q[k].w, bv[k].w), fl[k]), z.w);
      }
    }
    st4(ob + v * old_, z);
  }
}

struct BwdRedTermN { BwdRedTerm t[N3D_MAX_REDUCE_TERMS]; };   // 16: every term of a node level in ONE reduction launch
// grid (rows, B, terms): the reduction pass of affine_bwd_reduce2_kernel for term blockIdx.z
__global__ __launch_bounds__(256) void affine_bwd_reduceN_kernel(const float* __restrict__ dout, int64_t dld, BwdRedTermN ts, int64_t N, int C,
                                                                 EwMap m) { N3D_CHAIN_PRIO();
  __shared__ double lds[4 * 64 * 12];
  BwdRedTerm tm;
  N3D_PICK16(ts.t, blockIdx.z, tm);
  const int b = blockIdx.y;
  const int t = threadIdx.x;
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  const bool active = vl < m.vpb;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, sz[4] = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    const float4 av = tm.a ? ld4(tm.a + b * C + c4 * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bv = tm.b ? ld4(tm.b + b * C + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
    const float thr = tm.relu ? 0.f : -INFINITY;
    const float* db = dout + (int64_t)b * N * dld + c4 * 4;
    const float* rb = tm.raw + (int64_t)b * N * tm.rld + c4 * 4;
    const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
    for (int it0 = 0; it0 < m.iters; it0 += 4) {
      float4 dq[4], rq[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t v = v0 + (int64_t)(it0 + u) * m.vpb;
        ok[u] = (it0 + u < m.iters) && v < N;
        const int64_t vc = ok[u] ? v : 0;
        dq[u] = ld4(db + vc * dld);
        rq[u] = ld4(rb + vc * tm.rld);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) continue;
        const float d[4] = {dq[u].x, dq[u].y, dq[u].z, dq[u].w}, r[4] = {rq[u].x, rq[u].y, rq[u].z, rq[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float z = fmaf(a4[j], r[j], b4[j]);
          const float g = z > thr ? d[j] : 0.f;
          z = fmaxf(z, thr);
          s1[j] += g;
          s2[j] = fmaf(g, r[j], s2[j]);
          sz[j] = fmaf(d[j], z, sz[j]);
        }
      }
    }
  }
  double* row = tm.sums + ((int64_t)b * gridDim.x + blockIdx.x) * C * 3;
  if (m.cpb <= 16) {
    class_dispatch16(m.cpb, [&](auto cc) {
      constexpr int CPB = decltype(cc)::value;
      const float X[3] = {wave_classsum4_f<CPB>(s1[0], s1[1], s1[2], s1[3]), wave_classsum4_f<CPB>(s2[0], s2[1], s2[2], s2[3]),
                          wave_classsum4_f<CPB>(sz[0], sz[1], sz[2], sz[3])};
      double* const rows[1] = {row};
      block_reduce_packed_rows<1, 3, CPB>(X, rows, reinterpret_cast<float*>(lds));
    });
    return;
  }
  double vals[12];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    vals[j] = wave_classsum_f(s1[j], m.cpb); vals[4 + j] = wave_classsum_f(s2[j], m.cpb); vals[8 + j] = wave_classsum_f(sz[j], m.cpb);
  }
  block_reduce_to_row<3>(vals, m.cpb, row, lds, true);
}

struct BwdApplyTerm { const float* raw; int64_t rld; const float* a; const float* b; const float* cA; const float* cB; const float* cC;
                      float* draw; int64_t drld; int relu; };
struct BwdApplyTermN { BwdApplyTerm t[N3D_MAX_REDUCE_TERMS]; };
// grid (rows, B, terms): draw = cA * g + cB + cC * raw of term blockIdx.z (g = dout behind the term's ReLU mask)
__global__ __launch_bounds__(256) void affine_bwd_applyN_kernel(const float* __restrict__ dout, int64_t dld, BwdApplyTermN ts, int64_t N, int C,
                                                                EwMap m) { N3D_CHAIN_PRIO();
  BwdApplyTerm tm;
  N3D_PICK16(ts.t, blockIdx.z, tm);
  const int b = blockIdx.y, t = threadIdx.x;
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  if (vl >= m.vpb) return;
  const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
  if (v0 >= N) return;
  const int co = b * C + c4 * 4;
  const float* db = dout + (int64_t)b * N * dld + c4 * 4;
  const float* rb = tm.raw + (int64_t)b * N * tm.rld + c4 * 4;
  float* ob = tm.draw + (int64_t)b * N * tm.drld + c4 * 4;
  // the first voxel's operands are requested together with the coefficients
  float4 d0 = ld4(db + v0 * dld), r0 = ld4(rb + v0 * tm.rld);
  const float4 fa = ld4(tm.a + co), fb = ld4(tm.b + co);
  const float4 qA = ld4(tm.cA + co), qB = ld4(tm.cB + co), qC = ld4(tm.cC + co);
  const float a4[4] = {fa.x, fa.y, fa.z, fa.w}, b4[4] = {fb.x, fb.y, fb.z, fb.w};
  const float A4[4] = {qA.x, qA.y, qA.z, qA.w}, B4[4] = {qB.x, qB.y, qB.z, qB.w}, C4[4] = {qC.x, qC.y, qC.z, qC.w};
  const float thr = tm.relu ? 0.f : -INFINITY;
  for (int it = 0; it < m.iters; ++it) {
    const int64_t v = v0 + (int64_t)it * m.vpb;
    if (v >= N) break;
    const int64_t vn = v + m.vpb;
    const bool more = it + 1 < m.iters && vn < N;
    float4 d1 = d0, r1 = r0;
    if (more) { d1 = ld4(db + vn * dld); r1 = ld4(rb + vn * tm.rld); }
    const float d[4] = {d0.x, d0.y, d0.z, d0.w}, r[4] = {r0.x, r0.y, r0.z, r0.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float z = fmaf(a4[j], r[j], b4[j]);
      const float g = z > thr ? d[j] : 0.f;
      o[j] = fmaf(A4[j], g, fmaf(C4[j], r[j], B4[j]));
    }
    st4(ob + v * tm.drld, make_float4(o[0], o[1], o[2], o[3]));
    d0 = d1; r0 = r1;
  }
}

// The apply passes of a node level's SINGLE primitives (SE gates, identity-with-norm) in one launch: grid (rows, B, targets); a target is an
// input-gradient tensor and takes the sum of 1..4 terms -- x = t_0 (+ previous content); x = t_k + x -- i.e. exactly what the separate
// launches of affine_bwd_apply_kernel leave behind, in the same order (terms of different targets never meet).
struct ApplySumArgs { BwdApplyTerm t[N3D_MAX_REDUCE_TERMS]; int start[8], count[8], acc[8]; };
__global__ __launch_bounds__(256) void affine_bwd_apply_sum_kernel(const float* __restrict__ dout, int64_t dld, ApplySumArgs ts, int64_t N, int C,
                                                                   EwMap m) { N3D_CHAIN_PRIO();
  int start, count, acc;
  N3D_PICK8(ts.start, blockIdx.z, start);
  N3D_PICK8(ts.count, blockIdx.z, count);
  N3D_PICK8(ts.acc, blockIdx.z, acc);
  const int b = blockIdx.y, t = threadIdx.x;
  const int vl = (int)m.fcpb.div((uint32_t)t), c4 = t - vl * m.cpb;
  if (vl >= m.vpb) return;
  const int64_t v0 = (int64_t)blockIdx.x * m.vpc + vl;
  if (v0 >= N) return;
  const int co = b * C + c4 * 4;
  const float* db = dout + (int64_t)b * N * dld + c4 * 4;
  BwdApplyTerm tm[4];
  float4 fa[4], fb[4], qA[4], qB[4], qC[4];
  float thr[4];
  bool mask[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    N3D_PICK16(ts.t, start + (k < count ? k : 0), tm[k]);
    mask[k] = tm[k].relu && tm[k].a;
    fa[k] = tm[k].a ? ld4(tm[k].a + co) : make_float4(1.f, 1.f, 1.f, 1.f);
    fb[k] = tm[k].b ? ld4(tm[k].b + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    qA[k] = tm[k].cA ? ld4(tm[k].cA + co) : make_float4(1.f, 1.f, 1.f, 1.f);
    qB[k] = tm[k].cB ? ld4(tm[k].cB + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    qC[k] = tm[k].cC ? ld4(tm[k].cC + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    thr[k] = 0.f;
  }
  float* ob = tm[0].draw + (int64_t)b * N * tm[0].drld + c4 * 4;
  for (int it = 0; it < m.iters; ++it) {
    const int64_t v = v0 + (int64_t)it * m.vpb;
    if (v >= N) break;
    const float4 dq = ld4(db + v * dld);
    float4 rq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < count) rq[k] = ld4(tm[k].raw + (int64_t)b * N * tm[k].rld + c4 * 4 + v * tm[k].rld);
    float4 pq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (acc) pq = ld4(ob + v * tm[0].drld);
    const float d[4] = {dq.x, dq.y, dq.z, dq.w};
    float x[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < count) {
        const float r[4] = {rq[k].x, rq[k].y, rq[k].z, rq[k].w};
        const float a4[4] = {fa[k].x, fa[k].y, fa[k].z, fa[k].w}, b4[4] = {fb[k].x, fb[k].y, fb[k].z, fb[k].w};
        const float A4[4] = {qA[k].x, qA[k].y, qA[k].z, qA[k].w}, B4[4] = {qB[k].x, qB[k].y, qB[k].z, qB[k].w}, C4[4] = {qC[k].x, qC[k].y, qC[k].z, qC[k].w};
        const float p4[4] = {pq.x, pq.y, pq.z, pq.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float g = d[j];
          if (mask[k]) { const float z = fmaf(a4[j], r[j], b4[j]); g = z > thr[k] ? g : 0.f; }
          float o = fmaf(A4[j], g, fmaf(C4[j], r[j], B4[j]));
          if (k == 0) { if (acc) o += p4[j]; x[j] = o; }
          else x[j] = o + x[j];
        }
      }
    }
    st4(ob + v * tm[0].drld, make_float4(x[0], x[1], x[2], x[3]));
  }
}

// ------------------------------------------------------------------------------------------------
// SE gate
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void se_gate_fwd_body(const double* __restrict__ stats, int rows, double count, const float* __restrict__ w1,
                                                 const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                 int C, float* __restrict__ mean, float* __restrict__ hidden, float* __restrict__ gate) {
  __shared__ double part[256];
  __shared__ double tot[192];
  __shared__ float hsh;
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  __shared__ float msh[64], w1sh[64];
  const float w1_t = (t < C) ? w1[t] : 0.f, w2_t = (t < C) ? w2[t] : 0.f, b2_t = (t < C) ? b2[t] : 0.f;   // issued before the row loads
  const float b1_0 = b1[0];
  reduce_rows(stats + (int64_t)b * rows * C * 2, rows, C * 2, part, tot);
  if (t < C) {
    // the channel means (a double division each) in parallel; the hidden unit below adds them in channel order as before
    const float m = (float)(tot[t * 2] / count);
    mean[b * C + t] = m;
    msh[t] = m; w1sh[t] = w1_t;
  }
  __syncthreads();
  if (t == 0) {
    float h = b1_0;
    for (int c = 0; c < C; ++c) h = fmaf(w1sh[c], msh[c], h);
    h = fmaxf(h, 0.f);
    hidden[b] = h;
    hsh = h;
  }
  __syncthreads();
  if (t < C) {
    const float pre = fmaf(w2_t, hsh, b2_t);
    gate[b * C + t] = 1.0f / (1.0f + __expf(-pre));
  }
}
__global__ __launch_bounds__(256) void se_gate_fwd_kernel(const double* __restrict__ stats, int rows, double count, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                          int C, float* __restrict__ mean, float* __restrict__ hidden, float* __restrict__ gate) { N3D_CHAIN_PRIO();
  se_gate_fwd_body(stats, rows, count, w1, b1, w2, b2, C, mean, hidden, gate);
}
// the SE gates of up to 8 primitives of a supernet node: grid (B, gates) forward, grid (gates) backward
struct SeTerm {
  const double* sums; int rows; const float* w1; const float* b1; const float* w2; const float* b2; float* mean; float* hidden; float* gate;
  const float* wptr; float* dw1; float* db1; float* dw2; float* db2; float* dalpha; float* A; float* Bc;
};
struct SeTermN { SeTerm t[8]; };
__global__ __launch_bounds__(256) void se_gate_fwdN_kernel(SeTermN ts, double count, int C) { N3D_CHAIN_PRIO();
  SeTerm q;
  N3D_PICK8(ts.t, blockIdx.y, q);
  se_gate_fwd_body(q.sums, q.rows, count, q.w1, q.b1, q.w2, q.b2, C, q.mean, q.hidden, q.gate);
}
// the GroupNorm coefficients AND the SE gates of a supernet node's group in one launch: grid (B, n_gn + n_se), the same bodies
struct NodeFwdCoefArgs { GnCoefArgs g[8]; SeTerm s[8]; int n_gn; };
__global__ __launch_bounds__(256) void node_fwd_coeffs_kernel(NodeFwdCoefArgs qs, int C, int G, double count, float eps) { N3D_CHAIN_PRIO();
  const int i = blockIdx.y;
  if (i < qs.n_gn) {
    GnCoefArgs q;
    N3D_PICK8(qs.g, i, q);
    gn_coeffs_body(q, C, G, count, eps);
  } else {
    SeTerm q;
    N3D_PICK8(qs.s, i - qs.n_gn, q);
    se_gate_fwd_body(q.sums, q.rows, count, q.w1, q.b1, q.w2, q.b2, C, q.mean, q.hidden, q.gate);
  }
}

__device__ __forceinline__ void se_gate_bwd_body(const double* __restrict__ sums, int rows, const float* __restrict__ wptr,
                                                 const float* __restrict__ mean, const float* __restrict__ hidden,
                                                 const float* __restrict__ gate, const float* __restrict__ w1,
                                                 const float* __restrict__ w2, int B, int C, double count, float* __restrict__ dw1,
                                                 float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                                 float* __restrict__ dalpha, float* __restrict__ A, float* __restrict__ Bc) {
  __shared__ double part[256];
  __shared__ double tot[192];
  __shared__ double red[64];
  __shared__ double dpre1_sh;
  const int t = threadIdx.x;
  const double w = wptr ? (double)*wptr : 1.0;
  double a_dw1 = 0, a_dw2 = 0, a_db2 = 0, a_db1 = 0, a_dz = 0;
  for (int b = 0; b < B; ++b) {
    reduce_rows(sums + (int64_t)b * rows * C * 3, rows, C * 3, part, tot);
    double dpre2 = 0;
    if (t < C) {
      const double g = gate[b * C + t];
      const double dgate = w * tot[t * 3 + 1];
      dpre2 = dgate * g * (1.0 - g);
      a_dz += tot[t * 3 + 2];
      a_dw2 += dpre2 * (double)hidden[b];
      a_db2 += dpre2;
    }
    if (t < 64) red[t] = (t < C) ? dpre2 * (double)w2[t] : 0.0;
    __syncthreads();
    if (t == 0) {
      double dh = 0;
      for (int i = 0; i < 64; ++i) dh += red[i];
      dpre1_sh = hidden[b] > 0.f ? dh : 0.0;
    }
    __syncthreads();
    const double dpre1 = dpre1_sh;
    if (t < C) {
      a_dw1 += dpre1 * (double)mean[b * C + t];
      A[b * C + t] = (float)(w * (double)gate[b * C + t]);
      Bc[b * C + t] = (float)(dpre1 * (double)w1[t] / count);
    }
    if (t == 0) a_db1 += dpre1;
    __syncthreads();
  }
  if (t < C) { dw1[t] = (float)a_dw1; dw2[t] = (float)a_dw2; db2[t] = (float)a_db2; }
  if (t == 0) db1[0] = (float)a_db1;
  if (dalpha) {
    if (t < 64) red[t] = (t < C) ? a_dz : 0.0;
    __syncthreads();
    if (t == 0) {
      double s = 0;
      for (int i = 0; i < 64; ++i) s += red[i];
      *dalpha = (float)s;
    }
  }
}
// B == 2: the two samples side by side in one 512-thread workgroup (slice bl = tid / 256 takes sample bl): ONE dependent chain rows ->
// gate gradient -> hidden gradient -> coefficients instead of two in a row; the per-channel sums meet in LDS and are added in
// sample order, i.e. the same additions in the same order as the loop above (0 + s0 + s1)
__device__ __forceinline__ void se_gate_bwd_body2(const double* __restrict__ sums, int rows, const float* __restrict__ wptr,
                                                  const float* __restrict__ mean, const float* __restrict__ hidden,
                                                  const float* __restrict__ gate, const float* __restrict__ w1,
                                                  const float* __restrict__ w2, int C, double count, float* __restrict__ dw1,
                                                  float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                                  float* __restrict__ dalpha, float* __restrict__ A, float* __restrict__ Bc) {
  __shared__ double part[2][256];
  __shared__ double tot[2][192];
  __shared__ double red[2][64];
  __shared__ double dpre1_sh[2];
  __shared__ double acc[2][4][64];   // dw1, dw2, db2, dz of each sample
  const int b = threadIdx.x >> 8, t = threadIdx.x & 255;
  const double w = wptr ? (double)*wptr : 1.0;
  reduce_rows_b(sums + (int64_t)b * rows * C * 3, rows, C * 3, part[b], tot[b], t, true);
  double dpre2 = 0, s_dz = 0, s_dw2 = 0;
  if (t < C) {
    const double g = gate[b * C + t];
    const double dgate = w * tot[b][t * 3 + 1];
    dpre2 = dgate * g * (1.0 - g);
    s_dz = tot[b][t * 3 + 2];
    s_dw2 = dpre2 * (double)hidden[b];
  }
  if (t < 64) red[b][t] = (t < C) ? dpre2 * (double)w2[t] : 0.0;
  __syncthreads();
  if (t == 0) {
    double dh = 0;
    for (int i = 0; i < 64; ++i) dh += red[b][i];
    dpre1_sh[b] = hidden[b] > 0.f ? dh : 0.0;
  }
  __syncthreads();
  const double dpre1 = dpre1_sh[b];
  if (t < 64) {
    const bool in = t < C;
    acc[b][0][t] = in ? dpre1 * (double)mean[b * C + t] : 0.0;
    acc[b][1][t] = in ? s_dw2 : 0.0;
    acc[b][2][t] = in ? dpre2 : 0.0;
    acc[b][3][t] = in ? s_dz : 0.0;
    if (in) {
      A[b * C + t] = (float)(w * (double)gate[b * C + t]);
      Bc[b * C + t] = (float)(dpre1 * (double)w1[t] / count);
    }
  }
  __syncthreads();
  if (b == 0 && t < C) {
    dw1[t] = (float)(acc[0][0][t] + acc[1][0][t]); dw2[t] = (float)(acc[0][1][t] + acc[1][1][t]); db2[t] = (float)(acc[0][2][t] + acc[1][2][t]);
  }
  if (threadIdx.x == 0) db1[0] = (float)(dpre1_sh[0] + dpre1_sh[1]);
  if (dalpha && threadIdx.x == 0) {
    double s = 0;
    for (int i = 0; i < 64; ++i) s += acc[0][3][i] + acc[1][3][i];
    *dalpha = (float)s;
  }
}
__global__ __launch_bounds__(256) void se_gate_bwd_kernel(const double* __restrict__ sums, int rows, const float* __restrict__ wptr,
                                                          const float* __restrict__ mean, const float* __restrict__ hidden,
                                                          const float* __restrict__ gate, const float* __restrict__ w1,
                                                          const float* __restrict__ w2, int B, int C, double count, float* __restrict__ dw1,
                                                          float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                                          float* __restrict__ dalpha, float* __restrict__ A, float* __restrict__ Bc) { N3D_CHAIN_PRIO();
  se_gate_bwd_body(sums, rows, wptr, mean, hidden, gate, w1, w2, B, C, count, dw1, db1, dw2, db2, dalpha, A, Bc);
}
__global__ __launch_bounds__(256) void se_gate_bwdN_kernel(SeTermN ts, int B, int C, double count) { N3D_CHAIN_PRIO();
  SeTerm q;
  N3D_PICK8(ts.t, blockIdx.x, q);
  se_gate_bwd_body(q.sums, q.rows, q.wptr, q.mean, q.hidden, q.gate, q.w1, q.w2, B, C, count, q.dw1, q.db1, q.dw2, q.db2, q.dalpha, q.A, q.Bc);
}
__global__ __launch_bounds__(512) void se_gate_bwdN2_kernel(SeTermN ts, int C, double count) { N3D_CHAIN_PRIO();
  SeTerm q;
  N3D_PICK8(ts.t, blockIdx.x, q);
  se_gate_bwd_body2(q.sums, q.rows, q.wptr, q.mean, q.hidden, q.gate, q.w1, q.w2, C, count, q.dw1, q.db1, q.dw2, q.db2, q.dalpha, q.A, q.Bc);
}
__global__ __launch_bounds__(512) void se_gate_bwd2_kernel(SeTerm q, int C, double count) { N3D_CHAIN_PRIO();
  se_gate_bwd_body2(q.sums, q.rows, q.wptr, q.mean, q.hidden, q.gate, q.w1, q.w2, C, count, q.dw1, q.db1, q.dw2, q.db2, q.dalpha, q.A, q.Bc);
}
// Every coefficient computation of a node level of the supernet backward in ONE launch (B = 2): workgroups [0, n_gn) are GroupNorm terms
// (gn_bwd_coeffs_body), [n_gn, n_gn + n_se) SE gates (se_gate_bwd_body2) -- the same bodies, i.e. the same bits, as the two launches
struct NodeCoefArgs { GnBwdCoefArgs g[N3D_MAX_REDUCE_TERMS]; SeTerm s[8]; int n_gn; };
__global__ __launch_bounds__(512) void node_bwd_coeffs_kernel(NodeCoefArgs qs, int C, int G, double count) { N3D_CHAIN_PRIO();
  const int i = blockIdx.x;
  if (i < qs.n_gn) {
    GnBwdCoefArgs q;
    N3D_PICK16(qs.g, i, q);
    gn_bwd_coeffs_body<2>(q, 2, C, G, count);
  } else {
    SeTerm q;
    N3D_PICK8(qs.s, i - qs.n_gn, q);
    se_gate_bwd_body2(q.sums, q.rows, q.wptr, q.mean, q.hidden, q.gate, q.w1, q.w2, C, count, q.dw1, q.db1, q.dw2, q.db2, q.dalpha, q.A, q.Bc);
  }
}

// ------------------------------------------------------------------------------------------------
// pooling 2x2x2 / 2
// ------------------------------------------------------------------------------------------------
template <bool MAX, typename T = float>
__global__ __launch_bounds__(256) void pool2_fwd_kernel(const T* __restrict__ x, int64_t xld, T* __restrict__ y, int64_t yld, int Di,
                                                        int Hi, int Wi, int C) { N3D_CHAIN_PRIO();
  const int Do = Di / 2, Ho = Hi / 2, Wo = Wi / 2;
  const int cpb = C / 4;
  const int64_t total = (int64_t)Do * Ho * Wo * cpb;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = blockIdx.y;
  const int c4 = idx % cpb;
  int64_t v = idx / cpb;
  const int wo = v % Wo; v /= Wo;
  const int ho = v % Ho;
  const int d_o = v / Ho;
  const T* xb = x + (int64_t)b * Di * Hi * Wi * xld + c4 * 4;
  float4 acc = MAX ? make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY) : make_float4(0, 0, 0, 0);
#pragma unroll
  for (int kd = 0; kd < 2; ++kd)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) {
        const int64_t vi = ((int64_t)(2 * d_o + kd) * Hi + (2 * ho + kh)) * Wi + (2 * wo + kw);
        const float4 q = ld4(xb + vi * xld);
        if (MAX) { acc.x = fmaxf(acc.x, q.x); acc.y = fmaxf(acc.y, q.y); acc.z = fmaxf(acc.z, q.z); acc.w = fmaxf(acc.w, q.w); }
        else { acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
      }
  if (!MAX) { acc.x *= 0.125f; acc.y *= 0.125f; acc.z *= 0.125f; acc.w *= 0.125f; }
  const int64_t vo = ((int64_t)d_o * Ho + ho) * Wo + wo;
  st4(y + ((int64_t)b * Do * Ho * Wo + vo) * yld + c4 * 4, acc);
}

// average AND max pooling of one tensor in one pass (a stride-2 edge of a down cell carries both primitives, prim_ops.py:29-30)
__global__ __launch_bounds__(256) void pool2_fwd_both_kernel(const float* __restrict__ x, int64_t xld, float* __restrict__ ya, int64_t yald,
                                                             float* __restrict__ ym, int64_t ymld, int Di, int Hi, int Wi, int C) { N3D_CHAIN_PRIO();
  const int Do = Di / 2, Ho = Hi / 2, Wo = Wi / 2;
  const int cpb = C / 4;
  const int64_t total = (int64_t)Do * Ho * Wo * cpb;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = blockIdx.y;
  const int c4 = idx % cpb;
  int64_t v = idx / cpb;
  const int wo = v % Wo; v /= Wo;
  const int ho = v % Ho;
  const int d_o = v / Ho;
  const float* xb = x + (int64_t)b * Di * Hi * Wi * xld + c4 * 4;
  float4 q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t vi = ((int64_t)(2 * d_o + (k >> 2)) * Hi + (2 * ho + ((k >> 1) & 1))) * Wi + (2 * wo + (k & 1));
    q[k] = ld4(xb + vi * xld);
  }
  float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sm = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int k = 0; k < 8; ++k) {   // same summation / comparison order as the two single kernels
    sa.x += q[k].x; sa.y += q[k].y; sa.z += q[k].z; sa.w += q[k].w;
    sm.x = fmaxf(sm.x, q[k].x); sm.y = fmaxf(sm.y, q[k].y); sm.z = fmaxf(sm.z, q[k].z); sm.w = fmaxf(sm.w, q[k].w);
  }
  sa.x *= 0.125f; sa.y *= 0.125f; sa.z *= 0.125f; sa.w *= 0.125f;
  const int64_t vo = (int64_t)b * Do * Ho * Wo + ((int64_t)d_o * Ho + ho) * Wo + wo;
  st4(ya + vo * yald + c4 * 4, sa);
  st4(ym + vo * ymld + c4 * 4, sm);
}

// dx (+)= w_avg * avgpool^T(dy) + w_max * maxpool^T(dy): both pooling backwards of an edge in one pass over dx
template <bool ACC>
__global__ __launch_bounds__(256) void pool2_bwd_both_kernel(const float* __restrict__ dy, int64_t dyld, const float* __restrict__ x, int64_t xld,
                                                             float* __restrict__ dx, int64_t dxld, int Di, int Hi, int Wi, int C,
                                                             const float* __restrict__ wa, const float* __restrict__ wm) { N3D_CHAIN_PRIO();
  const int Do = Di / 2, Ho = Hi / 2, Wo = Wi / 2;
  const int cpb = C / 4;
  const int64_t total = (int64_t)Do * Ho * Wo * cpb;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = blockIdx.y;
  const int c4 = idx % cpb;
  int64_t v = idx / cpb;
  const int wo = v % Wo; v /= Wo;
  const int ho = v % Ho;
  const int d_o = v / Ho;
  const int64_t vo = ((int64_t)d_o * Ho + ho) * Wo + wo;
  const float4 gq = ld4(dy + ((int64_t)b * Do * Ho * Wo + vo) * dyld + c4 * 4);
  const float sa = (wa ? *wa : 1.0f) * 0.125f, sm = wm ? *wm : 1.0f;
  const float g[4] = {gq.x, gq.y, gq.z, gq.w};
  const float* xb = x + (int64_t)b * Di * Hi * Wi * xld + c4 * 4;
  float* db = dx + (int64_t)b * Di * Hi * Wi * dxld + c4 * 4;
  int arg[4] = {0, 0, 0, 0};
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  float4 prev[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t vi = ((int64_t)(2 * d_o + (k >> 2)) * Hi + (2 * ho + ((k >> 1) & 1))) * Wi + (2 * wo + (k & 1));
    const float4 q = ld4(xb + vi * xld);
    if (ACC) prev[k] = ld4(db + vi * dxld);
    const float qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (qq[j] > best[j] || qq[j] != qq[j]) { best[j] = qq[j]; arg[j] = k; }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t vi = ((int64_t)(2 * d_o + (k >> 2)) * Hi + (2 * ho + ((k >> 1) & 1))) * Wi + (2 * wo + (k & 1));
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = g[j] * sa + (arg[j] == k ? g[j] * sm : 0.f);
    if (ACC) { o[0] += prev[k].x; o[1] += prev[k].y; o[2] += prev[k].z; o[3] += prev[k].w; }
    st4(db + vi * dxld, make_float4(o[0], o[1], o[2], o[3]));
  }
}

// one thread per OUTPUT voxel quad: routes dy to its 8 inputs (avg: /8; max: first arg-max in
// (d,h,w) scan order, the choice torch's max_pool3d backward makes)
template <bool MAX, bool ACC, typename T = float>
__global__ __launch_bounds__(256) void pool2_bwd_kernel(const T* __restrict__ dy, int64_t dyld, const T* __restrict__ x, int64_t xld,
                                                        T* __restrict__ dx, int64_t dxld, int Di, int Hi, int Wi, int C,
                                                        const float* __restrict__ wptr) { N3D_CHAIN_PRIO();
  const int Do = Di / 2, Ho = Hi / 2, Wo = Wi / 2;
  const int cpb = C / 4;
  const int64_t total = (int64_t)Do * Ho * Wo * cpb;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int b = blockIdx.y;
  const int c4 = idx % cpb;
  int64_t v = idx / cpb;
  const int wo = v % Wo; v /= Wo;
  const int ho = v % Ho;
  const int d_o = v / Ho;
  const int64_t vo = ((int64_t)d_o * Ho + ho) * Wo + wo;
  const float4 gq = ld4(dy + ((int64_t)b * Do * Ho * Wo + vo) * dyld + c4 * 4);
  const float wsc = wptr ? *wptr : 1.0f;   // MixedOp weight of the pooling primitive: dx (+)= w * pool^T(dy)
  const float g[4] = {gq.x * wsc, gq.y * wsc, gq.z * wsc, gq.w * wsc};
  const T* xb = x ? x + (int64_t)b * Di * Hi * Wi * xld + c4 * 4 : nullptr;
  T* db = dx + (int64_t)b * Di * Hi * Wi * dxld + c4 * 4;
  int arg[4] = {0, 0, 0, 0};
  if (MAX) {
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t vi = ((int64_t)(2 * d_o + (k >> 2)) * Hi + (2 * ho + ((k >> 1) & 1))) * Wi + (2 * wo + (k & 1));
      const float4 q = ld4(xb + vi * xld);
      const float qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (qq[j] > best[j] || qq[j] != qq[j]) { best[j] = qq[j]; arg[j] = k; }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t vi = ((int64_t)(2 * d_o + (k >> 2)) * Hi + (2 * ho + ((k >> 1) & 1))) * Wi + (2 * wo + (k & 1));
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = MAX ? (arg[j] == k ? g[j] : 0.f) : g[j] * 0.125f;
    T* op = db + vi * dxld;
    if (ACC) { const float4 p = ld4(op); o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w; }
    st4(op, make_float4(o[0], o[1], o[2], o[3]));
  }
}

// ------------------------------------------------------------------------------------------------
// Dice
// ------------------------------------------------------------------------------------------------
#define DICE_CHUNK 4096
__global__ __launch_bounds__(256) void dice_reduce_kernel(const float* __restrict__ p, int64_t psb, int64_t psc, int64_t psv,
                                                          const float* __restrict__ t, int64_t tsb, int64_t tsc, int64_t tsv, int64_t N,
                                                          double* __restrict__ partial /*[B][C][rows][3]*/) {
  __shared__ double red[3][4];
  const int b = blockIdx.z, c = blockIdx.y;
  const float* pp = p + b * psb + c * psc;
  const float* tp = t + b * tsb + c * tsc;
  float spt = 0, sp = 0, st = 0;
  const int64_t v0 = (int64_t)blockIdx.x * DICE_CHUNK;
  for (int i = threadIdx.x; i < DICE_CHUNK; i += 256) {
    const int64_t v = v0 + i;
    if (v < N) {
      const float a = pp[v * psv], q = tp[v * tsv];
      spt = fmaf(a, q, spt); sp += a; st += q;
    }
  }
  double d0 = wave_sum_d(spt), d1 = wave_sum_d(sp), d2 = wave_sum_d(st);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = d0; red[1][wave] = d1; red[2][wave] = d2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double* o = partial + (((int64_t)b * gridDim.y + c) * gridDim.x + blockIdx.x) * 3;
    o[threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
  }
}

__global__ __launch_bounds__(256) void dice_finalize_kernel(const double* __restrict__ partial, int rows, int BC, double smooth,
                                                            double* __restrict__ sums /*[BC][3]*/, float* __restrict__ loss) {
  __shared__ double ratio[256];
  const int t = threadIdx.x;
  double acc = 0;
  // one wave per (b, c) row: lanes stride over the partial rows (a single thread walking them pays one memory latency per
  // row: 64 rows at 64^3, 512 at 128^3), fixed-order lane sum
  const int wave = t >> 6, lane = t & 63;
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
    double s = 0;
    for (int i = 0; i < 256; ++i) s += ratio[i];
    *loss = (float)(1.0 - s / BC);
  }
}

__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ p, int64_t psb, int64_t psc, int64_t psv,
                                                       const float* __restrict__ t, int64_t tsb, int64_t tsc, int64_t tsv, int64_t N, int BC,
                                                       double smooth, const double* __restrict__ sums, const float* __restrict__ dloss,
                                                       float* __restrict__ dp, int64_t dsb, int64_t dsc, int64_t dsv) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int i = b * gridDim.y + c;
  const double num = 2.0 * sums[i * 3] + smooth, den = sums[i * 3 + 1] + sums[i * 3 + 2] + smooth;
  const float gl = dloss ? *dloss : 1.0f;
  // d loss / d p = -(1/BC) * (2 t den - num) / den^2
  const float k2 = (float)(-(double)gl / BC * 2.0 / den);
  const float k0 = (float)((double)gl / BC * num / (den * den));
  const float* tp = t + b * tsb + c * tsc;
  float* op = dp + b * dsb + c * dsc;
  const int64_t v0 = (int64_t)blockIdx.x * DICE_CHUNK;
  for (int j = threadIdx.x; j < DICE_CHUNK; j += 256) {
    const int64_t v = v0 + j;
    if (v < N) op[v * dsv] = fmaf(k2, tp[v * tsv], k0);
  }
}

// ------------------------------------------------------------------------------------------------
// layout
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ncdhw_to_ndhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t dld, int C, int64_t N) {
  const int b = blockIdx.y;
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= N) return;
  const float* s = src + (int64_t)b * C * N + v;
  float* d = dst + ((int64_t)b * N + v) * dld;
  for (int c = 0; c < C; ++c) d[c] = s[(int64_t)c * N];
}
__global__ __launch_bounds__(256) void ndhwc_to_ncdhw_kernel(const float* __restrict__ src, int64_t sld, float* __restrict__ dst, int C, int64_t N) {
  const int b = blockIdx.y;
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= N) return;
  const float* s = src + ((int64_t)b * N + v) * sld;
  float* d = dst + (int64_t)b * C * N + v;
  for (int c = 0; c < C; ++c) d[(int64_t)c * N] = s[c];
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam, amsgrad=False, maximize=False)
// ------------------------------------------------------------------------------------------------
struct AdamGuard { const unsigned* timeouts; const unsigned* acked; const float* peer_flag; float* loss; unsigned* host_word; };
__global__ void guard_flag_kernel(const unsigned* timeouts, const unsigned* acked, float* flag) {
  *flag = (__hip_atomic_load(timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != *acked) ? 1.f : 0.f;
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float lr_imm, const float* __restrict__ lr_ptr, float b1, float b2, float eps,
                                                   float wd, float gscale, int32_t* __restrict__ step_ptr, int ticketed, AdamGuard gd) {
  // guard (include/n3d.h, n3d_adam_step_guarded): a device-side wait of this step's stream hand-offs gave up (time-outs counted
  // != time-outs the host has acknowledged) or a peer rank reported one (all-reduced flag != 0) -> the gradients are not to be
  // trusted: NO update, moments and step counter untouched (the ticket counter stays 0), the loss scalar becomes NaN and the
  // host-visible word is set.  Uniform over the grid, two scalar loads on the good path.
  {
    bool skip = false;
    if (gd.timeouts) skip = __hip_atomic_load(gd.timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != *gd.acked;
    if (gd.peer_flag) skip = skip || (*gd.peer_flag != 0.f);
    if (skip) {
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (gd.loss) *gd.loss = __builtin_nanf("");
        if (gd.host_word) __hip_atomic_store(gd.host_word, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
  }
  const int step = *step_ptr + 1;
  const float lr = lr_ptr ? *lr_ptr : lr_imm;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    gi *= gscale;
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    mi = fmaf(b1, mi, (1.f - b1) * gi);
    vi = fmaf(b2, vi, (1.f - b2) * gi * gi);
    const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
    pi = pi - step_size * (mi / denom);
  };
  // ADAM_U float4 per lane and array, all requested before the first update (the flat buffers are 16-byte aligned; n % 4 == 0 by
  // construction of the flat layout, a ragged tail goes element by element)
  constexpr int ADAM_U = 2;
  const bool vec = (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0);
  const int64_t n4 = vec ? n / 4 : 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * ADAM_U) {
    float4 p4[ADAM_U], g4[ADAM_U], m4[ADAM_U], v4[ADAM_U];
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const int64_t i = i0 + u * stride;
      const int64_t ic = i < n4 ? i : i0;
      p4[u] = reinterpret_cast<const float4*>(p)[ic]; g4[u] = reinterpret_cast<const float4*>(g)[ic];
      m4[u] = reinterpret_cast<const float4*>(m)[ic]; v4[u] = reinterpret_cast<const float4*>(v)[ic];
    }
#pragma unroll
    for (int u = 0; u < ADAM_U; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n4) {
        upd(p4[u].x, g4[u].x, m4[u].x, v4[u].x); upd(p4[u].y, g4[u].y, m4[u].y, v4[u].y);
        upd(p4[u].z, g4[u].z, m4[u].z, v4[u].z); upd(p4[u].w, g4[u].w, m4[u].w, v4[u].w);
        reinterpret_cast<float4*>(m)[i] = m4[u]; reinterpret_cast<float4*>(v)[i] = v4[u]; reinterpret_cast<float4*>(p)[i] = p4[u];
      }
    }
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(pi, g[i], mi, vi);
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
  if (ticketed) {
    // inc_step == 2: step_ptr[1] is a ticket counter.  Every workgroup has read the step before it draws its ticket, so the
    // workgroup that draws the last one may count the step and clear the counter (no second launch)
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned tk = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(step_ptr + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tk == gridDim.x - 1) {
        __hip_atomic_store(reinterpret_cast<unsigned*>(step_ptr + 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(step_ptr, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
__global__ void step_inc_kernel(int32_t* step_ptr) { *step_ptr += 1; }

}  // namespace n3d

// GroupNorm over zero-PADDED channels (round 5; include/n3d.h, "padded channels"): a group count G < 0 means ONE group whose REAL channel
// count is -G, stored inside C >= -G channels of which the others are exactly zero (zero weights, gamma, beta).  Sums over the padded
// tensor are the real tensor's sums, so only the element count of the group changes: the kernels take it as count * (C / G) with G = 1,
// i.e. count = N * (-G) / C.
static inline double gn_count(int64_t N, int C, int G) { return G < 0 ? (double)N * (double)(-G) / (double)C : (double)N; }
static inline int gn_groups(int G) { return G < 0 ? 1 : G; }

using namespace n3d;

// ================================================================================================
// C ABI
// ================================================================================================
// dispatch on the storage type of a call's activation tensors: f receives a null pointer of the element type
template <typename F>
static void with_act_type(bool bf16, F&& f) {
  if (bf16) f(static_cast<bf16_t*>(nullptr));
  else f(static_cast<float*>(nullptr));
}
#define N3D_T(tag) std::remove_pointer_t<decltype(tag)>

extern "C" {

int n3d_stats_rows(int64_t N, int C) { return ew_map(N, C).rows; }
int n3d_dice_rows(int64_t N) { return (int)cdiv(N, DICE_CHUNK); }

static int check_vec(const void* p, int64_t ld, int C, const char* what, bool bf16 = false) {
  if (C % 4 != 0 || ld % 4 != 0 || (reinterpret_cast<uintptr_t>(p) & (bf16 ? 7 : 15)) != 0 || C > 64 * 4) {
    set_error("%s: needs C %% 4 == 0, ld %% 4 == 0, 16-byte aligned base (C=%d ld=%lld)", what, C, (long long)ld);
    return N3D_ERR_UNSUPPORTED;
  }
  return 0;
}

int n3d_channel_stats_t(const void* x, int64_t ld, int dtype, int B, int64_t N, int C, double* stats, void* stream) {
  N3D_CHECK_ARG(x && stats && B > 0 && N > 0 && C > 0 && C <= 64 && (dtype == N3D_F32 || dtype == N3D_BF16), "channel_stats: bad args");
  if (int e = check_vec(x, ld, C, "channel_stats", dtype == N3D_BF16)) return e;
  EwMap m = ew_map(N, C);
  with_act_type(dtype == N3D_BF16, [&](auto* tag) {
    using T = N3D_T(tag);
    hipLaunchKernelGGL(channel_stats_kernel<T>, dim3(m.rows, B), dim3(256), 0, (hipStream_t)stream, (const T*)x, ld, N, C, m, stats);
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}
int n3d_channel_stats(const float* x, int64_t ld, int B, int64_t N, int C, double* stats, void* stream) {
  return n3d_channel_stats_t(x, ld, N3D_F32, B, N, C, stats, stream);
}

int n3d_channel_statsN(const float* const* xs, const int64_t* lds, double* const* stats, int n, int B, int64_t N, int C, void* stream) {
  N3D_CHECK_ARG(xs && lds && stats && n >= 1 && n <= 8 && B > 0 && N > 0, "channel_statsN: bad args");
  StatsJobN js;
  for (int i = 0; i < 8; ++i) {
    const int k = i < n ? i : 0;
    N3D_CHECK_ARG(xs[k] && stats[k], "channel_statsN: null pointers");
    if (int e = check_vec(xs[k], lds[k], C, "channel_statsN(x)")) return e;
    js.x[i] = xs[k]; js.ld[i] = lds[k]; js.stats[i] = stats[k];
  }
  EwMap m = ew_map(N, C);
  hipLaunchKernelGGL(channel_statsN_kernel, dim3(m.rows, B, n), dim3(256), 0, (hipStream_t)stream, js, N, C, m);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_gn_coeffs(const double* stats, int rows, const float* gamma, const float* beta, int B, int C, int G, int64_t N, float eps,
                  float* a, float* b, float* mean_rstd, double* sumraw, void* stream) {
  const double gn_cnt_ = gn_count(N, C, G);      // G < 0: ONE group of -G real channels inside C zero-padded ones (include/n3d.h)
  G = gn_groups(G);
  N3D_CHECK_ARG(stats && gamma && beta && a && b && C <= 64 && G >= 1 && C % G == 0 && rows >= 1, "gn_coeffs: bad args");
  const GnCoefArgs q{stats, rows, gamma, beta, a, b, mean_rstd, sumraw};
  hipLaunchKernelGGL(gn_coeffs_kernel, dim3(B, 1), dim3(256), 0, (hipStream_t)stream, q, q, C, G, gn_cnt_, eps);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act(const float* raw, int64_t rld, const float* a, const float* b, const float* wptr, float* out, int64_t old_, int B,
                   int64_t N, int C, int flags, void* stream) {
  N3D_CHECK_ARG(raw && out && B > 0 && N > 0, "affine_act: bad args");
  const bool bf = flags & N3D_ACT_BF16;
  if (int e = check_vec(raw, rld, C, "affine_act(raw)", bf)) return e;
  if (int e = check_vec(out, old_, C, "affine_act(out)", bf)) return e;
  EwMap m = ew_map(N, C);
  dim3 grid(m.rows, B), blk(256);
  hipStream_t s = (hipStream_t)stream;
  const bool relu = flags & N3D_RELU, acc = flags & N3D_ACCUMULATE;
  with_act_type(bf, [&](auto* tag) {
    using T = N3D_T(tag);
    const T* r = (const T*)raw; T* o = (T*)out;
    if (relu && acc) hipLaunchKernelGGL((affine_act_kernel<true, true, T>), grid, blk, 0, s, r, rld, a, b, wptr, o, old_, N, C, m);
    else if (relu) hipLaunchKernelGGL((affine_act_kernel<true, false, T>), grid, blk, 0, s, r, rld, a, b, wptr, o, old_, N, C, m);
    else if (acc) hipLaunchKernelGGL((affine_act_kernel<false, true, T>), grid, blk, 0, s, r, rld, a, b, wptr, o, old_, N, C, m);
    else hipLaunchKernelGGL((affine_act_kernel<false, false, T>), grid, blk, 0, s, r, rld, a, b, wptr, o, old_, N, C, m);
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act_bwd_reduce(const float* dout, int64_t dld, const float* raw, int64_t rld, const float* a, const float* b, int B,
                              int64_t N, int C, int flags, double* sums, void* stream) {
  N3D_CHECK_ARG(dout && raw && sums && C <= 64, "affine_act_bwd_reduce: bad args");
  const bool bf = flags & N3D_ACT_BF16;
  if (int e = check_vec(dout, dld, C, "bwd_reduce(dout)", bf)) return e;
  if (int e = check_vec(raw, rld, C, "bwd_reduce(raw)", bf)) return e;
  EwMap m = ew_map(N, C);
  dim3 grid(m.rows, B), blk(256);
  with_act_type(bf, [&](auto* tag) {
    using T = N3D_T(tag);
    if (flags & N3D_RELU) hipLaunchKernelGGL((affine_bwd_reduce_kernel<true, T>), grid, blk, 0, (hipStream_t)stream, (const T*)dout, dld, (const T*)raw, rld, a, b, N, C, m, sums);
    else hipLaunchKernelGGL((affine_bwd_reduce_kernel<false, T>), grid, blk, 0, (hipStream_t)stream, (const T*)dout, dld, (const T*)raw, rld, a, b, N, C, m, sums);
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_gn_bwd_coeffs(const double* sums, int rows, const float* gamma, const float* mean_rstd, const float* wptr, int B, int C, int G,
                      int64_t N, float* dgamma, float* dbeta, float* dalpha, float* A, float* Bc, float* Cc, const double* sumraw,
                      float* dbias_conv, void* stream) {
  const double gn_cnt_ = gn_count(N, C, G);      // G < 0: ONE group of -G real channels inside C zero-padded ones (include/n3d.h)
  G = gn_groups(G);
  N3D_CHECK_ARG(sums && gamma && mean_rstd && A && Bc && Cc && C <= 64 && C % G == 0, "gn_bwd_coeffs: bad args");
  N3D_CHECK_ARG(!dbias_conv || sumraw, "gn_bwd_coeffs: dbias_conv needs the forward per-channel sums");
  const GnBwdCoefArgs q{sums, rows, gamma, mean_rstd, wptr, dgamma, dbeta, dalpha, A, Bc, Cc, sumraw, dbias_conv};
  if (B <= 2) hipLaunchKernelGGL(gn_bwd_coeffs_kernel<2>, dim3(1), dim3(512), 0, (hipStream_t)stream, q, q, B, C, G, gn_cnt_);
  else hipLaunchKernelGGL(gn_bwd_coeffs_kernel<GNB_BP>, dim3(1), dim3(256 * GNB_BP), 0, (hipStream_t)stream, q, q, B, C, G, gn_cnt_);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

static bool pair_shape_ok(int C, int G) { return C >= 4 && C <= 64 && (C & (C - 1)) == 0 && G >= 1 && C % G == 0 && C / G <= 16; }

int n3d_affine_act_gn2(const n3d_gn_fwd_term* t0, const n3d_gn_fwd_term* t1, int G, float eps, float* out, int64_t old_, float* out1,
                       int64_t old1, int B, int64_t N, int C, int flags, void* stream) {
  const double gn_cnt_ = gn_count(N, C, G);      // G < 0: ONE group of -G real channels inside C zero-padded ones (include/n3d.h)
  G = gn_groups(G);
  N3D_CHECK_ARG(t0 && t1 && out && B > 0 && N > 0, "affine_act_gn2: bad args");
  if (!pair_shape_ok(C, G) || t0->rows < 1 || t1->rows < 1 || t0->rows > n3d_fused_max_rows() || t1->rows > n3d_fused_max_rows())
    N3D_UNSUPPORTED("affine_act_gn2: shape not supported by the pair kernel (C=%d G=%d rows=%d/%d)", C, G, t0->rows, t1->rows);
  GnFwdTerm k[2];
  const n3d_gn_fwd_term* ts[2] = {t0, t1};
  for (int i = 0; i < 2; ++i) {
    const n3d_gn_fwd_term* t = ts[i];
    N3D_CHECK_ARG(t->raw && t->stats && t->gamma && t->beta && t->a_out && t->b_out && t->mean_rstd_out, "affine_act_gn2: null term pointer");
    if (int e = check_vec(t->raw, t->rld, C, "affine_act_gn2(raw)", flags & N3D_ACT_BF16)) return e;
    k[i] = GnFwdTerm{t->raw, t->rld, t->stats, t->rows, t->gamma, t->beta, t->wptr, t->a_out, t->b_out, t->mean_rstd_out, t->sumraw, t->relu};
  }
  const bool bf = flags & N3D_ACT_BF16;
  if (int e = check_vec(out, old_, C, "affine_act_gn2(out)", bf)) return e;
  if (out1) { if (int e = check_vec(out1, old1, C, "affine_act_gn2(out1)", bf)) return e; }
  EwMap m = ew_map(N, C);
  dim3 grid(m.rows, B), blk(256);
  hipStream_t s = (hipStream_t)stream;
  with_act_type(bf, [&](auto* tag) {
    using T = N3D_T(tag);
    if (flags & N3D_ACCUMULATE) hipLaunchKernelGGL((affine_act_gn2_kernel<true, false, T>), grid, blk, 0, s, k[0], k[1], G, gn_cnt_, eps, (T*)out, old_, (T*)out1, old1, N, C, m);
    else hipLaunchKernelGGL((affine_act_gn2_kernel<false, false, T>), grid, blk, 0, s, k[0], k[1], G, gn_cnt_, eps, (T*)out, old_, (T*)out1, old1, N, C, m);
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act_bwd_reduce2(const float* dout, int64_t dld, const float* dout1, int64_t dld1, const n3d_gn_bwd_term* t0,
                               const n3d_gn_bwd_term* t1, int B, int64_t N, int C, void* stream) {
  N3D_CHECK_ARG(dout && t0 && t1 && B > 0 && N > 0 && C <= 64 && t0->dtype == t1->dtype, "affine_act_bwd_reduce2: bad args");
  const bool bf = t0->dtype == N3D_BF16;
  if (int e = check_vec(dout, dld, C, "bwd_reduce2(dout)", bf)) return e;
  EwMap m = ew_map(N, C);
  if ((m.cpb & (m.cpb - 1)) != 0) N3D_UNSUPPORTED("affine_act_bwd_reduce2: C / 4 must be a power of two");
  BwdRedTerm k[2];
  const n3d_gn_bwd_term* ts[2] = {t0, t1};
  for (int i = 0; i < 2; ++i) {
    const n3d_gn_bwd_term* t = ts[i];
    N3D_CHECK_ARG(t->raw && t->a && t->b && t->sums, "affine_act_bwd_reduce2: null term pointer");
    if (int e = check_vec(t->raw, t->rld, C, "bwd_reduce2(raw)", bf)) return e;
    k[i] = BwdRedTerm{t->raw, t->rld, t->a, t->b, t->sums, t->relu};
  }
  if (dout1) { if (int e = check_vec(dout1, dld1, C, "bwd_reduce2(dout1)", bf)) return e; }
  with_act_type(bf, [&](auto* tag) {
    using T = N3D_T(tag);
    if (dout1) hipLaunchKernelGGL((affine_bwd_reduce2_kernel<true, T>), dim3(m.rows, B), dim3(256), 0, (hipStream_t)stream, (const T*)dout, dld, (const T*)dout1, dld1, k[0], k[1], N, C, m);
    else hipLaunchKernelGGL((affine_bwd_reduce2_kernel<false, T>), dim3(m.rows, B), dim3(256), 0, (hipStream_t)stream, (const T*)dout, dld, (const T*)dout1, dld1, k[0], k[1], N, C, m);
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act_bwd_apply_gn2(const float* dout, int64_t dld, const float* dout1, int64_t dld1, const n3d_gn_bwd_term* t0,
                                 const n3d_gn_bwd_term* t1, int B, int64_t N, int C, int G, void* stream) {
  const double gn_cnt_ = gn_count(N, C, G);      // G < 0: ONE group of -G real channels inside C zero-padded ones (include/n3d.h)
  G = gn_groups(G);
  N3D_CHECK_ARG(dout && t0 && t1 && B > 0 && N > 0, "affine_act_bwd_apply_gn2: bad args");
  if (!pair_shape_ok(C, G) || t0->rows < 1 || t1->rows < 1 || t0->rows > n3d_fused_max_rows() || t1->rows > n3d_fused_max_rows())
    N3D_UNSUPPORTED("affine_act_bwd_apply_gn2: shape not supported by the pair kernel (C=%d G=%d rows=%d/%d)", C, G, t0->rows, t1->rows);
  N3D_CHECK_ARG(t0->dtype == t1->dtype, "affine_act_bwd_apply_gn2: both terms must share one storage type");
  const bool bf = t0->dtype == N3D_BF16;
  if (int e = check_vec(dout, dld, C, "bwd_apply_gn2(dout)", bf)) return e;
  GnBwdTerm k[2];
  const n3d_gn_bwd_term* ts[2] = {t0, t1};
  for (int i = 0; i < 2; ++i) {
    const n3d_gn_bwd_term* t = ts[i];
    N3D_CHECK_ARG(t->raw && t->a && t->b && t->sums && t->gamma && t->mean_rstd && t->draw, "affine_act_bwd_apply_gn2: null term pointer");
    N3D_CHECK_ARG(!t->dbias_conv || t->sumraw, "affine_act_bwd_apply_gn2: dbias_conv needs the forward per-channel sums");
    if (int e = check_vec(t->raw, t->rld, C, "bwd_apply_gn2(raw)", bf)) return e;
    if (int e = check_vec(t->draw, t->drld, C, "bwd_apply_gn2(draw)", bf)) return e;
    k[i] = GnBwdTerm{t->raw, t->rld, t->a, t->b, t->sums, t->rows, t->gamma, t->mean_rstd, t->wptr, t->sumraw, t->draw, t->drld,
                     t->dgamma, t->dbeta, t->dalpha, t->dbias_conv, t->relu, nullptr, nullptr, nullptr};
  }
  EwMap m = ew_map(N, C);
  if (dout1) { if (int e = check_vec(dout1, dld1, C, "bwd_apply_gn2(dout1)", bf)) return e; }
  with_act_type(bf, [&](auto* tag) {
    using T = N3D_T(tag);
    // (grid.x = rows + 1: the last workgroup column forms the parameter gradients)
    if (dout1) hipLaunchKernelGGL((affine_bwd_apply_gn2_kernel<false, true, T>), dim3(m.rows + 1, B), dim3(256), 0, (hipStream_t)stream, (const T*)dout, dld, (const T*)dout1, dld1, k[0], k[1], B, G, gn_cnt_, N, C, m);
    else hipLaunchKernelGGL((affine_bwd_apply_gn2_kernel<false, false, T>), dim3(m.rows + 1, B), dim3(256), 0, (hipStream_t)stream, (const T*)dout, dld, (const T*)dout1, dld1, k[0], k[1], B, G, gn_cnt_, N, C, m);
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

// 1 if n3d_affine_act_bwd_small2 takes this shape: the group's quads fit one 1024-thread workgroup two deep, samples start on
// wave boundaries
int n3d_bwd_small2_ok(int B, int64_t N, int C, int G) {
  if (G < 0) return 0;
  if (!pair_shape_ok(C, G) || B < 1 || B > 4 || N < 1 || N > 4096) return 0;
  const int cpg4 = (C / G) / 4;
  if (cpg4 < 1) return 0;
  const int64_t per_b = (N * cpg4 + 63) / 64 * 64;   // a sample's quads are padded to whole waves
  // two quads per thread at most: four deep (4096 quads) measured slower than the two-launch path (18 vs 16 us)
  return ((int64_t)B * per_b <= 2048) ? 1 : 0;
}

// 0 = not taken; 1 = one workgroup per group (all samples: n3d_bwd_small2_ok); 2 = one workgroup per (group, sample)
int n3d_bwd_small_mode(int B, int64_t N, int C, int G) {
  if (G < 0) return 0;
  if (n3d_bwd_small2_ok(B, N, C, G)) return 1;
  if (!pair_shape_ok(C, G) || B < 2 || B > 4 || N < 1 || N > 4096) return 0;
  const int cpg4 = (C / G) / 4;
  if (cpg4 < 1) return 0;
  return ((N * cpg4 + 63) / 64 * 64 <= 2048) ? 2 : 0;
}

size_t n3d_bwd_small_scratch_bytes(int B, int G) { return (size_t)G * B * 2 * 3 * 16 * sizeof(double); }

static int bwd_small_launch(const float* dout, int64_t dld, const float* dout1, int64_t dld1, const n3d_gn_bwd_term* t0,
                            const n3d_gn_bwd_term* t1, int B, int64_t N, int C, int G, void* scratch, size_t scratch_bytes, uint32_t* tickets,
                            void* stream, const char* what) {
  N3D_CHECK_ARG(dout && t0 && B > 0 && N > 0 && (!t1 || t0->dtype == t1->dtype), "affine_act_bwd_small: bad args");
  N3D_CHECK_ARG(t1 || !dout1, "affine_act_bwd_small: dout1 needs a second term");
  const bool bf = t0->dtype == N3D_BF16;
  if (dout1) { if (int e = check_vec(dout1, dld1, C, "bwd_small(dout1)", bf)) return e; }
  const int mode = n3d_bwd_small_mode(B, N, C, G);
  if (!mode || (mode == 2 && !(scratch && tickets)))
    N3D_UNSUPPORTED("%s: shape not supported (B=%d N=%lld C=%d G=%d%s)", what, B, (long long)N, C, G, mode == 2 ? "; needs scratch and tickets" : "");
  if (mode == 2) N3D_CHECK_ARG(scratch_bytes >= n3d_bwd_small_scratch_bytes(B, G), "affine_act_bwd_small: scratch too small");
  if (int e = check_vec(dout, dld, C, "bwd_small(dout)", bf)) return e;
  GnBwdTerm k[2];
  const n3d_gn_bwd_term* ts[2] = {t0, t1 ? t1 : t0};
  for (int i = 0; i < 2; ++i) {
    const n3d_gn_bwd_term* t = ts[i];
    N3D_CHECK_ARG(t->raw && t->a && t->b && t->gamma && t->mean_rstd && t->draw, "affine_act_bwd_small: null term pointer");
    N3D_CHECK_ARG(!t->dalpha, "affine_act_bwd_small: MixedOp weight gradients are not produced here (use the reduce2 / apply_gn2 pair)");
    N3D_CHECK_ARG(!t->dbias_conv || t->sumraw, "affine_act_bwd_small: dbias_conv needs the forward per-channel sums");
    if (int e = check_vec(t->raw, t->rld, C, "bwd_small(raw)", bf)) return e;
    if (int e = check_vec(t->draw, t->drld, C, "bwd_small(draw)", bf)) return e;
    k[i] = GnBwdTerm{t->raw, t->rld, t->a, t->b, nullptr, 0, t->gamma, t->mean_rstd, t->wptr, t->sumraw, t->draw, t->drld,
                     t->dgamma, t->dbeta, nullptr, t->dbias_conv, t->relu, nullptr, nullptr, nullptr};
  }
  const int64_t per_b = (N * ((C / G) / 4) + 63) / 64 * 64;
  const int64_t quads = mode == 2 ? per_b : (int64_t)B * per_b;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(G, mode == 2 ? B : 1);
  double* sc = (double*)scratch;
  with_act_type(bf, [&](auto* tag) {
    using T = N3D_T(tag);
    const T* d0 = (const T*)dout; const T* d1 = (const T*)dout1;
#define N3D_BS(Q, TW, SP, NT_) hipLaunchKernelGGL((gn_bwd_small2_kernel<Q, TW, SP, NT_, T>), grid, dim3(1024), 0, s, d0, dld, d1, dld1, k[0], k[1], B, (int)N, C, G, (double)N, sc, tickets)
#define N3D_BS_Q(TW, SP, NT_) do { if (quads <= 1024) N3D_BS(1, TW, SP, NT_); else N3D_BS(2, TW, SP, NT_); } while (0)
    if (!t1) { if (mode == 2) N3D_BS_Q(false, true, 1); else N3D_BS_Q(false, false, 1); }
    else if (dout1) { if (mode == 2) N3D_BS_Q(true, true, 2); else N3D_BS_Q(true, false, 2); }
    else { if (mode == 2) N3D_BS_Q(false, true, 2); else N3D_BS_Q(false, false, 2); }
#undef N3D_BS_Q
#undef N3D_BS
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act_bwd_small2(const float* dout, int64_t dld, const float* dout1, int64_t dld1, const n3d_gn_bwd_term* t0,
                              const n3d_gn_bwd_term* t1, int B, int64_t N, int C, int G, void* stream) {
  if (G < 0) N3D_UNSUPPORTED("%s: padded-channel GroupNorm (G < 0) is not taken by the one-launch backward", "n3d_affine_act_bwd_small2");
  N3D_CHECK_ARG(t1, "affine_act_bwd_small2: two terms");
  if (!n3d_bwd_small2_ok(B, N, C, G)) N3D_UNSUPPORTED("affine_act_bwd_small2: shape not supported (B=%d N=%lld C=%d G=%d)", B, (long long)N, C, G);
  return bwd_small_launch(dout, dld, dout1, dld1, t0, t1, B, N, C, G, nullptr, 0, nullptr, stream, "affine_act_bwd_small2");
}

int n3d_affine_act_bwd_small(const float* dout, int64_t dld, const float* dout1, int64_t dld1, const n3d_gn_bwd_term* t0,
                             const n3d_gn_bwd_term* t1, int B, int64_t N, int C, int G, void* scratch, size_t scratch_bytes,
                             uint32_t* tickets, void* stream) {
  if (G < 0) N3D_UNSUPPORTED("%s: padded-channel GroupNorm (G < 0) is not taken by the one-launch backward", "n3d_affine_act_bwd_small");
  return bwd_small_launch(dout, dld, dout1, dld1, t0, t1, B, N, C, G, scratch, scratch_bytes, tickets, stream, "affine_act_bwd_small");
}

int n3d_gn_coeffs2(const n3d_gn_fwd_term* t0, const n3d_gn_fwd_term* t1, int B, int C, int G, int64_t N, float eps, void* stream) {
  const double gn_cnt_ = gn_count(N, C, G);      // G < 0: ONE group of -G real channels inside C zero-padded ones (include/n3d.h)
  G = gn_groups(G);
  N3D_CHECK_ARG(t0 && t1 && C <= 64 && G >= 1 && C % G == 0, "gn_coeffs2: bad args");
  GnCoefArgs q[2];
  const n3d_gn_fwd_term* ts[2] = {t0, t1};
  for (int i = 0; i < 2; ++i) {
    const n3d_gn_fwd_term* t = ts[i];
    N3D_CHECK_ARG(t->stats && t->gamma && t->beta && t->a_out && t->b_out && t->rows >= 1, "gn_coeffs2: null term pointer");
    q[i] = GnCoefArgs{t->stats, t->rows, t->gamma, t->beta, t->a_out, t->b_out, t->mean_rstd_out, t->sumraw};
  }
  hipLaunchKernelGGL(gn_coeffs_kernel, dim3(B, 2), dim3(256), 0, (hipStream_t)stream, q[0], q[1], C, G, gn_cnt_, eps);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act2(const n3d_gn_fwd_term* t0, const n3d_gn_fwd_term* t1, float* out, int64_t old_, float* out1, int64_t old1, int B,
                    int64_t N, int C, int flags, void* stream) {
  N3D_CHECK_ARG(t0 && t1 && out && B > 0 && N > 0 && C <= 64, "affine_act2: bad args");
  GnFwdTerm k[2];
  const n3d_gn_fwd_term* ts[2] = {t0, t1};
  for (int i = 0; i < 2; ++i) {
    const n3d_gn_fwd_term* t = ts[i];
    N3D_CHECK_ARG(t->raw && t->a_out && t->b_out, "affine_act2: null term pointer");
    if (int e = check_vec(t->raw, t->rld, C, "affine_act2(raw)", flags & N3D_ACT_BF16)) return e;
    k[i] = GnFwdTerm{t->raw, t->rld, nullptr, 0, nullptr, nullptr, t->wptr, t->a_out, t->b_out, nullptr, nullptr, t->relu};
  }
  const bool bf = flags & N3D_ACT_BF16;
  if (int e = check_vec(out, old_, C, "affine_act2(out)", bf)) return e;
  if (out1) { if (int e = check_vec(out1, old1, C, "affine_act2(out1)", bf)) return e; }
  EwMap m = ew_map(N, C);
  dim3 grid(m.rows, B), blk(256);
  hipStream_t s = (hipStream_t)stream;
  with_act_type(bf, [&](auto* tag) {
    using T = N3D_T(tag);
    if (flags & N3D_ACCUMULATE) hipLaunchKernelGGL((affine_act_gn2_kernel<true, true, T>), grid, blk, 0, s, k[0], k[1], 1, (double)N, 0.f, (T*)out, old_, (T*)out1, old1, N, C, m);
    else hipLaunchKernelGGL((affine_act_gn2_kernel<false, true, T>), grid, blk, 0, s, k[0], k[1], 1, (double)N, 0.f, (T*)out, old_, (T*)out1, old1, N, C, m);
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_gn_bwd_coeffs2(const n3d_gn_bwd_term* t0, const n3d_gn_bwd_term* t1, int B, int C, int G, int64_t N, void* stream) {
  const double gn_cnt_ = gn_count(N, C, G);      // G < 0: ONE group of -G real channels inside C zero-padded ones (include/n3d.h)
  G = gn_groups(G);
  N3D_CHECK_ARG(t0 && t1 && C <= 64 && G >= 1 && C % G == 0, "gn_bwd_coeffs2: bad args");
  GnBwdCoefArgs q[2];
  const n3d_gn_bwd_term* ts[2] = {t0, t1};
  for (int i = 0; i < 2; ++i) {
    const n3d_gn_bwd_term* t = ts[i];
    N3D_CHECK_ARG(t->sums && t->gamma && t->mean_rstd && t->cA && t->cB && t->cC && t->rows >= 1, "gn_bwd_coeffs2: null term pointer");
    N3D_CHECK_ARG(!t->dbias_conv || t->sumraw, "gn_bwd_coeffs2: dbias_conv needs the forward per-channel sums");
    q[i] = GnBwdCoefArgs{t->sums, t->rows, t->gamma, t->mean_rstd, t->wptr, t->dgamma, t->dbeta, t->dalpha, t->cA, t->cB, t->cC, t->sumraw,
                         t->dbias_conv};
  }
  if (B <= 2) hipLaunchKernelGGL(gn_bwd_coeffs_kernel<2>, dim3(2), dim3(512), 0, (hipStream_t)stream, q[0], q[1], B, C, G, gn_cnt_);
  else hipLaunchKernelGGL(gn_bwd_coeffs_kernel<GNB_BP>, dim3(2), dim3(256 * GNB_BP), 0, (hipStream_t)stream, q[0], q[1], B, C, G, gn_cnt_);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act_bwd_apply2(const float* dout, int64_t dld, const float* dout1, int64_t dld1, const n3d_gn_bwd_term* t0,
                              const n3d_gn_bwd_term* t1, int B, int64_t N, int C, void* stream) {
  N3D_CHECK_ARG(dout && t0 && t1 && B > 0 && N > 0 && C <= 64 && t0->dtype == t1->dtype, "affine_act_bwd_apply2: bad args");
  const bool bf = t0->dtype == N3D_BF16;
  if (int e = check_vec(dout, dld, C, "bwd_apply2(dout)", bf)) return e;
  GnBwdTerm k[2];
  const n3d_gn_bwd_term* ts[2] = {t0, t1};
  for (int i = 0; i < 2; ++i) {
    const n3d_gn_bwd_term* t = ts[i];
    N3D_CHECK_ARG(t->raw && t->a && t->b && t->cA && t->cB && t->cC && t->draw, "affine_act_bwd_apply2: null term pointer");
    if (int e = check_vec(t->raw, t->rld, C, "bwd_apply2(raw)", bf)) return e;
    if (int e = check_vec(t->draw, t->drld, C, "bwd_apply2(draw)", bf)) return e;
    k[i] = GnBwdTerm{t->raw, t->rld, t->a, t->b, nullptr, 0, nullptr, nullptr, nullptr, nullptr, t->draw, t->drld, nullptr, nullptr, nullptr,
                     nullptr, t->relu, t->cA, t->cB, t->cC};
  }
  EwMap m = ew_map(N, C);
  if (dout1) { if (int e = check_vec(dout1, dld1, C, "bwd_apply2(dout1)", bf)) return e; }
  with_act_type(bf, [&](auto* tag) {
    using T = N3D_T(tag);
    if (dout1) hipLaunchKernelGGL((affine_bwd_apply_gn2_kernel<true, true, T>), dim3(m.rows, B), dim3(256), 0, (hipStream_t)stream, (const T*)dout, dld, (const T*)dout1, dld1, k[0], k[1], B, 1, (double)N, N, C, m);
    else hipLaunchKernelGGL((affine_bwd_apply_gn2_kernel<true, false, T>), dim3(m.rows, B), dim3(256), 0, (hipStream_t)stream, (const T*)dout, dld, (const T*)dout1, dld1, k[0], k[1], B, 1, (double)N, N, C, m);
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

// ---- N-term forms (supernet nodes) -----------------------------------------------------------------
static int check_group(int n, int C, const char* who) {
  if (n < 1 || n > N3D_MAX_GROUP_TERMS) { set_error("%s: 1..%d terms, got %d", who, N3D_MAX_GROUP_TERMS, n); return N3D_ERR_INVALID; }
  if (C < 4 || C > 64 || (C & (C - 1)) != 0) { set_error("%s: C must be a power of two in 4..64 (C=%d)", who, C); return N3D_ERR_UNSUPPORTED; }
  return N3D_OK;
}

int n3d_gn_coeffsN(const n3d_gn_fwd_term* terms, int n, int B, int C, int G, int64_t N, float eps, void* stream) {
  const double gn_cnt_ = gn_count(N, C, G);      // G < 0: ONE group of -G real channels inside C zero-padded ones (include/n3d.h)
  G = gn_groups(G);
  N3D_CHECK_ARG(terms && B > 0 && N > 0 && G >= 1 && C % G == 0, "gn_coeffsN: bad args");
  if (int e = check_group(n, C, "gn_coeffsN")) return e;
  GnCoefArgsN qs;
  for (int i = 0; i < 8; ++i) {
    const n3d_gn_fwd_term* t = &terms[i < n ? i : 0];
    N3D_CHECK_ARG(t->stats && t->gamma && t->beta && t->a_out && t->b_out && t->rows >= 1, "gn_coeffsN: null term pointer");
    qs.q[i] = GnCoefArgs{t->stats, t->rows, t->gamma, t->beta, t->a_out, t->b_out, t->mean_rstd_out, t->sumraw};
  }
  hipLaunchKernelGGL(gn_coeffsN_kernel, dim3(B, n), dim3(256), 0, (hipStream_t)stream, qs, C, G, gn_cnt_, eps);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_actN(const n3d_gn_fwd_term* terms, int n, float* out, int64_t old_, int B, int64_t N, int C, int flags, void* stream) {
  N3D_CHECK_ARG(terms && out && B > 0 && N > 0, "affine_actN: bad args");
  if (int e = check_group(n, C, "affine_actN")) return e;
  FwdTermN ts;
  ts.n = n;
  for (int i = 0; i < 8; ++i) {
    const n3d_gn_fwd_term* t = &terms[i < n ? i : 0];
    N3D_CHECK_ARG(t->raw, "affine_actN: null term pointer");  // a_out / b_out may be NULL: scale 1 / shift 0
    if (int e = check_vec(t->raw, t->rld, C, "affine_actN(raw)")) return e;
    ts.raw[i] = t->raw; ts.rld[i] = t->rld; ts.a[i] = t->a_out; ts.b[i] = t->b_out; ts.wptr[i] = t->wptr; ts.relu[i] = t->relu;
  }
  if (int e = check_vec(out, old_, C, "affine_actN(out)")) return e;
  EwMap m = ew_map(N, C);
  dim3 grid(m.rows, B), blk(256);
  if (flags & N3D_ACCUMULATE) hipLaunchKernelGGL(affine_actN_kernel<true>, grid, blk, 0, (hipStream_t)stream, ts, out, old_, N, C, m);
  else hipLaunchKernelGGL(affine_actN_kernel<false>, grid, blk, 0, (hipStream_t)stream, ts, out, old_, N, C, m);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act_bwd_reduceN(const float* dout, int64_t dld, const n3d_gn_bwd_term* terms, int n, int B, int64_t N, int C, void* stream) {
  N3D_CHECK_ARG(dout && terms && B > 0 && N > 0, "affine_act_bwd_reduceN: bad args");
  N3D_CHECK_ARG(n >= 1 && n <= N3D_MAX_REDUCE_TERMS, "affine_act_bwd_reduceN: 1..16 terms");
  if (int e = check_group(1, C, "affine_act_bwd_reduceN")) return e;
  if (int e = check_vec(dout, dld, C, "bwd_reduceN(dout)")) return e;
  BwdRedTermN ts;
  for (int i = 0; i < N3D_MAX_REDUCE_TERMS; ++i) {
    const n3d_gn_bwd_term* t = &terms[i < n ? i : 0];
    N3D_CHECK_ARG(t->raw && t->sums, "affine_act_bwd_reduceN: null term pointer");  // a / b may be NULL: scale 1 / shift 0
    if (int e = check_vec(t->raw, t->rld, C, "bwd_reduceN(raw)")) return e;
    ts.t[i] = BwdRedTerm{t->raw, t->rld, t->a, t->b, t->sums, t->relu};
  }
  EwMap m = ew_map(N, C);
  hipLaunchKernelGGL(affine_bwd_reduceN_kernel, dim3(m.rows, B, n), dim3(256), 0, (hipStream_t)stream, dout, dld, ts, N, C, m);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_gn_bwd_coeffsN(const n3d_gn_bwd_term* terms, int n, int B, int C, int G, int64_t N, void* stream) {
  const double gn_cnt_ = gn_count(N, C, G);      // G < 0: ONE group of -G real channels inside C zero-padded ones (include/n3d.h)
  G = gn_groups(G);
  N3D_CHECK_ARG(terms && B > 0 && N > 0 && G >= 1 && C % G == 0, "gn_bwd_coeffsN: bad args");
  if (int e = check_group(n, C, "gn_bwd_coeffsN")) return e;
  GnBwdCoefArgsN qs;
  for (int i = 0; i < 8; ++i) {
    const n3d_gn_bwd_term* t = &terms[i < n ? i : 0];
    N3D_CHECK_ARG(t->sums && t->gamma && t->mean_rstd && t->cA && t->cB && t->cC && t->rows >= 1, "gn_bwd_coeffsN: null term pointer");
    N3D_CHECK_ARG(!t->dbias_conv || t->sumraw, "gn_bwd_coeffsN: dbias_conv needs the forward per-channel sums");
    qs.q[i] = GnBwdCoefArgs{t->sums, t->rows, t->gamma, t->mean_rstd, t->wptr, t->dgamma, t->dbeta, t->dalpha, t->cA, t->cB, t->cC, t->sumraw,
                            t->dbias_conv};
  }
  if (B <= 2) hipLaunchKernelGGL(gn_bwd_coeffsN_kernel<2>, dim3(n), dim3(512), 0, (hipStream_t)stream, qs, B, C, G, gn_cnt_);
  else hipLaunchKernelGGL(gn_bwd_coeffsN_kernel<GNB_BP>, dim3(n), dim3(256 * GNB_BP), 0, (hipStream_t)stream, qs, B, C, G, gn_cnt_);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act_bwd_applyN(const float* dout, int64_t dld, const n3d_gn_bwd_term* terms, int n, int B, int64_t N, int C, void* stream) {
  N3D_CHECK_ARG(dout && terms && B > 0 && N > 0 && n >= 1 && n <= N3D_MAX_REDUCE_TERMS, "affine_act_bwd_applyN: bad args (1..16 terms)");
  if (int e = check_group(1, C, "affine_act_bwd_applyN")) return e;
  if (int e = check_vec(dout, dld, C, "bwd_applyN(dout)")) return e;
  BwdApplyTermN ts;
  for (int i = 0; i < N3D_MAX_REDUCE_TERMS; ++i) {
    const n3d_gn_bwd_term* t = &terms[i < n ? i : 0];
    N3D_CHECK_ARG(t->raw && t->a && t->b && t->cA && t->cB && t->cC && t->draw, "affine_act_bwd_applyN: null term pointer");
    if (int e = check_vec(t->raw, t->rld, C, "bwd_applyN(raw)")) return e;
    if (int e = check_vec(t->draw, t->drld, C, "bwd_applyN(draw)")) return e;
    ts.t[i] = BwdApplyTerm{t->raw, t->rld, t->a, t->b, t->cA, t->cB, t->cC, t->draw, t->drld, t->relu};
  }
  EwMap m = ew_map(N, C);
  hipLaunchKernelGGL(affine_bwd_applyN_kernel, dim3(m.rows, B, n), dim3(256), 0, (hipStream_t)stream, dout, dld, ts, N, C, m);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act_bwd_apply_sum(const float* dout, int64_t dld, const n3d_gn_bwd_term* terms, int n, int B, int64_t N, int C, void* stream) {
  N3D_CHECK_ARG(dout && terms && n >= 1 && n <= N3D_MAX_REDUCE_TERMS && B > 0 && N > 0, "affine_act_bwd_apply_sum: 1..16 terms");
  if (int e = check_group(1, C, "affine_act_bwd_apply_sum")) return e;
  if (int e = check_vec(dout, dld, C, "bwd_apply_sum(dout)")) return e;
  ApplySumArgs ts;
  int nt = 0;
  for (int i = 0; i < 8; ++i) ts.start[i] = ts.count[i] = ts.acc[i] = 0;
  for (int i = 0; i < N3D_MAX_REDUCE_TERMS; ++i) {
    const n3d_gn_bwd_term* t = &terms[i < n ? i : 0];
    N3D_CHECK_ARG(t->raw && t->draw, "affine_act_bwd_apply_sum: null term pointer");   // a / b / cA / cB / cC may be NULL: 1 / 0 / 1 / 0 / 0
    if (int e = check_vec(t->raw, t->rld, C, "bwd_apply_sum(raw)")) return e;
    if (int e = check_vec(t->draw, t->drld, C, "bwd_apply_sum(target)")) return e;
    ts.t[i] = BwdApplyTerm{t->raw, t->rld, t->a, t->b, t->cA, t->cB, t->cC, t->draw, t->drld, t->relu};
    if (i < n) {
      if (i > 0 && terms[i - 1].draw == t->draw) {
        N3D_CHECK_ARG(ts.count[nt - 1] < 4 && terms[i - 1].drld == t->drld, "affine_act_bwd_apply_sum: at most 4 terms per target, one pitch");
        ++ts.count[nt - 1];
      } else {
        N3D_CHECK_ARG(nt < 8, "affine_act_bwd_apply_sum: at most 8 targets");
        ts.start[nt] = i; ts.count[nt] = 1; ts.acc[nt] = t->pad_ & 1;
        ++nt;
      }
    }
  }
  for (int i = 0; i < nt; ++i)
    for (int j = 0; j < i; ++j)
      N3D_CHECK_ARG(terms[ts.start[i]].draw != terms[ts.start[j]].draw, "affine_act_bwd_apply_sum: the terms of a target must be consecutive");
  EwMap m = ew_map(N, C);
  hipLaunchKernelGGL(affine_bwd_apply_sum_kernel, dim3(m.rows, B, nt), dim3(256), 0, (hipStream_t)stream, dout, dld, ts, N, C, m);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_plain_bwd_coeffs(const double* sums, int rows, const float* wptr, int B, int C, float* dalpha, float* A, void* stream) {
  N3D_CHECK_ARG(C <= 64 && (A || dalpha), "plain_bwd_coeffs: bad args");
  N3D_CHECK_ARG(!dalpha || sums, "plain_bwd_coeffs: dalpha needs sums");
  hipLaunchKernelGGL(plain_bwd_coeffs_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, rows, wptr, B, C, dalpha, A);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_plain_bwd_coeffsN(const n3d_plain_coef_term* terms, int n, int B, int C, void* stream) {
  N3D_CHECK_ARG(terms && n >= 1 && n <= N3D_MAX_GROUP_TERMS && B > 0 && C >= 1 && C <= 64, "plain_bwd_coeffsN: bad args");
  PlainCoefArgsN qs;
  for (int i = 0; i < 8; ++i) {
    const n3d_plain_coef_term* t = &terms[i < n ? i : 0];
    N3D_CHECK_ARG((!t->dalpha || (t->sums && t->rows >= 1)) && (t->dalpha || t->A), "plain_bwd_coeffsN: null term pointer");
    qs.q[i] = PlainCoefArgs{t->sums, t->rows, t->wptr, t->dalpha, t->A};
  }
  hipLaunchKernelGGL(plain_bwd_coeffsN_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, qs, B, C);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act_bwd_apply(const float* dout, int64_t dld, const float* raw, int64_t rld, const float* a, const float* b, const float* A,
                             const float* Bc, const float* Cc, float* draw, int64_t drld, int B, int64_t N, int C, int flags,
                             void* stream) {
  N3D_CHECK_ARG(dout && raw && draw, "affine_act_bwd_apply: bad args");
  const bool bf = flags & N3D_ACT_BF16;
  if (int e = check_vec(dout, dld, C, "bwd_apply(dout)", bf)) return e;
  if (int e = check_vec(raw, rld, C, "bwd_apply(raw)", bf)) return e;
  if (int e = check_vec(draw, drld, C, "bwd_apply(draw)", bf)) return e;
  EwMap m = ew_map(N, C);
  dim3 grid(m.rows, B), blk(256);
  hipStream_t s = (hipStream_t)stream;
  const bool relu = flags & N3D_RELU, acc = flags & N3D_ACCUMULATE;
  with_act_type(bf, [&](auto* tag) {
    using T = N3D_T(tag);
    const T* d = (const T*)dout; const T* r = (const T*)raw; T* o = (T*)draw;
    if (relu && acc) hipLaunchKernelGGL((affine_bwd_apply_kernel<true, true, T>), grid, blk, 0, s, d, dld, r, rld, a, b, A, Bc, Cc, o, drld, N, C, m);
    else if (relu) hipLaunchKernelGGL((affine_bwd_apply_kernel<true, false, T>), grid, blk, 0, s, d, dld, r, rld, a, b, A, Bc, Cc, o, drld, N, C, m);
    else if (acc) hipLaunchKernelGGL((affine_bwd_apply_kernel<false, true, T>), grid, blk, 0, s, d, dld, r, rld, a, b, A, Bc, Cc, o, drld, N, C, m);
    else hipLaunchKernelGGL((affine_bwd_apply_kernel<false, false, T>), grid, blk, 0, s, d, dld, r, rld, a, b, A, Bc, Cc, o, drld, N, C, m);
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_se_gate_fwd(const double* stats, int rows, int64_t N, const float* w1, const float* b1, const float* w2, const float* b2, int B,
                    int C, float* mean, float* hidden, float* gate, void* stream) {
  N3D_CHECK_ARG(stats && w1 && b1 && w2 && b2 && mean && hidden && gate && C <= 64, "se_gate_fwd: bad args");
  hipLaunchKernelGGL(se_gate_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, stats, rows, (double)N, w1, b1, w2, b2, C, mean, hidden, gate);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_se_gate_bwd(const double* sums, int rows, const float* wptr, const float* mean, const float* hidden, const float* gate,
                    const float* w1, const float* w2, int B, int C, int64_t N, float* dw1, float* db1, float* dw2, float* db2,
                    float* dalpha, float* A, float* Bc, void* stream) {
  N3D_CHECK_ARG(sums && mean && hidden && gate && w1 && w2 && dw1 && db1 && dw2 && db2 && A && Bc && C <= 64, "se_gate_bwd: bad args");
  if (B == 2) {
    const SeTerm q{sums, rows, w1, nullptr, w2, nullptr, const_cast<float*>(mean), const_cast<float*>(hidden), const_cast<float*>(gate), wptr, dw1, db1, dw2,
                   db2, dalpha, A, Bc};
    hipLaunchKernelGGL(se_gate_bwd2_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, q, C, (double)N);
  } else {
    hipLaunchKernelGGL(se_gate_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sums, rows, wptr, mean, hidden, gate, w1, w2, B, C,
                       (double)N, dw1, db1, dw2, db2, dalpha, A, Bc);
  }
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

static int se_terms(const n3d_se_term* terms, int n, bool bwd, SeTermN* out, const char* who) {
  if (n < 1 || n > N3D_MAX_GROUP_TERMS) { set_error("%s: 1..%d gates, got %d", who, N3D_MAX_GROUP_TERMS, n); return N3D_ERR_INVALID; }
  for (int i = 0; i < 8; ++i) {
    const n3d_se_term* t = &terms[i < n ? i : 0];
    if (!(t->sums && t->rows >= 1 && t->w1 && t->w2 && t->mean && t->hidden && t->gate) || (!bwd && !(t->b1 && t->b2)) ||
        (bwd && !(t->dw1 && t->db1 && t->dw2 && t->db2 && t->A && t->Bc))) {
      set_error("%s: null pointer in gate %d", who, i);
      return N3D_ERR_INVALID;
    }
    out->t[i] = SeTerm{t->sums, t->rows, t->w1, t->b1, t->w2, t->b2, t->mean, t->hidden, t->gate, t->wptr, t->dw1, t->db1, t->dw2, t->db2,
                       t->dalpha, t->A, t->Bc};
  }
  return N3D_OK;
}

int n3d_se_gate_fwdN(const n3d_se_term* terms, int n, int64_t N, int B, int C, void* stream) {
  N3D_CHECK_ARG(terms && B > 0 && N > 0 && C >= 1 && C <= 64, "se_gate_fwdN: bad args");
  SeTermN ts;
  if (int e = se_terms(terms, n, false, &ts, "se_gate_fwdN")) return e;
  hipLaunchKernelGGL(se_gate_fwdN_kernel, dim3(B, n), dim3(256), 0, (hipStream_t)stream, ts, (double)N, C);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_se_gate_bwdN(const n3d_se_term* terms, int n, int64_t N, int B, int C, void* stream) {
  N3D_CHECK_ARG(terms && B > 0 && N > 0 && C >= 1 && C <= 64, "se_gate_bwdN: bad args");
  SeTermN ts;
  if (int e = se_terms(terms, n, true, &ts, "se_gate_bwdN")) return e;
  if (B == 2) hipLaunchKernelGGL(se_gate_bwdN2_kernel, dim3(n), dim3(512), 0, (hipStream_t)stream, ts, C, (double)N);
  else hipLaunchKernelGGL(se_gate_bwdN_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, ts, B, C, (double)N);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_node_fwd_coeffs(const n3d_gn_fwd_term* gn, int n_gn, const n3d_se_term* se, int n_se, int B, int C, int G, int64_t N, float eps, void* stream) {
  if (G < 0) N3D_UNSUPPORTED("%s: padded-channel GroupNorm (G < 0) is not taken by the node-level launches (their SE gates share the count)", "n3d_node_fwd_coeffs");
  N3D_CHECK_ARG(gn && se && n_gn >= 1 && n_gn <= N3D_MAX_GROUP_TERMS && n_se >= 1 && n_se <= N3D_MAX_GROUP_TERMS && B > 0 && N > 0 && G >= 1 &&
                C % G == 0 && C <= 64, "node_fwd_coeffs: 1..8 GroupNorm terms and 1..8 SE gates");
  if (int e = check_group(n_gn, C, "node_fwd_coeffs")) return e;
  NodeFwdCoefArgs qs;
  qs.n_gn = n_gn;
  for (int i = 0; i < 8; ++i) {
    const n3d_gn_fwd_term* t = &gn[i < n_gn ? i : 0];
    N3D_CHECK_ARG(t->stats && t->gamma && t->beta && t->a_out && t->b_out && t->rows >= 1, "node_fwd_coeffs: null GroupNorm term pointer");
    qs.g[i] = GnCoefArgs{t->stats, t->rows, t->gamma, t->beta, t->a_out, t->b_out, t->mean_rstd_out, t->sumraw};
  }
  SeTermN ts;
  if (int e = se_terms(se, n_se, false, &ts, "node_fwd_coeffs")) return e;
  for (int i = 0; i < 8; ++i) qs.s[i] = ts.t[i];
  hipLaunchKernelGGL(node_fwd_coeffs_kernel, dim3(B, n_gn + n_se), dim3(256), 0, (hipStream_t)stream, qs, C, G, (double)N, eps);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_node_bwd_coeffs(const n3d_gn_bwd_term* gn, int n_gn, const n3d_se_term* se, int n_se, int B, int C, int G, int64_t N, void* stream) {
  if (G < 0) N3D_UNSUPPORTED("%s: padded-channel GroupNorm (G < 0) is not taken by the node-level launches (their SE gates share the count)", "n3d_node_bwd_coeffs");
  N3D_CHECK_ARG(n_gn >= 0 && n_gn <= N3D_MAX_REDUCE_TERMS && n_se >= 0 && n_se <= N3D_MAX_GROUP_TERMS && n_gn + n_se >= 1 && (n_gn == 0 || gn) &&
                (n_se == 0 || se), "node_bwd_coeffs: 0..16 GroupNorm terms, 0..8 SE gates");
  if (B == 2 && n_gn >= 1 && n_se >= 1) {
    N3D_CHECK_ARG(B > 0 && N > 0 && G >= 1 && C % G == 0 && C >= 1 && C <= 64, "node_bwd_coeffs: bad args");
    if (int e = check_group(1, C, "node_bwd_coeffs")) return e;
    NodeCoefArgs qs;
    qs.n_gn = n_gn;
    for (int i = 0; i < N3D_MAX_REDUCE_TERMS; ++i) {
      const n3d_gn_bwd_term* t = &gn[i < n_gn ? i : 0];
      N3D_CHECK_ARG(t->sums && t->gamma && t->mean_rstd && t->cA && t->cB && t->cC && t->rows >= 1, "node_bwd_coeffs: null GroupNorm term pointer");
      N3D_CHECK_ARG(!t->dbias_conv || t->sumraw, "node_bwd_coeffs: dbias_conv needs the forward per-channel sums");
      qs.g[i] = GnBwdCoefArgs{t->sums, t->rows, t->gamma, t->mean_rstd, t->wptr, t->dgamma, t->dbeta, t->dalpha, t->cA, t->cB, t->cC, t->sumraw,
                              t->dbias_conv};
    }
    SeTermN ts;
    if (int e = se_terms(se, n_se, true, &ts, "node_bwd_coeffs")) return e;
    for (int i = 0; i < 8; ++i) qs.s[i] = ts.t[i];
    hipLaunchKernelGGL(node_bwd_coeffs_kernel, dim3(n_gn + n_se), dim3(512), 0, (hipStream_t)stream, qs, C, G, (double)N);
    N3D_LAUNCH_CHECK();
    return N3D_OK;
  }
  // other batch sizes / one kind only: the separate launches
  for (int i = 0; i < n_gn; i += N3D_MAX_GROUP_TERMS)
    if (int e = n3d_gn_bwd_coeffsN(gn + i, n_gn - i < N3D_MAX_GROUP_TERMS ? n_gn - i : N3D_MAX_GROUP_TERMS, B, C, G, N, stream)) return e;
  if (n_se >= 1)
    if (int e = n3d_se_gate_bwdN(se, n_se, N, B, C, stream)) return e;
  return N3D_OK;
}

int n3d_pool2_fwd(const float* x, int64_t xld, float* y, int64_t yld, int B, int Di, int Hi, int Wi, int C, int flags, void* stream) {
  N3D_CHECK_ARG(x && y && Di % 2 == 0 && Hi % 2 == 0 && Wi % 2 == 0, "pool2_fwd: spatial dims must be even");
  const bool bf = flags & N3D_ACT_BF16;       // both tensors in bf16 storage (round 5)
  if (int e = check_vec(x, xld, C, "pool2_fwd(x)", bf)) return e;
  if (int e = check_vec(y, yld, C, "pool2_fwd(y)", bf)) return e;
  const int64_t total = (int64_t)(Di / 2) * (Hi / 2) * (Wi / 2) * (C / 4);
  dim3 grid((unsigned)cdiv(total, 256), B), blk(256);
  if (bf) {
    const bf16_t* xb = (const bf16_t*)x; bf16_t* yb = (bf16_t*)y;
    if (flags & N3D_POOL_MAX) hipLaunchKernelGGL((pool2_fwd_kernel<true, bf16_t>), grid, blk, 0, (hipStream_t)stream, xb, xld, yb, yld, Di, Hi, Wi, C);
    else hipLaunchKernelGGL((pool2_fwd_kernel<false, bf16_t>), grid, blk, 0, (hipStream_t)stream, xb, xld, yb, yld, Di, Hi, Wi, C);
  } else if (flags & N3D_POOL_MAX) hipLaunchKernelGGL((pool2_fwd_kernel<true>), grid, blk, 0, (hipStream_t)stream, x, xld, y, yld, Di, Hi, Wi, C);
  else hipLaunchKernelGGL((pool2_fwd_kernel<false>), grid, blk, 0, (hipStream_t)stream, x, xld, y, yld, Di, Hi, Wi, C);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_pool2_fwd_both(const float* x, int64_t xld, float* y_avg, int64_t yald, float* y_max, int64_t ymld, int B, int Di, int Hi, int Wi, int C,
                       void* stream) {
  N3D_CHECK_ARG(x && y_avg && y_max && Di % 2 == 0 && Hi % 2 == 0 && Wi % 2 == 0, "pool2_fwd_both: bad args / odd spatial dims");
  if (int e = check_vec(x, xld, C, "pool2_fwd_both(x)")) return e;
  if (int e = check_vec(y_avg, yald, C, "pool2_fwd_both(y_avg)")) return e;
  if (int e = check_vec(y_max, ymld, C, "pool2_fwd_both(y_max)")) return e;
  const int64_t total = (int64_t)(Di / 2) * (Hi / 2) * (Wi / 2) * (C / 4);
  hipLaunchKernelGGL(pool2_fwd_both_kernel, dim3((unsigned)cdiv(total, 256), B), dim3(256), 0, (hipStream_t)stream, x, xld, y_avg, yald, y_max, ymld,
                     Di, Hi, Wi, C);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_pool2_bwd_both(const float* dy, int64_t dyld, const float* x, int64_t xld, float* dx, int64_t dxld, int B, int Di, int Hi, int Wi, int C,
                       int flags, const float* w_avg, const float* w_max, void* stream) {
  N3D_CHECK_ARG(dy && x && dx && Di % 2 == 0 && Hi % 2 == 0 && Wi % 2 == 0, "pool2_bwd_both: bad args / odd spatial dims");
  if (int e = check_vec(dy, dyld, C, "pool2_bwd_both(dy)")) return e;
  if (int e = check_vec(dx, dxld, C, "pool2_bwd_both(dx)")) return e;
  if (int e = check_vec(x, xld, C, "pool2_bwd_both(x)")) return e;
  const int64_t total = (int64_t)(Di / 2) * (Hi / 2) * (Wi / 2) * (C / 4);
  dim3 grid((unsigned)cdiv(total, 256), B), blk(256);
  if (flags & N3D_ACCUMULATE) hipLaunchKernelGGL(pool2_bwd_both_kernel<true>, grid, blk, 0, (hipStream_t)stream, dy, dyld, x, xld, dx, dxld, Di, Hi, Wi, C, w_avg, w_max);
  else hipLaunchKernelGGL(pool2_bwd_both_kernel<false>, grid, blk, 0, (hipStream_t)stream, dy, dyld, x, xld, dx, dxld, Di, Hi, Wi, C, w_avg, w_max);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_pool2_bwd(const float* dy, int64_t dyld, const float* x, int64_t xld, float* dx, int64_t dxld, int B, int Di, int Hi, int Wi, int C,
                  int flags, void* stream) {
  return n3d_pool2_bwd_scaled(dy, dyld, x, xld, dx, dxld, B, Di, Hi, Wi, C, flags, nullptr, stream);
}

int n3d_pool2_bwd_scaled(const float* dy, int64_t dyld, const float* x, int64_t xld, float* dx, int64_t dxld, int B, int Di, int Hi, int Wi,
                         int C, int flags, const float* wptr, void* stream) {
  const bool mx = flags & N3D_POOL_MAX, acc = flags & N3D_ACCUMULATE, bf = flags & N3D_ACT_BF16;
  N3D_CHECK_ARG(dy && dx && (!mx || x), "pool2_bwd: bad args");
  if (int e = check_vec(dy, dyld, C, "pool2_bwd(dy)", bf)) return e;
  if (int e = check_vec(dx, dxld, C, "pool2_bwd(dx)", bf)) return e;
  if (mx) if (int e = check_vec(x, xld, C, "pool2_bwd(x)", bf)) return e;
  const int64_t total = (int64_t)(Di / 2) * (Hi / 2) * (Wi / 2) * (C / 4);
  dim3 grid((unsigned)cdiv(total, 256), B), blk(256);
  hipStream_t s = (hipStream_t)stream;
  if (bf) {       // all three tensors in bf16 storage (round 5)
    const bf16_t* dyb = (const bf16_t*)dy; const bf16_t* xb = (const bf16_t*)x; bf16_t* dxb = (bf16_t*)dx;
    if (mx && acc) hipLaunchKernelGGL((pool2_bwd_kernel<true, true, bf16_t>), grid, blk, 0, s, dyb, dyld, xb, xld, dxb, dxld, Di, Hi, Wi, C, wptr);
    else if (mx) hipLaunchKernelGGL((pool2_bwd_kernel<true, false, bf16_t>), grid, blk, 0, s, dyb, dyld, xb, xld, dxb, dxld, Di, Hi, Wi, C, wptr);
    else if (acc) hipLaunchKernelGGL((pool2_bwd_kernel<false, true, bf16_t>), grid, blk, 0, s, dyb, dyld, xb, xld, dxb, dxld, Di, Hi, Wi, C, wptr);
    else hipLaunchKernelGGL((pool2_bwd_kernel<false, false, bf16_t>), grid, blk, 0, s, dyb, dyld, xb, xld, dxb, dxld, Di, Hi, Wi, C, wptr);
    N3D_LAUNCH_CHECK();
    return N3D_OK;
  }
  if (mx && acc) hipLaunchKernelGGL((pool2_bwd_kernel<true, true>), grid, blk, 0, s, dy, dyld, x, xld, dx, dxld, Di, Hi, Wi, C, wptr);
  else if (mx) hipLaunchKernelGGL((pool2_bwd_kernel<true, false>), grid, blk, 0, s, dy, dyld, x, xld, dx, dxld, Di, Hi, Wi, C, wptr);
  else if (acc) hipLaunchKernelGGL((pool2_bwd_kernel<false, true>), grid, blk, 0, s, dy, dyld, x, xld, dx, dxld, Di, Hi, Wi, C, wptr);
  else hipLaunchKernelGGL((pool2_bwd_kernel<false, false>), grid, blk, 0, s, dy, dyld, x, xld, dx, dxld, Di, Hi, Wi, C, wptr);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_dice_fwd(const float* p, int64_t psb, int64_t psc, int64_t psv, const float* t, int64_t tsb, int64_t tsc, int64_t tsv, int B, int C,
                 int64_t N, float smooth, double* partial, double* sums, float* loss, void* stream) {
  N3D_CHECK_ARG(p && t && partial && sums && loss && B > 0 && C > 0 && N > 0, "dice_fwd: bad args");
  const int rows = (int)cdiv(N, DICE_CHUNK);
  hipLaunchKernelGGL(dice_reduce_kernel, dim3(rows, C, B), dim3(256), 0, (hipStream_t)stream, p, psb, psc, psv, t, tsb, tsc, tsv, N, partial);
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, rows, B * C, (double)smooth, sums, loss);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_dice_bwd(const float* p, int64_t psb, int64_t psc, int64_t psv, const float* t, int64_t tsb, int64_t tsc, int64_t tsv, int B, int C,
                 int64_t N, float smooth, const double* sums, const float* dloss, float* dp, int64_t dsb, int64_t dsc, int64_t dsv,
                 void* stream) {
  N3D_CHECK_ARG(t && sums && dp, "dice_bwd: bad args");
  const int rows = (int)cdiv(N, DICE_CHUNK);
  hipLaunchKernelGGL(dice_bwd_kernel, dim3(rows, C, B), dim3(256), 0, (hipStream_t)stream, p, psb, psc, psv, t, tsb, tsc, tsv, N, B * C,
                     (double)smooth, sums, dloss, dp, dsb, dsc, dsv);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_ncdhw_to_ndhwc(const float* src, float* dst, int64_t dld, int B, int C, int64_t N, void* stream) {
  N3D_CHECK_ARG(src && dst && dld >= C, "ncdhw_to_ndhwc: bad args");
  hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel, dim3((unsigned)cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, src, dst, dld, C, N);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}
int n3d_ndhwc_to_ncdhw(const float* src, int64_t sld, float* dst, int B, int C, int64_t N, void* stream) {
  N3D_CHECK_ARG(src && dst && sld >= C, "ndhwc_to_ncdhw: bad args");
  hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel, dim3((unsigned)cdiv(N, 256), B), dim3(256), 0, (hipStream_t)stream, src, sld, dst, C, N);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, const float* lr_ptr, float beta1,
                  float beta2, float eps, float weight_decay, float grad_scale, int32_t* step_ptr, int inc_step, void* stream) {
  return n3d_adam_step_guarded(param, grad, exp_avg, exp_avg_sq, n, lr, lr_ptr, beta1, beta2, eps, weight_decay, grad_scale, step_ptr, inc_step,
                               nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}
int n3d_guard_flag(const void* timeouts, const void* acked, float* flag, void* stream) {
  N3D_CHECK_ARG(timeouts && acked && flag, "guard_flag: null pointer");
  hipLaunchKernelGGL(guard_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const unsigned*)timeouts, (const unsigned*)acked, flag);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}
int n3d_adam_step_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, const float* lr_ptr, float beta1,
                          float beta2, float eps, float weight_decay, float grad_scale, int32_t* step_ptr, int inc_step,
                          const void* timeouts, const void* acked, const float* peer_flag, float* loss, void* host_word, void* stream) {
  N3D_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step_ptr && n > 0, "adam_step: bad args");
  N3D_CHECK_ARG((timeouts == nullptr) == (acked == nullptr), "adam_step_guarded: timeouts and acked come together");
  N3D_CHECK_ARG(inc_step != 1 || (!timeouts && !peer_flag), "adam_step_guarded: a guarded update counts its own step (inc_step 0 or 2)");
  const AdamGuard gd{(const unsigned*)timeouts, (const unsigned*)acked, peer_flag, loss, (unsigned*)host_word};
  // elements per workgroup: 8192 = four rounds of two float4 per lane.  Few fat workgroups beat many thin ones here: 1.8 M
  // parameters take 25.9 us at 1024 per workgroup (1771 workgroups), 16.5 at 2048, 12.4 at 4096, 11.1 at 8192 (4.6 TB/s), 13.4 at 16384
  constexpr int adam_epb = 8192;
  int64_t blocks = cdiv(n, adam_epb);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, lr_ptr,
                     beta1, beta2, eps, weight_decay, grad_scale, step_ptr, inc_step == 2 ? 1 : 0, gd);
  if (inc_step == 1) hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_ptr);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

// fused variants: statistics rows -> coefficients in the kernel prologue (rows <= N3D_FUSED_MAX_ROWS)
int n3d_fused_max_rows(void) {
  constexpr int v = 64;
  return v;
}

int n3d_affine_act_gn(const float* raw, int64_t rld, const double* stats, int rows, const float* gamma, const float* beta, int G, float eps,
                      const float* wptr, float* out, int64_t old_, int B, int64_t N, int C, int flags, float* a_out, float* b_out,
                      float* mean_rstd_out, double* sumraw, void* stream) {
  const double gn_cnt_ = gn_count(N, C, G);      // G < 0: ONE group of -G real channels inside C zero-padded ones (include/n3d.h)
  G = gn_groups(G);
  N3D_CHECK_ARG(raw && out && stats && gamma && beta && a_out && b_out && mean_rstd_out && C <= 64 && C % G == 0 && rows >= 1 && rows <= n3d_fused_max_rows(),
                "affine_act_gn: bad args");
  const bool bf = flags & N3D_ACT_BF16;
  if (int e = check_vec(raw, rld, C, "affine_act_gn(raw)", bf)) return e;
  if (int e = check_vec(out, old_, C, "affine_act_gn(out)", bf)) return e;
  EwMap m = ew_map(N, C);
  dim3 grid(m.rows, B), blk(256);
  hipStream_t s = (hipStream_t)stream;
  const bool relu = flags & N3D_RELU, acc = flags & N3D_ACCUMULATE;
  with_act_type(bf, [&](auto* tag) {
    using T = N3D_T(tag);
#define N3D_AAG(R, A_) hipLaunchKernelGGL((affine_act_gn_kernel<R, A_, T>), grid, blk, 0, s, (const T*)raw, rld, stats, rows, gamma, beta, G, gn_cnt_, eps, wptr, (T*)out, old_, N, C, m, a_out, b_out, mean_rstd_out, sumraw)
    if (relu && acc) N3D_AAG(true, true); else if (relu) N3D_AAG(true, false); else if (acc) N3D_AAG(false, true); else N3D_AAG(false, false);
#undef N3D_AAG
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_affine_act_bwd_apply_gn(const float* dout, int64_t dld, const float* raw, int64_t rld, const float* a, const float* b,
                                const double* sums, int rows, const float* gamma, const float* mean_rstd, const float* wptr,
                                const double* sumraw, float* draw, int64_t drld, int B, int64_t N, int C, int G, int flags,
                                float* dgamma, float* dbeta, float* dalpha, float* dbias_conv, void* stream) {
  const double gn_cnt_ = gn_count(N, C, G);      // G < 0: ONE group of -G real channels inside C zero-padded ones (include/n3d.h)
  G = gn_groups(G);
  N3D_CHECK_ARG(dout && raw && draw && sums && gamma && mean_rstd && C <= 64 && C % G == 0 && rows >= 1 && rows <= n3d_fused_max_rows() && B <= GNF_MAXB,
                "affine_act_bwd_apply_gn: bad args (needs rows <= 64, B <= 4)");
  N3D_CHECK_ARG(!dbias_conv || sumraw, "affine_act_bwd_apply_gn: dbias_conv needs the forward per-channel sums");
  const bool bf = flags & N3D_ACT_BF16;
  if (int e = check_vec(dout, dld, C, "bwd_apply_gn(dout)", bf)) return e;
  if (int e = check_vec(raw, rld, C, "bwd_apply_gn(raw)", bf)) return e;
  if (int e = check_vec(draw, drld, C, "bwd_apply_gn(draw)", bf)) return e;
  EwMap m = ew_map(N, C);
  // wave-level prologue (C a power of two, group width <= 16): one extra workgroup column forms the parameter gradients
  const bool wave_path = (C & (C - 1)) == 0 && C / G <= 16;
  dim3 grid(m.rows + (wave_path ? 1 : 0), B), blk(256);
  hipStream_t s = (hipStream_t)stream;
  const bool relu = flags & N3D_RELU, acc = flags & N3D_ACCUMULATE;
  with_act_type(bf, [&](auto* tag) {
    using T = N3D_T(tag);
#define N3D_ABG(R, A_) hipLaunchKernelGGL((affine_bwd_apply_gn_kernel<R, A_, T>), grid, blk, 0, s, (const T*)dout, dld, (const T*)raw, rld, a, b, sums, rows, gamma, mean_rstd, wptr, sumraw, B, G, gn_cnt_, (T*)draw, drld, N, C, m, dgamma, dbeta, dalpha, dbias_conv)
    if (relu && acc) N3D_ABG(true, true); else if (relu) N3D_ABG(true, false); else if (acc) N3D_ABG(false, true); else N3D_ABG(false, false);
#undef N3D_ABG
  });
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

int n3d_zero(void* p, size_t bytes, void* stream) {
  N3D_CHECK_ARG(p || bytes == 0, "zero: null");
  if (bytes == 0) return N3D_OK;
  hipError_t e = hipMemsetAsync(p, 0, bytes, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return N3D_ERR_HIP; }
  return N3D_OK;
}

}  // extern "C"
