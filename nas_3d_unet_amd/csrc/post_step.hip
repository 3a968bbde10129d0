// Step after the hot path: stitching of the per-patch predictions into the brain-wide volume with
// mean blending (patches.py:172-207, used by prediction.py:132-148) and threshold + label fusion (prediction.py:150-170).
// Both are HBM-bound gathers with integer index work: one thread per output voxel, no atomics -- every voxel visits
// the covering patches in list order and sums in fp64, which reproduces the reference's patch-by-patch float64
// accumulation bit for bit.
#include "n3d_common.h"

namespace n3d {

__global__ __launch_bounds__(256) void stitch_kernel(const float* __restrict__ patches, int64_t sb, int64_t sc, int64_t sv, int C, int P,
                                                     const int32_t* __restrict__ corners, int B, int X, int Y, int Z, double* __restrict__ out,
                                                     int FX, int FY, int FZ, int ox, int oy, int oz, FastDiv fZ, FastDiv fYZ) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  const uint32_t N = (uint32_t)X * Y * Z;
  if (v >= N) return;
  uint32_t x, r, y, z;
  fYZ.divmod(v, x, r);
  fZ.divmod(r, y, z);
  double acc[4] = {0, 0, 0, 0};
  int cnt = 0;
  for (int b = 0; b < B; ++b) {
    const int lx = (int)x - corners[b * 3], ly = (int)y - corners[b * 3 + 1], lz = (int)z - corners[b * 3 + 2];  // uniform loads
    if ((unsigned)lx < (unsigned)P && (unsigned)ly < (unsigned)P && (unsigned)lz < (unsigned)P) {
      const float* p = patches + b * sb + (((int64_t)lx * P + ly) * P + lz) * sv;
      for (int c = 0; c < C; ++c) acc[c] += (double)p[c * sc];
      ++cnt;
    }
  }
  const double inv = 1.0 / (double)(cnt > 0 ? cnt : 1);  // uncovered voxels stay 0 (patches.py:203-205)
  const int64_t FN = (int64_t)FX * FY * FZ;
  const int64_t o = ((int64_t)(x + ox) * FY + (y + oy)) * FZ + (z + oz);
  for (int c = 0; c < C; ++c) out[c * FN + o] = acc[c] / (double)(cnt > 0 ? cnt : 1);
  (void)inv;
}

__global__ __launch_bounds__(256) void tumor_labels_kernel(const double* __restrict__ pred, int64_t N, double thr, int inclusive,
                                                           uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const double p0 = pred[i], p1 = pred[N + i], p2 = pred[2 * N + i];
  const bool a = p0 >= thr, b = p1 >= thr, c = p2 >= thr;
  int t;
  if (inclusive) {
    t = c ? 4 : (a ? 1 : (b ? 2 : 0));          // later assignments win: WT(2), then TC(1), then ET(4)
  } else {
    // channels vote; two votes are settled by the larger probability, the earlier channel on equality (np.argmax)
    t = (a && b) ? (p0 >= p1 ? 1 : 2) : (a ? 1 : 0) + (b ? 2 : 0);
    if (c) t = t == 1 ? (p0 >= p2 ? 1 : 4) : (t == 2 ? (p1 >= p2 ? 2 : 4) : t + 4);
  }
  out[i] = (uint8_t)t;
}

}  // namespace n3d

using namespace n3d;

extern "C" int n3d_stitch(const float* patches, int64_t sb, int64_t sc, int64_t sv, int C, int P, const int32_t* corners, int B, int X, int Y, int Z,
                          double* out, int FX, int FY, int FZ, int ox, int oy, int oz, void* stream) {
  N3D_CHECK_ARG(patches && corners && out && C >= 1 && C <= 4 && P > 0 && B >= 1 && X > 0 && Y > 0 && Z > 0, "stitch: bad args (1 <= C <= 4)");
  N3D_CHECK_ARG(ox >= 0 && oy >= 0 && oz >= 0 && ox + X <= FX && oy + Y <= FY && oz + Z <= FZ, "stitch: brain-wide box outside the full image");
  N3D_CHECK_ARG((int64_t)X * Y * Z < (1ll << 31), "stitch: volume too large");
  const uint32_t N = (uint32_t)X * Y * Z;
  hipLaunchKernelGGL(stitch_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, patches, sb, sc, sv, C, P, corners, B, X, Y, Z, out,
                     FX, FY, FZ, ox, oy, oz, FastDiv((uint32_t)Z), FastDiv((uint32_t)Y * Z));
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}

extern "C" int n3d_tumor_labels(const double* pred, int64_t N, double threshold, int inclusive, uint8_t* out, void* stream) {
  N3D_CHECK_ARG(pred && out && N > 0, "tumor_labels: bad args");
  hipLaunchKernelGGL(tumor_labels_kernel, dim3((unsigned)cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, pred, N, threshold, inclusive, out);
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}
