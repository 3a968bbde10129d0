// On-device data step: the per-batch work the reference's generator does on the host just before the
// hot path (train.py:117-119 receives its result): patch crop with zero padding (patches.py:99-115,152-169), one of
// the 48 cube isometries per patch (augment.py:73-131, the same key for data and truth) and the expansion of the
// BraTS label volume into three binary channels (generator.py:230-248).  ONE gather launch produces the NDHWC input
// batch and the target batch straight from the volume resident in HBM: integer index arithmetic, HBM-bound.
#include "n3d_common.h"

namespace n3d {

struct PatchDescs { n3d_patch_desc d[N3D_PATCH_MAX_BATCH]; };

// one thread = one output voxel (b, i0, i1, i2); writes are lane-consecutive along i2 (x: Cv floats per voxel)
// TT: storage of the three target maps -- float, or uint8_t (N3D_PATCH_T_U8: the generator's booleans as bytes, what n3d_head_fwd /
// n3d_head_bwd read with t_dtype = N3D_U8)
template <typename TT>
__global__ __launch_bounds__(256) void patch_batch_kernel(const float* __restrict__ vol, int Cv, const uint8_t* __restrict__ truth, int X, int Y, int Z,
                                                          PatchDescs descs, int P, int inclusive, float* __restrict__ x_out, int64_t xld,
                                                          TT* __restrict__ t_out, FastDiv fP, FastDiv fPP) {
  const int b = blockIdx.y;
  const n3d_patch_desc d = descs.d[b];
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  const uint32_t P3 = (uint32_t)P * P * P;
  if (v >= P3) return;
  uint32_t i0, r, i1, i2;
  fPP.divmod(v, i0, r);
  fP.divmod(r, i1, i2);
  const int idx[3] = {(int)i0, (int)i1, (int)i2};
  int s[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int t = d.perm[a] == 0 ? idx[0] : (d.perm[a] == 1 ? idx[1] : idx[2]);
    s[a] = d.corner[a] + (d.flip[a] ? P - 1 - t : t);
  }
  const bool in = s[0] >= 0 && s[0] < X && s[1] >= 0 && s[1] < Y && s[2] >= 0 && s[2] < Z;
  const int64_t sv = in ? ((int64_t)s[0] * Y + s[1]) * Z + s[2] : 0;
  const int64_t XYZ = (int64_t)X * Y * Z;
  float* xo = x_out + ((int64_t)b * P3 + v) * xld;
  if (Cv == 4 && (xld & 3) == 0) {
    float4 q;
    q.x = in ? vol[sv] : 0.f; q.y = in ? vol[XYZ + sv] : 0.f; q.z = in ? vol[2 * XYZ + sv] : 0.f; q.w = in ? vol[3 * XYZ + sv] : 0.f;
    *reinterpret_cast<float4*>(xo) = q;
  } else {
    for (int c = 0; c < Cv; ++c) xo[c] = in ? vol[c * XYZ + sv] : 0.f;
  }
  if (t_out) {
    const int l = in ? (int)truth[sv] : 0;
    // generator.py:241-243 -- the inclusive "whole tumour" channel is labels {1, 2}: np.logical_or's third argument
    // is its OUT array there, so label 4 does not enter (reproduced, not corrected)
    const TT c0 = inclusive ? (TT)(l == 1 || l == 4) : (TT)(l == 1);
    const TT c1 = inclusive ? (TT)(l == 1 || l == 2) : (TT)(l == 2);
    const TT c2 = (TT)(l == 4);
    TT* to = t_out + (int64_t)b * 3 * P3 + v;
    to[0] = c0; to[P3] = c1; to[2 * (int64_t)P3] = c2;
  }
}

}  // namespace n3d

using namespace n3d;

extern "C" int n3d_patch_batch(const float* vol, int Cv, const uint8_t* truth, int X, int Y, int Z, const n3d_patch_desc* descs, int B, int P,
                               int flags, float* x_out, int64_t xld, void* t_out, void* stream) {
  N3D_CHECK_ARG(vol && descs && x_out && Cv >= 1 && X > 0 && Y > 0 && Z > 0 && P > 0 && B >= 1 && xld >= Cv, "patch_batch: bad args");
  N3D_CHECK_ARG(B <= N3D_PATCH_MAX_BATCH, "patch_batch: at most %d patches per call", N3D_PATCH_MAX_BATCH);
  N3D_CHECK_ARG(!t_out || truth, "patch_batch: targets requested without a truth volume");
  N3D_CHECK_ARG((int64_t)P * P * P < (1ll << 31) && (int64_t)X * Y * Z < (1ll << 40), "patch_batch: volume too large");
  PatchDescs pd;
  for (int i = 0; i < B; ++i) {
    pd.d[i] = descs[i];
    int seen = 0;
    for (int a = 0; a < 3; ++a) {
      N3D_CHECK_ARG(descs[i].perm[a] >= 0 && descs[i].perm[a] < 3, "patch_batch: perm entries must be 0..2");
      seen |= 1 << descs[i].perm[a];
    }
    N3D_CHECK_ARG(seen == 7, "patch_batch: perm must be a permutation of (0,1,2)");
  }
  for (int i = B; i < N3D_PATCH_MAX_BATCH; ++i) pd.d[i] = pd.d[0];
  const uint32_t P3 = (uint32_t)P * P * P;
  N3D_CHECK_ARG((flags & ~(N3D_PATCH_INCLUSIVE | N3D_PATCH_T_U8)) == 0, "patch_batch: unknown flag bits %d", flags);
  const int inclusive = flags & N3D_PATCH_INCLUSIVE;
  const dim3 grid((unsigned)cdiv(P3, 256), B);
  if (flags & N3D_PATCH_T_U8)
    hipLaunchKernelGGL(patch_batch_kernel<uint8_t>, grid, dim3(256), 0, (hipStream_t)stream, vol, Cv, truth, X, Y, Z, pd, P, inclusive, x_out, xld,
                       (uint8_t*)t_out, FastDiv((uint32_t)P), FastDiv((uint32_t)P * P));
  else
    hipLaunchKernelGGL(patch_batch_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, vol, Cv, truth, X, Y, Z, pd, P, inclusive, x_out, xld,
                       (float*)t_out, FastDiv((uint32_t)P), FastDiv((uint32_t)P * P));
  N3D_LAUNCH_CHECK();
  return N3D_OK;
}
