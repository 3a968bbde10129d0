// rows64 kernel of the deep U-net levels (conv_r64.hip): argument block, plan and host entry points shared with conv_mfma.hip
#pragma once
#include "n3d_common.h"

namespace n3d {

struct R64Args {
  const float* src; int64_t sld; int Ds, Hs, Ws, Cs;
  float* dst; int64_t dld; int Dd, Hd, Wd, Cd;
  const float* wq;        // packed [tap][Cd / 4][Cs / 4][j = cd & 3][e = cs & 3]
  const float* bias;
  int k, sn, off, dt, den, flags, B;
  const float* in_gate;
  const float* relu_src; int64_t rld;
  const float* out_gate;
  double* stats; int rows_per_sample;
  int nct;                // column tiles (CT quads each) per row tile
  int Nd, Ns, HWs, HWd;   // voxels per sample (destination / source), voxels per source / destination plane
  int xslots;             // float4 slots of the staged-voxel area (a multiple of 64; the zero slot follows it)
  int gslots;             // float4 slots of the input-gate table (samples of a tile x Cs / 4, a multiple of 64)
  FastDiv fNd, fWd, fHd, fnct, fHWd;
};

struct R64Plan { bool ok; int ct, ntile, xslots, gslots; size_t lds; };

R64Plan r64_plan(const n3d_conv_geom* g, bool data_grad);
int r64_stats_rows(const n3d_conv_geom* g, bool data_grad);
// fills the arguments for one conv (packing the weights unless N3D_PREPACKED); 1 = ready, 0 = shape not served (nothing touched), < 0 error
int r64_prepare(const n3d_conv_geom* g, bool data_grad, const float* src, int64_t sld, const float* w, const float* bias, float* dst,
                int64_t dld, int flags, const float* in_gate, const float* relu_src, int64_t rld, const float* out_gate, double* stats,
                void* ws, size_t ws_bytes, hipStream_t s, R64Args* out, R64Plan* plan);
// n = 1 .. 4 prepared convs of one column-tile width in ONE launch; 1 = launched, < 0 error
int r64_launch(int n, const R64Args* as, const R64Plan* ps, hipStream_t s);

}  // namespace n3d
