// MFMA implicit-GEMM kernels for the FLOP-heavy 3x3x3 shapes (placeholder: filled in below).
#include "n3d_common.h"
namespace n3d {
int mfma_conv_try(const n3d_conv_geom*, bool, const float*, int64_t, const float*, const float*, float*, int64_t, int, const float*,
                  const float*, int64_t, const float*, double*, void*, size_t, hipStream_t) { return 0; }
int mfma_conv_stats_rows(const n3d_conv_geom*, bool, int) { return 0; }
}  // namespace n3d
