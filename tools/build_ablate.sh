#!/bin/bash
# builds ablation variants of libn3d into tools/bin (scratch; never used by the product)
set -e
cd "$(dirname "$0")/.."
SRC=nas_3d_unet_amd/csrc
for v in ${VARIANTS:-base NO_STORE NO_MFMA NO_LOAD STAMP}; do
  D=""; [ $v != base ] && D="-DVOX_$v"; [ $v = STAMP_NOLDS ] && D="-DVOX_STAMP -DVOX_NO_LDSREAD"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast $D -shared -o tools/bin/libn3d_$v.so $SRC/n3d_core.hip $SRC/elementwise.hip $SRC/conv_generic.hip $SRC/conv_mfma.hip $SRC/conv_bf16.hip $SRC/head.hip $SRC/comm.hip $SRC/data_step.hip $SRC/post_step.hip -ldl &
done
wait
/opt/rocm/bin/hipcc -O2 tools/conv_bench.cpp -o tools/bin/conv_bench -ldl
