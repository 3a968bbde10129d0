#!/bin/bash
# Build libn3d with one source compiled under an extra -D switch (ablation builds loaded through N3D_LIB=...):
#   tools/build_variant.sh VOX_NO_LOAD conv_mfma     -> tools/build/libn3d_VOX_NO_LOAD.so   (the 3x3x3 kernels without their halo fill)
# Runs here (hipcc cross-compiles gfx950); tools/build/ is git-ignored and travels to the GPU box with the snapshot.
set -e
DEF=$1; SRC=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/nas_3d_unet_amd/csrc
mkdir -p $ROOT/tools/build
make -C $CS -j8 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -D$DEF -c $CS/$SRC.hip -o $ROOT/tools/build/${SRC}_$DEF.o
OBJS=""
for f in n3d_core elementwise conv_generic conv_mfma data_step post_step head comm conv_bf16; do
  if [ "$f" = "$SRC" ]; then OBJS="$OBJS $ROOT/tools/build/${SRC}_$DEF.o"; else OBJS="$OBJS $CS/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/build/libn3d_$DEF.so $OBJS -ldl
echo built $ROOT/tools/build/libn3d_$DEF.so
