cd $GRAFT_REPO_ROOT
L=nas_3d_unet_amd/libn3d.so
for d in 1 2; do
echo "== tile16 dil=$d"; ./tools/bin/conv_bench $L 16 32 32 32 $d 2 50 32 | tail -2
echo "== gemm16 dil=$d"; N3D_NO_TILE16=1 ./tools/bin/conv_bench $L 16 32 32 32 $d 2 50 32 | tail -2
done
echo "== 64^3 C16 (B=1)"; ./tools/bin/conv_bench $L 16 64 64 64 1 1 20 32 | tail -1;  N3D_NO_TILE16=1 ./tools/bin/conv_bench $L 16 64 64 64 1 1 20 32 | tail -1
