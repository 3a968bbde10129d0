mkdir -p gpurun_out/r05; cd $GRAFT_REPO_ROOT
export CONV_AB_DT=bf16 CONV_AB_EXTRA=1
for lib in "" tools/build/libn3d_VXM_NOSTORE.so tools/build/libn3d_VXM_NOMFMA.so tools/build/libn3d_VXM_NOFILL.so tools/build/libn3d_VXB_NO_MARCH.so; do
  if [ -n "$lib" ]; then export N3D_LIB=$GRAFT_REPO_ROOT/$lib; else unset N3D_LIB; fi
  python3 tools/conv_ab.py 4 128 1 2 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05/bf16_march_ablation.log 2>&1
cat gpurun_out/r05/bf16_march_ablation.log
