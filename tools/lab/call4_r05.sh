mkdir -p gpurun_out/r05; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bf16.py -q -x > gpurun_out/r05/t_bf16.log 2>&1; tail -8 gpurun_out/r05/t_bf16.log
export CONV_AB_DT=bf16
for lib in "" tools/build/libn3d_VXB_NO_MARCH.so; do
  if [ -n "$lib" ]; then export N3D_LIB=$GRAFT_REPO_ROOT/$lib; else unset N3D_LIB; fi
  python3 tools/conv_ab.py 4 128 1 2 4 128 2 2 2>&1 | grep -v amdgpu.ids
  python3 tools/conv_ab.py 4 128 1 2 4 128 2 2 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05/bf16_conv_ab_march.log 2>&1
unset N3D_LIB CONV_AB_DT
cat gpurun_out/r05/bf16_conv_ab_march.log
TAG=march bash tools/collect_pmc_r05.sh > gpurun_out/r05/pmc_march.log 2>&1; tail -32 gpurun_out/r05/pmc_march.log
