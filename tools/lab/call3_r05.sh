mkdir -p gpurun_out/r05; cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/r05/gpu_tests.log 2>&1; tail -15 gpurun_out/r05/gpu_tests.log
export CONV_AB_DT=bf16
for lib in "" tools/build/libn3d_BF16_R04.so tools/build/libn3d_VXB_ONE_CHAIN.so; do
  if [ -n "$lib" ]; then export N3D_LIB=$GRAFT_REPO_ROOT/$lib; else unset N3D_LIB; fi
  python3 tools/conv_ab.py 4 128 1 2 4 128 2 2 4 64 1 2 8 64 1 2 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r05/bf16_conv_ab.log 2>&1
unset N3D_LIB CONV_AB_DT
cat gpurun_out/r05/bf16_conv_ab.log
bash tools/same_x_bound.sh > gpurun_out/r05/conv_ab.log 2>&1; cat gpurun_out/r05/conv_ab.log
TAG=after bash tools/collect_pmc_r05.sh > gpurun_out/r05/pmc_after.log 2>&1
N3D_LIB=$GRAFT_REPO_ROOT/tools/build/libn3d_BF16_R04.so TAG=before bash tools/collect_pmc_r05.sh > gpurun_out/r05/pmc_before.log 2>&1
tail -30 gpurun_out/r05/pmc_before.log; tail -30 gpurun_out/r05/pmc_after.log
