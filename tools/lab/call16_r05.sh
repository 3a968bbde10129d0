mkdir -p gpurun_out/r05; cd $GRAFT_REPO_ROOT
{ echo "# tools/soak.py on the round-5 build: single process"; python3 tools/soak.py 30000 500 2>&1 | grep -v amdgpu;
  echo "# the same under the data-parallel code path (1-rank RCCL group, N3D_FORCE_DP=1, two buckets: all-reduces on the weight-gradient stream through n3d_comm_*)";
  N3D_FORCE_DP=1 N3D_DP_BUCKETS=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29591 python3 -c "
import os, sys, torch.distributed as dist
dist.init_process_group('nccl', rank=0, world_size=1)
sys.argv=['soak.py','20000','200']
exec(open('tools/soak.py').read())
dist.destroy_process_group()
" 2>&1 | grep -v "amdgpu\|^RCCL\|^HIP v\|^ROCm\|^Hostname\|^Librccl"; } > gpurun_out/r05/soak.log 2>&1
cat gpurun_out/r05/soak.log
CASE="f32 8 32 2 1" KNAME=conv_vox64_kernel NAME=conv_vox64_f32_2x8x32 TAG=r05 bash tools/collect_pmc_r05.sh > gpurun_out/r05/pmc_c8.log 2>&1; tail -3 gpurun_out/r05/pmc_c8.log | cut -c1-200
