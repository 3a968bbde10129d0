cd $GRAFT_REPO_ROOT
echo "default:"; python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-kernel-table --no-other-configs 2>&1 | grep -o '"ms_per_step": [0-9.]*'
echo "HIP_FORCE_DEV_KERNARG=1:"; HIP_FORCE_DEV_KERNARG=1 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-kernel-table --no-other-configs 2>&1 | grep -o '"ms_per_step": [0-9.]*'
echo "HIP_FORCE_DEV_KERNARG=0:"; HIP_FORCE_DEV_KERNARG=0 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-kernel-table --no-other-configs 2>&1 | grep -o '"ms_per_step": [0-9.]*'
export HIP_FORCE_DEV_KERNARG=1
bash tools/profile_r02.sh t64k --steps 10 --warmup 3
grep -h "final_batch\|pack_batch" gpurun_out/r02/t64k_kernel_stats.csv
