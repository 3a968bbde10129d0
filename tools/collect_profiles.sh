#!/bin/bash
# Run on the GPU box (gpurun): regenerates the evidence behind DESIGN.md section 5 under gpurun_out/r01/.
#   1. the default bench run (JSON line incl. roofline + cpu_baseline)
#   2. rocprofv3 --kernel-trace --stats of the same bench command (per-kernel durations)
#   3. PMC passes (one counter group per run, counters only) over the dominant kernel via tools/bin/conv_bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r01; rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench_stdout.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/trace.log 2>&1
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
for grp in "FETCH_SIZE WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -- ./tools/bin/conv_bench nas_3d_unet_amd/libn3d.so 4 64 64 64 1 2 30 32 > $O/pmc_$tag.log 2>&1
done
tail -1 $O/bench_stdout.log | cut -c1-300
ls $O
