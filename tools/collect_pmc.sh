#!/bin/bash
# PMC passes over the dominant kernel (run on the GPU box): ONE counter group per run, counters only, each under `timeout`
# (a group the hardware cannot collect in one pass aborts rocprofv3 and then hangs until killed).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r01; mkdir -p $O; rm -rf $O/pmc_*
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -- ./tools/bin/conv_bench nas_3d_unet_amd/libn3d.so 4 64 64 64 1 2 30 32 > $O/pmc_$tag.log 2>&1
  echo "$tag rc=$?"
done
ls $O
