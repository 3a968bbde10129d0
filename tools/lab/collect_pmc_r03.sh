#!/bin/bash
# PMC passes (one counter group per run, counters only) over the bf16 3x3x3 conv kernel of round 3 -- conv_vox64b_kernel<4,8,1,1,true>
# (dense two-voxels-per-slot image, 8-plane tiles) at (2,4,128^3) -> gpurun_out/r03/pmc_<case>.json via tools/pmc_summary.py,
# plus the kernel-only duration of the same launches from a kernel trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
for case in "bf16 4 128 2 1"; do
  tag=$(echo $case | tr ' ' '_')
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE"; do
    gt=$(echo $grp | cut -d' ' -f1)
    rm -rf $O/pmc_${tag}/pmc_$gt
    timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_${tag}/pmc_$gt -- python3 tools/conv_pmc.py $case 20 > $O/pmc_${tag}_$gt.log 2>&1
    echo "$tag $gt rc=$?"
  done
  python3 tools/pmc_summary.py $O/pmc_${tag} conv_vox64b $O/pmc_${tag}.json > /dev/null 2>&1
  rm -rf $O/pmc_${tag} $O/pmc_${tag}_*.log
  rm -rf $O/kt_${tag}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${tag} -- python3 tools/conv_pmc.py $case 200 > /dev/null 2>&1
  grep -h conv_vox64b $O/kt_${tag}/*/*kernel_stats.csv | head -2 > $O/pmc_${tag}_kernel_time.csv
  rm -rf $O/kt_${tag}
done
cat $O/pmc_*.json $O/pmc_*_kernel_time.csv
