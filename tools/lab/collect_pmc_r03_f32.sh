#!/bin/bash
# PMC passes (one counter group per run, counters only) over the HEADLINE conv kernel of bench.py's roofline object -- conv_vox64_kernel<4,4,1,1>,
# 3x3x3 stride-1 conv, C = 4, (2,4,64^3) fp32 -> gpurun_out/r03/pmc_conv_vox64_f32_2x4x64.json via tools/pmc_summary.py, plus the kernel-only
# duration of the same launches from a kernel trace (round-3 build: the kernel starts with s_setprio 3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
case="f32 4 64 2 1"; tag=conv_vox64_f32_2x4x64
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE"; do
  gt=$(echo $grp | cut -d' ' -f1)
  rm -rf $O/pmc_${tag}/pmc_$gt
  timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_${tag}/pmc_$gt -- python3 tools/conv_pmc.py $case 20 > $O/pmc_${tag}_$gt.log 2>&1
  echo "$tag $gt rc=$?"
done
python3 tools/pmc_summary.py $O/pmc_${tag} conv_vox64_kernel $O/pmc_${tag}.json > /dev/null 2>&1
rm -rf $O/pmc_${tag} $O/pmc_${tag}_*.log $O/kt_${tag}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${tag} -- python3 tools/conv_pmc.py $case 200 > /dev/null 2>&1
grep -h conv_vox64_kernel $O/kt_${tag}/*/*kernel_stats.csv | head -2 > $O/pmc_${tag}_kernel_time.csv
rm -rf $O/kt_${tag}
cat $O/pmc_${tag}.json $O/pmc_${tag}_kernel_time.csv
