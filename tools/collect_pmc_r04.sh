#!/bin/bash
# PMC passes of round 4 (one counter group per run, counters only -- never combined with a trace):
#   * the three node-epilogue streaming kernels at (2, 128^3, 4), node as a DENSE tensor (round 4) vs a 16-of-48-byte SLICE (round 3):
#     FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B) + WRITE_SIZE, KiB, against the algorithmic bytes -> gpurun_out/r04/pmc_ew_{dense,slice}.json
#   * the headline conv kernel of bench.py's roofline object on the round-4 build -> gpurun_out/r04/pmc_conv_vox64_f32_2x4x64.json (+ kernel time)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
for lay in dense slice; do
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf $O/pmc_ew_$lay/pmc_$grp
    timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_ew_$lay/pmc_$grp -- python3 tools/ew_pmc.py $lay 128 2 10 > $O/pmc_ew_${lay}_$grp.log 2>&1
    echo "ew $lay $grp rc=$?"
  done
  for k in affine_act_gn2_kernel affine_bwd_reduce2_kernel affine_bwd_apply_gn2_kernel; do
    python3 tools/pmc_summary.py $O/pmc_ew_$lay $k $O/pmc_ew_${lay}_$k.json > /dev/null 2>&1
  done
  rm -rf $O/kt_ew_$lay
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ew_$lay -- python3 tools/ew_pmc.py $lay 128 2 50 > /dev/null 2>&1
  grep -h "affine_act_gn2_kernel\|affine_bwd_reduce2_kernel\|affine_bwd_apply_gn2_kernel" $O/kt_ew_$lay/*/*kernel_stats.csv > $O/pmc_ew_${lay}_kernel_time.csv
  rm -rf $O/pmc_ew_$lay $O/pmc_ew_${lay}_*.log $O/kt_ew_$lay
done
case="f32 4 64 2 1"; tag=conv_vox64_f32_2x4x64
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
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
ls $O | grep pmc; cat $O/pmc_ew_dense_*.json $O/pmc_ew_slice_*.json | grep -v dispatch | head -60; cat $O/pmc_ew_*_kernel_time.csv | cut -c1-160
