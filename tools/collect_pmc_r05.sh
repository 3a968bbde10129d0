#!/bin/bash
# PMC passes of round 5 over the bf16-storage 3x3x3 conv at (2,4,128^3) (conv_vox64b_kernel<4,8,1,1,true,0>): one counter
# group per run, counters only -- never combined with a trace -- then the kernel's time from a kernel-trace run of the same command.
#   TAG names the output:  TAG=after tools/collect_pmc_r05.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
TAG=${TAG:-after}
#   CASE="f32 4 64 2 1" KNAME=conv_vox64_kernel NAME=conv_vox64_f32_2x4x64 selects another conv (tools/conv_pmc.py arguments, kernel substring, output name)
case=${CASE:-"bf16 4 128 2 1"}; KNAME=${KNAME:-conv_vox64b_}; tag=${NAME:-conv_vox64b_bf16_2x4x128}_$TAG
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  gt=$(echo $grp | cut -d' ' -f1)
  rm -rf $O/pmc_${tag}/pmc_$gt
  timeout -k 5 150 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_${tag}/pmc_$gt -- python3 tools/conv_pmc.py $case 20 > $O/pmc_${tag}_$gt.log 2>&1
  echo "$tag $gt rc=$?"
done
python3 tools/pmc_summary.py $O/pmc_${tag} $KNAME $O/pmc_${tag}.json > /dev/null 2>&1
rm -rf $O/pmc_${tag} $O/pmc_${tag}_*.log $O/kt_${tag}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${tag} -- python3 tools/conv_pmc.py $case 200 > /dev/null 2>&1
grep -h "$KNAME" $O/kt_${tag}/*/*kernel_stats.csv | head -2 > $O/pmc_${tag}_kernel_time.csv
rm -rf $O/kt_${tag}
cat $O/pmc_${tag}.json | grep -v dispatch | head -40; cat $O/pmc_${tag}_kernel_time.csv | cut -c1-200
