#!/bin/bash
# Round 6, VERDICT r5 item 1(a): PMC passes over the C >= 16 conv family (conv_gemm16_pair / conv_bwd16_quad / conv_bwd16_dual and
# the tile16 kernels) -- one counter group per run, counters only, never combined with a trace; then a kernel-trace run of the same
# command for the durations.  Output: gpurun_out/r06/pmc_gemm16_$TAG.json  (TAG=before|after; N3D_LIB selects another build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
TAG=${TAG:-before}
D=$O/pmc_gemm16_$TAG; rm -rf $D; mkdir -p $D
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  gt=$(echo $grp | cut -d' ' -f1)
  timeout -k 5 200 rocprofv3 --pmc $grp --output-format csv -d $D/pmc_$gt -- python3 tools/g16_pmc.py 12 $CASES > $D/pmc_$gt.log 2>&1
  echo "pmc $gt rc=$?"
done
timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $D/kt -- python3 tools/g16_pmc.py 50 $CASES > $D/kt.log 2>&1
echo "kt rc=$?"
python3 tools/g16_pmc_summary.py $D $O/pmc_gemm16_$TAG.json > /dev/null 2> $D/summary.err; echo "summary rc=$?"
tail -3 $D/*.log | cut -c1-200 > $O/pmc_gemm16_${TAG}_logs.txt; cat $D/summary.err >> $O/pmc_gemm16_${TAG}_logs.txt
rm -rf $D
