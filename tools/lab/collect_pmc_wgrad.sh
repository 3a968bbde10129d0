cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04w; mkdir -p $O
for cs in "f32 4 128 2 1" "bf16 4 128 2 1" "f32 4 128 2 2" "bf16 4 128 2 2"; do
  tag=$(echo $cs | tr ' ' _)
  for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "FETCH_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    gt=$(echo $grp | cut -d' ' -f1)
    timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag/pmc_$gt -- python3 tools/wgrad_pmc.py $cs 6 > /dev/null 2>&1
  done
  python3 tools/pmc_summary.py $O/pmc_$tag vox_wgrad_kernel $O/pmc_wgrad_$tag.json > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -- python3 tools/wgrad_pmc.py $cs 30 > /dev/null 2>&1
  echo "== $cs"; grep -h vox_wgrad_kernel $O/kt_$tag/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-120; grep -v dispatch $O/pmc_wgrad_$tag.json | tr -d '\n'; echo
  rm -rf $O/pmc_$tag $O/kt_$tag
done
