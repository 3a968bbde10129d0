#!/bin/bash
# Run on the GPU box (gpurun): regenerates the round-6 evidence under gpurun_out/r06c/ (copied into profiles/r06_* afterwards):
#   bench_line.log                 the default bench line (roofline, roofline_c8, roofline_by_time, cpu_baseline, other_configs, sync_timeouts)
#   {bench,search,p128_f32,p128_bf16}_kernel_stats.csv   rocprofv3 --kernel-trace --stats, one workload each (single-stream schedule), stdout next to it
#   {p128_f32,p128_bf16,search}_bench.log   the other workloads' own bench lines with their roofline_by_time tables
#   launch_table_seq_*.txt, deep_ab*.log, dp_host_budget.log, schedules.log, pmc_conv_vox64_f32_2x4x64*  tables, one-conv timings, DP host budget, PMC
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c; mkdir -p $O
python3 bench.py > $O/bench_line.log 2> $O/bench_line.err
prof() {  # prof <name> <bench args...>
  N=$1; shift
  rm -rf $O/trace_$N
  N3D_SIDE_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$N -- python3 bench.py "$@" > $O/${N}_stdout_under_rocprof.log 2>&1
  cp $O/trace_$N/*/*_kernel_stats.csv $O/${N}_kernel_stats.csv 2>/dev/null
  rm -rf $O/trace_$N
  grep metric $O/${N}_stdout_under_rocprof.log | cut -c1-200
}
prof bench --no-other-configs --no-kernel-table --no-cpu-baseline --steps 20 --warmup 5
prof search --workload search --steps 5 --warmup 2 --no-kernel-table --no-cpu-baseline
prof p128_f32 --size 128 --steps 6 --warmup 2 --no-kernel-table --no-cpu-baseline --no-other-configs
prof p128_bf16 --size 128 --dtype bf16 --steps 6 --warmup 2 --no-kernel-table --no-cpu-baseline --no-other-configs
python3 bench.py --size 128 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $O/p128_f32_bench.log 2>&1
python3 bench.py --size 128 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs > $O/p128_bf16_bench.log 2>&1
python3 bench.py --workload search --steps 10 --warmup 3 > $O/search_bench.log 2>&1
{
  echo "# data-parallel code path on a 1-rank RCCL group (N3D_FORCE_DP=1): ms per step, 30 steps after 5 warm-up, HIP-graph replay"
  for args in "" "--buckets 2" "--buckets 3" "--comm torch"; do
    N3D_FORCE_DP=1 MASTER_PORT=29577 python3 bench.py $args --no-other-configs --no-kernel-table --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | tail -1 | \
      python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('dp1 %-26s' % '$args', d['ms_per_step'], 'ms  buckets', d['config']['dp_buckets'], ' sync_timeouts', d['sync_timeouts'], ' schedule', d['config']['schedule'])"
  done
  python3 bench.py --no-other-configs --no-kernel-table --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('single GPU, no process group   ', d['ms_per_step'], 'ms')"
} > $O/schedules.log 2>&1
python3 tools/dp_host_budget.py 2 300 > $O/dp_host_budget.log 2>&1
python3 tools/dp_host_budget.py 1 300 >> $O/dp_host_budget.log 2>&1
python3 tools/deep_ab.py > $O/deep_ab.log 2>&1
DEEP_AB_MM=1 python3 tools/deep_ab.py > $O/deep_ab_mm_bf16.log 2>&1
for a in "64" "128" "128 2 bf16"; do python3 tools/table_seq.py $a > "$O/launch_table_seq_$(echo $a | tr ' ' _ | sed 's/_2_/_/').txt" 2>&1; done
python3 tools/side_timeline.py > $O/side_timeline.txt 2>&1
python3 tools/search_table.py 70 > $O/search_table.log 2>&1
CASE="f32 4 64 2 1" KNAME=conv_vox64_kernel NAME=conv_vox64_f32_2x4x64 TAG=r06 bash tools/collect_pmc_r05.sh > $O/pmc_f32.log 2>&1
cp gpurun_out/r05/pmc_conv_vox64_f32_2x4x64_r06* $O/ 2>/dev/null
TAG=after bash tools/collect_pmc_r06.sh > $O/collect_pmc_after.log 2>&1
cp gpurun_out/r06/pmc_gemm16_after.json $O/ 2>/dev/null
grep -h metric $O/bench_line.log | cut -c1-300
cat $O/schedules.log; cat $O/dp_host_budget.log
ls $O
