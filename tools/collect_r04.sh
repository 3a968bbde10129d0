#!/bin/bash
# Run on the GPU box (gpurun): regenerates the round-4 evidence under gpurun_out/r04/ (copied into profiles/r04_* afterwards):
#   bench_line.log                 the default bench line (roofline, roofline_by_time, cpu_baseline, other_configs, sync_timeouts)
#   {bench,search,p128_f32,p128_bf16}_kernel_stats.csv   rocprofv3 --kernel-trace --stats, one workload each (single-stream schedule: the
#                                  kernel trace serialises the streams), stdout next to it
#   side_timeline*.txt + side_stamps*.txt   timeline of the three-stream schedule from device clock stamps, the raw stamp dump, busy fractions
#   schedules.log                  side schedule off / auto / forced, 1-rank RCCL group with 1 / 2 / 3 buckets (tools/collect_schedules.sh)
#   nol_probe.log, conv_ab.log     the normalise-on-load probe; forward / data-gradient timing of the C in {4, 8} convs on dense tensors
#   search_timeline.txt, search_table.log, search_phases.log   the search step: joins / cuts of both passes from device stamps, launch table, phases
#   pmc_wgrad_*.json + pmc_wgrad.log   PMC of the stride-1 weight-gradient kernel, fp32 and bf16 storage, (2,4,128^3) d = 1 / 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
python3 bench.py > $O/bench_line.log 2>&1
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
bash tools/collect_schedules.sh $O > /dev/null 2>&1
python3 tools/side_timeline.py --raw $O/side_stamps.txt > $O/side_timeline.txt 2>&1
python3 tools/side_timeline.py --size 128 --dtype bf16 --raw $O/side_stamps_p128_bf16.txt > $O/side_timeline_p128_bf16.txt 2>&1
python3 tools/side_timeline.py --size 128 > $O/side_timeline_p128_f32.txt 2>&1
python3 tools/search_timeline.py > $O/search_timeline.txt 2>&1
python3 tools/search_table.py 70 > $O/search_table.log 2>&1
python3 tools/search_phases.py > $O/search_phases.log 2>&1
bash tools/collect_pmc_wgrad.sh > $O/pmc_wgrad.log 2>&1; cp gpurun_out/r04w/pmc_wgrad_*.json $O/ 2>/dev/null
python3 tools/nol_probe.py > $O/nol_probe.log 2>&1
python3 tools/conv_ab.py > $O/conv_ab.log 2>&1
for a in "64" "128" "128 2 bf16"; do python3 tools/table_seq.py $a > "$O/launch_table_seq_$(echo $a | tr ' ' _ | sed 's/_2_/_/').txt" 2>&1; done
grep -h metric $O/bench_line.log | cut -c1-300
ls $O
