#!/bin/bash
# Run on the GPU box (gpurun): regenerates the round-3 evidence under gpurun_out/r03/ (copied into profiles/r03_* afterwards):
#   bench_line.log                 the default bench line (roofline, roofline_by_time, cpu_baseline, other_configs)
#   bench_kernel_stats.csv         rocprofv3 --kernel-trace --stats of the HEADLINE workload only (--no-other-configs: train step + the
#                                  roofline loop of the conv kernel), stdout of that run in bench_stdout_under_rocprof.log
#   {search,p128_f32,p128_bf16}_kernel_stats.csv + *_bench.log   the other three workloads, one summary each
#   side_timeline*.txt             timeline of the side-stream schedule from device clock stamps (tools/side_timeline.py)
#   side_phases.log, search_phases.log   main graph / tail times with the side work dropped and running (tools/*_phases.py)
#   two_chain_probe.log, handoff_cost.log, seg_overlap.log   the scheduling probes of DESIGN.md section 5
#   schedules.log                  train step with the side schedule off / on / forced, and the 1-rank RCCL variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
python3 bench.py > $O/bench_line.log 2>&1
prof() {  # prof <name> <bench args...>
  N=$1; shift
  rm -rf $O/trace_$N
  # N3D_SIDE_WGRAD=0: rocprofv3's kernel trace serialises the two streams (a forced side schedule runs 19 ms per step under it and the
  # schedule choice picks the single-stream graph anyway), so the per-kernel summaries are taken on the single-stream schedule -- the
  # same kernels, minus the one-lane hand-off kernels
  N3D_SIDE_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$N -- python3 bench.py "$@" > $O/${N}_stdout_under_rocprof.log 2>&1
  cp $O/trace_$N/*/*_kernel_stats.csv $O/${N}_kernel_stats.csv 2>/dev/null
  rm -rf $O/trace_$N
  grep metric $O/${N}_stdout_under_rocprof.log | cut -c1-200
}
prof bench --no-other-configs --no-kernel-table --no-cpu-baseline --steps 20 --warmup 5
prof search --workload search --steps 5 --warmup 2 --no-kernel-table --no-cpu-baseline
prof p128_f32 --size 128 --steps 6 --warmup 2 --no-kernel-table --no-cpu-baseline --no-other-configs
prof p128_bf16 --size 128 --dtype bf16 --steps 6 --warmup 2 --no-kernel-table --no-cpu-baseline --no-other-configs
python3 bench.py --workload search --steps 10 --warmup 3 > $O/search_bench.log 2>&1
python3 bench.py --size 128 --steps 20 --warmup 3 --no-cpu-baseline > $O/p128_f32_bench.log 2>&1
python3 bench.py --size 128 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > $O/p128_bf16_bench.log 2>&1
for v in "N3D_SIDE_WGRAD=0" "N3D_SIDE_WGRAD=1" "N3D_SIDE_WGRAD=force" "N3D_FORCE_DP=1" "N3D_FORCE_DP=1 N3D_DP_BUCKETS=2" "N3D_FORCE_DP=1 N3D_COMM=rccl"; do
  echo "== $v"; env $v python3 bench.py --steps 100 --warmup 10 --no-roofline --no-cpu-baseline --no-kernel-table --no-other-configs 2>&1 | grep metric | cut -c1-420
done > $O/schedules.log 2>&1
python3 tools/two_chain_probe.py > $O/two_chain_probe.log 2>&1
for t in handoff_cost seg_overlap; do
  [ -x tools/bin/$t ] || /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/$t.cpp -o tools/bin/$t 2>/dev/null
  timeout 120 tools/bin/$t > $O/$t.log 2>&1
done
python3 tools/side_timeline.py > $O/side_timeline.txt 2>&1
python3 tools/side_timeline.py --size 128 --dtype bf16 > $O/side_timeline_p128_bf16.txt 2>&1
python3 tools/side_phases.py > $O/side_phases.log 2>&1
python3 tools/search_phases.py > $O/search_phases.log 2>&1
grep -h metric $O/bench_line.log | cut -c1-200
ls $O
