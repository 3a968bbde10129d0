#!/bin/bash
# Run on the GPU box (gpurun): regenerates the evidence behind DESIGN.md section 5 under gpurun_out/r01/.
#   1. the default bench run (JSON line incl. roofline + cpu_baseline)
#   2. rocprofv3 --kernel-trace --stats of the same bench command (per-kernel durations)
#   (PMC passes over the dominant kernel: tools/collect_pmc.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r01; mkdir -p $O; rm -rf $O/trace
python3 bench.py > $O/bench_stdout.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/trace.log 2>&1
cp $O/trace/*/*_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
# PMC passes: tools/collect_pmc.sh (one counter group per run, each under `timeout`: a group the hardware cannot collect
# in one pass aborts rocprofv3 and then hangs until killed)
tail -1 $O/bench_stdout.log | cut -c1-300
ls $O
