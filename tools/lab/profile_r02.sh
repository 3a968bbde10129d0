#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel-trace summaries of the round-2 workloads -> gpurun_out/r02/<name>_kernel_stats.csv
#   usage: tools/profile_r02.sh <name> <bench.py args...>     (the args are passed as they are: `bench` with none = the default command)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=$1; shift
O=gpurun_out/r02; mkdir -p $O; rm -rf $O/trace_$N
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$N -- python3 bench.py "$@" > $O/${N}_stdout.log 2>&1
cp $O/trace_$N/*/*_kernel_stats.csv $O/${N}_kernel_stats.csv 2>/dev/null
rm -rf $O/trace_$N
grep metric $O/${N}_stdout.log | cut -c1-220
