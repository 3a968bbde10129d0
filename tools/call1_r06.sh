#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
python3 bench.py > $O/bench_line_baseline.log 2> $O/bench_line_baseline.err
python3 tools/deep_ab.py > $O/deep_ab_before.log 2>&1
TAG=before bash tools/collect_pmc_r06.sh > $O/collect_pmc_before.log 2>&1
grep -h '"metric"' $O/bench_line_baseline.log | cut -c1-250
cat $O/deep_ab_before.log
cat $O/collect_pmc_before.log
