#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export N3D_FORCE_DP=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$1 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/$1.log 2>&1
tail -1 gpurun_out/$1.log | cut -c1-200
