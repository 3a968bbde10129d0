#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$1 -- python3 bench.py --size 128 --batch 2 --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/$1.log 2>&1
tail -1 gpurun_out/$1.log | cut -c1-200
