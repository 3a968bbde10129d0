#!/bin/bash
# usage: tools/prof_variants.sh "<variants>" C D H W dil B   -- prints rocprof kernel durations per variant
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in $1; do
  rm -rf /tmp/pv; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- $R/tools/bin/conv_bench $R/tools/bin/libn3d_$v.so $2 $3 $4 $5 $6 $7 30 > /dev/null 2>&1
  echo "== $v  C=$2 $3x$4x$5 dil=$6 B=$7"; cat /tmp/pv/*/*kernel_stats.csv | grep -v rocclr | cut -d, -f1-7 | cut -c1-110
done
