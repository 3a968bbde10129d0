#!/bin/bash
# Run on the GPU box (gpurun): regenerates the round-2 evidence under gpurun_out/r02/ (copied into profiles/r02_* afterwards):
#   bench_line.log              the default bench line (roofline, roofline_by_time, cpu_baseline)
#   {bench,search,p128_f32,p128_bf16}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the four workloads
#   dp_variants.log             1-rank RCCL group on one GPU: single bucket / two buckets / C-ABI comm wrapper
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
python3 bench.py > $O/bench_line.log 2>&1
tools/profile_r02.sh bench          # the default command itself: the roofline loop of the conv kernel is part of the profile
tools/profile_r02.sh search --workload search --steps 5 --warmup 2 --no-kernel-table --no-cpu-baseline
tools/profile_r02.sh p128_f32 --size 128 --steps 6 --warmup 2 --no-kernel-table --no-cpu-baseline
tools/profile_r02.sh p128_bf16 --size 128 --dtype bf16 --steps 6 --warmup 2 --no-kernel-table --no-cpu-baseline
python3 bench.py --workload search --steps 10 --warmup 3 > $O/search_stdout.log 2>&1
python3 bench.py --size 128 --steps 20 --warmup 3 --no-cpu-baseline > $O/p128_f32_bench.log 2>&1
python3 bench.py --size 128 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > $O/p128_bf16_bench.log 2>&1
for v in "N3D_X=1" "N3D_FORCE_DP=1" "N3D_FORCE_DP=1 N3D_DP_BUCKETS=2" "N3D_FORCE_DP=1 N3D_COMM=rccl" "N3D_FORCE_DP=1 N3D_DP_BUCKETS=2 N3D_COMM=rccl"; do
  echo "== $v"; env $v python3 bench.py --steps 100 --warmup 10 --no-roofline --no-cpu-baseline --no-kernel-table --no-other-configs 2>&1 | grep metric | cut -c1-330
done > $O/dp_variants.log 2>&1
grep -h metric $O/bench_line.log | cut -c1-200
ls $O
