# 1-GPU schedule variants of the headline step (profiles/rNN_schedules.log): side schedule off / auto / forced, and the data-parallel code
# path in a 1-rank RCCL group (N3D_FORCE_DP=1): one bucket, two buckets on device flags, two buckets as event-tied graph segments
O=${1:-gpurun_out/schedules}
mkdir -p $O
run() { echo -n "$1 : "; shift; env "$@" python bench.py --no-other-configs --no-cpu-baseline --no-kernel-table --no-roofline --steps 30 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], 'ms', d['value'], 'patches/s', d['config']['schedule'], d['config']['schedule_ms'], 'buckets', d['config']['dp_buckets'], 'timeouts', d['sync_timeouts'])"; }
{
EXTRA="" run "single GPU, side schedule off        " N3D_SIDE_WGRAD=0
EXTRA="" run "single GPU, auto                     " N3D_NONE=1
EXTRA="" run "single GPU, side schedule forced     " N3D_SIDE_WGRAD=force
EXTRA="" run "1-rank RCCL group, one bucket        " N3D_FORCE_DP=1
EXTRA="--buckets 2" run "1-rank RCCL group, two buckets, flags" N3D_FORCE_DP=1
EXTRA="--buckets 3" run "1-rank RCCL group, three buckets, flags" N3D_FORCE_DP=1
EXTRA="--buckets 2" run "1-rank RCCL group, two buckets, event-tied segments (no side schedule)" N3D_FORCE_DP=1 N3D_SIDE_WGRAD=0
EXTRA="--buckets 2 --comm rccl" run "1-rank RCCL group, two buckets, flags, n3d_comm" N3D_FORCE_DP=1
} 2>&1 | tee $O/schedules.log
