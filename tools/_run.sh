cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "X=1" "N3D_SIDE_PAIRS=0" "N3D_SIDE_PAIRS_BOTH=0" "N3D_SIDE_PAIRS_BWD=0" "N3D_SIDE_STEM1_BWD=0" "N3D_SIDE_PRE0_BWD=0" "N3D_SIDE_MIN_QUEUE=3" "N3D_SIDE_MIN_QUEUE=8" "N3D_SIDE_EARLY_AT=0"; do
  echo "== $v"
  env $v python3 bench.py --size 128 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-kernel-table --no-other-configs 2>&1 | grep metric | python3 -c "import sys,json; [print('   f32', json.loads(l)['ms_per_step']) for l in sys.stdin]"
  env $v python3 bench.py --size 128 --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-kernel-table --no-other-configs 2>&1 | grep metric | python3 -c "import sys,json; [print('   bf16', json.loads(l)['ms_per_step']) for l in sys.stdin]"
done
