cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04x; mkdir -p $O
for cs in "bf16 4 128 2 1" "bf16 4 128 2 2" "bf16 8 64 2 1" "f32 4 128 2 1"; do
  tag=$(echo $cs | tr ' ' _)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$tag -- python3 tools/wgrad_pmc.py $cs 30 > /dev/null 2>&1
  echo "== $cs"; python3 - $O/kt_$tag <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "vox_wgrad" in r["Name"]: print("  %-70s avg %.1f us" % (r["Name"][:70], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/kt_$tag
done
python -m pytest tests/test_gpu_conv.py tests/test_gpu_bf16.py -q -x 2>&1 | tail -2
for i in 1 2; do python3 bench.py --size 128 --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-kernel-table --no-other-configs 2>&1 | grep metric | cut -c1-160; done
for i in 1 2; do python3 bench.py --size 128 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-kernel-table --no-other-configs 2>&1 | grep metric | cut -c1-160; done
for i in 1 2; do python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-kernel-table --no-other-configs 2>&1 | grep metric | cut -c1-160; done
