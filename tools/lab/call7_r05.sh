mkdir -p gpurun_out/r05; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_head.py -q -x > gpurun_out/r05/t_head.log 2>&1; tail -15 gpurun_out/r05/t_head.log
python -m pytest tests -q -m gpu > gpurun_out/r05/gpu_tests3.log 2>&1; tail -4 gpurun_out/r05/gpu_tests3.log
for cfg in "--size 128" "--size 128 --dtype bf16" ""; do
  python bench.py $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-kernel-table --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['ms_per_step'], d['config']['schedule_ms'])"
done > gpurun_out/r05/planar_bench.log 2>&1
cat gpurun_out/r05/planar_bench.log
