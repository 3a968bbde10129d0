mkdir -p gpurun_out/r05; cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/r05/gpu_tests2.log 2>&1; tail -6 gpurun_out/r05/gpu_tests2.log
python tools/search_table.py 70 gpurun_out/r05/search_seq.txt > gpurun_out/r05/search_table.log 2>&1; head -100 gpurun_out/r05/search_table.log
