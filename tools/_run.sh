cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_nets.py tests/test_gpu_search.py tests/test_gpu_prims.py -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for v in "N3D_NODE_APPLY=0" "N3D_NODE_APPLY=1"; do echo "== $v"; for i in 1 2; do env $v python3 tools/search_phases.py 2>&1 | grep -E "drop_side|Error" | cut -c40-200; done; done
