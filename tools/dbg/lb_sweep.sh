cd $GRAFT_REPO_ROOT
for v in 2 3 4; do L=tools/bin/libn3d_LB$v.so; echo "== launch_bounds(64,$v)"; for sz in 64 128; do ./tools/bin/conv_bench $L 4 $sz $sz $sz 1 2 50 32 2>&1 | tail -1; done; ./tools/bin/conv_bench $L 4 64 64 64 2 2 50 32 2>&1 | tail -1; ./tools/bin/conv_bench $L 8 32 32 32 1 2 50 32 2>&1 | tail -1; done
