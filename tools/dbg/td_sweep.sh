cd /root/repo
L=tools/bin/libn3d_TUNE.so
for sz in 64 128; do for td in 1 2 4; do echo "== $sz td=$td"; VOX_TD=$td ./tools/bin/conv_bench $L 4 $sz $sz $sz 1 2 50 32 2>&1 | tail -2; done; done
for sz in 64 128; do for td in 1 2; do echo "== C8 $((sz/2)) td=$td"; VOX_TD=$td ./tools/bin/conv_bench $L 8 $((sz/2)) $((sz/2)) $((sz/2)) 1 2 50 32 2>&1 | tail -2; done; done
