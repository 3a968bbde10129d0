#!/usr/bin/env python3
"""Per-launch cost of individual libn3d kernels on tiny tensors, replayed from a HIP graph (no host cost)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nas_3d_unet_amd import kernels as K, programs as P, prim_ops

dev = torch.device("cuda")


def bench(name, fn, n=200):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); e1.synchronize()
    print("%-46s %7.2f us/call" % (name, e0.elapsed_time(e1) * 1e3 / (5 * n)))


for (C, S) in ((64, 4), (32, 8), (16, 16), (8, 32), (4, 64)):
    B = 2
    x = K.as_view(K.empty_ndhwc(B, C, S, S, S, dev).normal_())
    y = K.as_view(K.empty_ndhwc(B, C, S, S, S, dev))
    d = K.as_view(K.empty_ndhwc(B, C, S, S, S, dev).normal_())
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    bias = torch.zeros(C, device=dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    g = K.conv_geom(B, S, S, S, C, C, 3, 1, 1, 1)
    rows = K.conv_stats_rows(g, False)
    G = P.group_count(C)
    print("---- C=%d %d^3 B=%d (conv stats rows %d, ew rows %d)" % (C, S, B, rows, K.stats_rows(S ** 3, C)))
    stats = torch.zeros((B, max(rows, 1), C, 2), dtype=torch.float64, device=dev)
    bench("conv_fwd (+pack)", lambda: K.conv_fwd(g, x, w, bias, y, 0, None, stats if rows > 0 else None, False))
    bench("conv_bwd_data (+pack)", lambda: K.conv_bwd_data(g, d, w, y, 0, None, None, False))
    dw, db = torch.empty_like(w), torch.empty_like(bias)
    bench("conv_bwd_weight (+final)", lambda: K.conv_bwd_weight(g, x, d, dw, db, 0, None, False))
    st, r2 = K.channel_stats(x)
    bench("channel_stats", lambda: K.channel_stats(x))
    if r2 <= K.fused_max_rows():
        bench("affine_act_gn (E1 fused)", lambda: K.affine_act_gn(x, st, r2, gam, bet, G, 1e-5, None, y, K.RELU))
    a, b, mr, sr = K.gn_coeffs(st, r2, gam, bet, B, C, G, S ** 3)
    bench("gn_coeffs", lambda: K.gn_coeffs(st, r2, gam, bet, B, C, G, S ** 3))
    bench("affine_act (E1)", lambda: K.affine_act(x, a, b, None, y, K.RELU))
    sums, r3 = K.affine_act_bwd_reduce(d, x, a, b, K.RELU)
    bench("affine_act_bwd_reduce (E2)", lambda: K.affine_act_bwd_reduce(d, x, a, b, K.RELU))
    if r3 <= K.fused_max_rows():
        bench("affine_act_bwd_apply_gn (E3 fused)", lambda: K.affine_act_bwd_apply_gn(d, x, a, b, sums, r3, torch.nn.Parameter(gam), torch.nn.Parameter(bet), mr, None, sr, None, y, G, K.RELU))
    gp, bp = torch.nn.Parameter(gam), torch.nn.Parameter(bet)
    bench("gn_bwd_coeffs", lambda: K.gn_bwd_coeffs(sums, r3, gp, mr, None, B, C, G, S ** 3, None, bp, sr, None))
    A = torch.ones((B, C), device=dev)
    bench("affine_act_bwd_apply (E3)", lambda: K.affine_act_bwd_apply(d, x, a, b, A, A, A, y, K.RELU))
