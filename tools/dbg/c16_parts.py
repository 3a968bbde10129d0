"""Time the pieces of the C=16 level of a 128^3 patch (2 x 16 x 32^3): forward / data gradient / weight gradient of the stride-1,
stride-2 and transposed 3x3x3 convs, each alone inside a HIP graph."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K, _lib
from nas_3d_unet_amd.prim_ops import _padding
dev = torch.device("cuda")

def timeit(fn, reps=10, rounds=3):
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn(); torch.cuda.synchronize()
        g.capture_begin(capture_error_mode="thread_local")
        for _ in range(reps): fn()
        g.capture_end()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)

C, S, B = (int(sys.argv[2]) if len(sys.argv) > 2 else 16), (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 2
for (stride, dil, transposed) in [(1, 1, False), (1, 2, False), (2, 1, False), (2, 2, False), (2, 1, True)]:
    pad = _padding(3, stride, dil)
    si = S // 2 if transposed else S          # conv-input side of the call
    so = S if transposed else S // stride
    x = K.as_view(K.empty_ndhwc(B, C, si, si, si, dev).normal_())
    y = K.as_view(K.empty_ndhwc(B, C, so, so, so, dev).normal_())
    w = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
    dw = torch.empty_like(w)
    g = K.conv_geom(B, so, so, so, C, C, 3, stride, dil, pad) if transposed else K.conv_geom(B, si, si, si, C, C, 3, stride, dil, pad)
    ctx = K.StepContext(dev)
    with K.step_context(ctx):
        K.conv_fwd(g, x, w, None, y, 0, None, None, transposed)
        K.conv_bwd_data(g, y, w, x, 0, None, None, transposed)
        ctx.freeze(); ctx.pack_all()
        tf = timeit(lambda: K.conv_fwd(g, x, w, None, y, 0, None, None, transposed))
        td = timeit(lambda: K.conv_bwd_data(g, y, w, x, 0, None, None, transposed))
        def wg():
            K.conv_bwd_weight(g, x, y, dw, None, 0, None, transposed)
            ctx.final.clear()
        tw = timeit(wg)
    print("C%d %s s%d d%d in %d^3 out %d^3: fwd %.1f us  dgrad %.1f us  wgrad %.1f us" % (C, "convT" if transposed else "conv ", stride, dil, si, so, tf, td, tw))

# the forward pair of a node as the net issues it: stride-1 + stride-2 conv from the same input level, with statistics
S2 = S
x = K.as_view(K.empty_ndhwc(B, C, S2, S2, S2, dev).normal_())
w1 = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05; w2 = torch.randn(C, C, 3, 3, 3, device=dev) * 0.05
g1 = K.conv_geom(B, S2, S2, S2, C, C, 3, 1, 1, 1); g2 = K.conv_geom(B, S2, S2, S2, C, C, 3, 2, 1, 1)
y1 = K.as_view(K.empty_ndhwc(B, C, S2, S2, S2, dev)); y2 = K.as_view(K.empty_ndhwc(B, C, S2 // 2, S2 // 2, S2 // 2, dev))
r1, r2 = K.conv_stats_rows(g1, False), K.conv_stats_rows(g2, False)
st1 = torch.zeros((B, r1, C, 2), dtype=torch.float64, device=dev); st2 = torch.zeros((B, r2, C, 2), dtype=torch.float64, device=dev)
ctx = K.StepContext(dev)
with K.step_context(ctx):
    b1 = torch.randn(C, device=dev); b2 = torch.randn(C, device=dev)
    calls = [(g1, x, w1, b1, y1, 0, None, st1, False), (g2, x, w2, b2, y2, 0, None, st2, False)]
    K.conv_fwd2(calls)
    ctx.freeze(); ctx.pack_all()
    print("rows", r1, r2)
    print("fwd2 (s1 + s2, stats): %.1f us" % timeit(lambda: K.conv_fwd2(calls)))
    print("fwd s1 stats alone: %.1f us" % timeit(lambda: K.conv_fwd(g1, x, w1, None, y1, 0, None, st1, False)))
    print("fwd s2 stats alone: %.1f us" % timeit(lambda: K.conv_fwd(g2, x, w2, None, y2, 0, None, st2, False)))

# the same stride-1 conv on channel slices of 48-channel concat buffers (voxel pitch 192 bytes), as inside a cell
big_x = K.empty_ndhwc(B, 48, S2, S2, S2, dev).normal_(); big_y = K.empty_ndhwc(B, 48, S2, S2, S2, dev)
for name, xs, ys in (("dense -> dense", x, y1), ("slice -> dense", K.as_view(big_x[:, 16:32]), y1), ("dense -> slice", x, K.as_view(big_y[:, 16:32])),
                     ("slice -> slice", K.as_view(big_x[:, 16:32]), K.as_view(big_y[:, 16:32]))):
    with K.step_context(ctx):
        print("fwd s1 stats %s (ld %d -> %d): %.1f us" % (name, xs.ld, ys.ld, timeit(lambda: K.conv_fwd(g1, xs, w1, None, ys, 0, None, st1, False))))
        print("fwd s1 stats + accumulate %s: %.1f us" % (name, timeit(lambda: K.conv_fwd(g1, xs, w1, None, ys, K.ACCUMULATE, None, st1, False))))

with K.step_context(ctx):
    for name, fill in (("relu(randn)", lambda t: t.normal_().clamp_(min=0)), ("denormals 1e-40", lambda t: t.fill_(1e-40)), ("zeros", lambda t: t.zero_()), ("randn*1e-20", lambda t: t.normal_().mul_(1e-20))):
        fill(x.t)
        print("fwd s1 stats, x = %s: %.1f us" % (name, timeit(lambda: K.conv_fwd(g1, x, w1, b1, y1, 0, None, st1, False))))
