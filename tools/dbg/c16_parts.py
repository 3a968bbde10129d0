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

C, S, B = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 2
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
    print("C16 %s s%d d%d in %d^3 out %d^3: fwd %.1f us  dgrad %.1f us  wgrad %.1f us" % ("convT" if transposed else "conv ", stride, dil, si, so, tf, td, tw))
