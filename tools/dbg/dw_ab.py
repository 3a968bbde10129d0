import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K
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
for (c, S) in [(4, 64), (8, 32), (16, 16), (32, 8)]:
    x = K.as_view(K.empty_ndhwc(2, c, S, S, S, dev).normal_()); dy = K.as_view(K.empty_ndhwc(2, c, S, S, S, dev).normal_())
    w = torch.randn(c, 1, 3, 3, 3, device=dev); dw = torch.empty_like(w)
    g = K.conv_geom(2, S, S, S, c, c, 3, 1, 1, 1, True)
    ctx = K.StepContext(dev)
    with K.step_context(ctx):
        def wg():
            K.conv_bwd_weight(g, x, dy, dw, None, 0, None, False); ctx.final.clear()
        print("dw wgrad C=%d %d^3: %.1f us" % (c, S, timeit(wg)))
