"""depthwise 3x3x3 weight gradient alone (the supernet's depthwise-separable primitives): us per launch, slab reduction deferred"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K
dev = torch.device("cuda")
def timeit(fn, reps=20, rounds=5):
    st = torch.cuda.Stream(device=dev); g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
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
out = []
for (c, size, stride) in [(4, 64, 1), (8, 32, 1), (16, 16, 1), (32, 8, 1), (64, 4, 1), (8, 32, 2), (16, 16, 2)]:
    x = K.as_view(K.empty_ndhwc(2, c, size, size, size, dev)); x.t.normal_()
    so = (size - 1) // stride + 1
    dy = K.as_view(K.empty_ndhwc(2, c, so, so, so, dev)); dy.t.normal_()
    dw = torch.empty(c, 1, 3, 3, 3, device=dev)
    g = K.conv_geom(2, size, size, size, c, c, 3, stride, 1, 1, depthwise=True)
    ctx = K.StepContext(dev)
    def fn():
        with K.step_context(ctx):
            K.conv_bwd_weight(g, x, dy, dw, None, 0, None, False)
        del ctx.final[:]; del ctx.keep[:]
    out.append("%d@%d^3s%d %.1f" % (c, size, stride, timeit(fn)))
print("  ".join(out))
