"""Deep-level conv backward (C >= 16): data gradient alone, weight gradient alone, both in one launch, and the pair of a node."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K, _lib
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

def case(c, size, stride, dil):
    g = K.conv_geom(2, size, size, size, c, c, 3, stride, dil, dil)
    x = K.as_view(K.empty_ndhwc(2, c, size, size, size, dev)); x.t.normal_()
    dy = K.as_view(K.empty_ndhwc(2, c, g.Do, g.Ho, g.Wo, dev)); dy.t.normal_()
    dx = K.as_view(K.empty_ndhwc(2, c, size, size, size, dev))
    w = torch.randn(c, c, 3, 3, 3, device=dev)
    dw = torch.empty_like(w)
    return g, x, dy, dx, w, dw

ctx = K.StepContext(dev)     # as in a training step: the slab reductions are deferred (not launched here)
K._ctx = ctx
for (c, size) in [(16, 16), (32, 8), (64, 4), (16, 8)]:
    A = case(c, size, 1, 2)
    Bc = case(c, size, 2, 1)
    for name, (g, x, dy, dx, w, dw) in (("s1 d2", A), ("s2 d1", Bc)):
        td = timeit(lambda: K.conv_bwd_data(g, dy, w, dx))
        tw = timeit(lambda: K.conv_bwd_weight(g, x, dy, dw, None, 0, None, False))
        tb = timeit(lambda: K.conv_bwd_both(g, x, dy, w, dx, dw, None))
        print("C=%d %d^3 %s: data %.2f us   weight %.2f us   both %.2f us" % (c, size, name, td, tw, tb))
    def pair():
        K.conv_bwd_both2([(A[0], A[1], A[2], A[4], A[3], A[5], None, 0, None, None, 0, None, False),
                          (Bc[0], Bc[1], Bc[2], Bc[4], Bc[3], Bc[5], None, 0, None, None, 0, None, False)])
    print("C=%d %d^3 pair (both2): %.2f us" % (c, size, timeit(pair)))
