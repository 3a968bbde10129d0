"""Normalise-on-load probe (round 4): what does a node hand-off cost the dependent chain
   A: node epilogue launch (n3d_affine_act2: both raw terms -> node) + the next conv reading the node     [today]
   B: the next conv reading the two raw terms and normalising in its LDS tile (n3d_conv_fwd_nol)          [epilogue off the chain]
at the C = 4 levels (up-cell 4: 2 x 64^3; 2 x 128^3 for the large patches), dilation 1 and 2?  HIP-graph replay of dependent
launches + HIP events; results are checked bit for bit first."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K
from nas_3d_unet_amd.train import capture_stream

dev = torch.device("cuda", 0)


def timed(fn, iters=40, reps=5):
    s = capture_stream(dev)
    g = torch.cuda.CUDAGraph()
    fn()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(iters):
                fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


def case(size, dil, batch=2, planar=True):
    c = 4
    mk = lambda: K.as_view(K.empty_ndhwc(batch, c, size, size, size, dev).normal_())
    raw0, raw1, y = mk(), mk(), mk()
    if planar:
        node = mk()                                    # dense node (the last cell's layout since round 4)
    else:
        buf = K.as_view(K.empty_ndhwc(batch, 3 * c, size, size, size, dev))
        node = K.View(buf.t[:, c:2 * c], buf.ld)       # a channel slice of the concatenation buffer (inner cells)
    w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.1
    b = torch.randn(c, device=dev) * 0.1
    g = K.conv_geom(batch, size, size, size, c, c, 3, 1, dil, dil)
    rows = K.conv_stats_rows(g, False)
    stats = torch.empty((batch, rows, c, 2), dtype=torch.float64, device=dev)
    a0, b0, a1, b1 = (torch.randn(batch, c, device=dev) for _ in range(4))
    coef = (a0, b0, a1, b1)
    terms = [(raw0, None, 0, None, None, None, True), (raw1, None, 0, None, None, None, True)]

    def act2():
        from nas_3d_unet_amd import _lib
        import ctypes as C
        t = []
        for raw, aa, bb in ((raw0, a0, b0), (raw1, a1, b1)):
            t.append(_lib.GnFwdTerm(raw.p.value, raw.ld, None, 0, 1, None, None, None, aa.data_ptr(), bb.data_ptr(), None, None, 0, 0))
        _lib.check(_lib.load().n3d_affine_act2(C.byref(t[0]), C.byref(t[1]), node.p, node.ld, None, 0, batch, size ** 3, c, 0, K.stream_ptr()), "act2")

    ctx = K.StepContext(dev)
    with K.step_context(ctx):
        act2()
        K.conv_fwd(g, node, w, b, y, 0, None, stats, False)
        ctx.freeze()
        ctx.pack_all()
        act2()
        K.conv_fwd(g, node, w, b, y, 0, None, stats, False)
        ref = y.t.clone()
        K.conv_fwd_nol(g, raw0, raw1, coef, 3, w, b, y, stats)
        torch.cuda.synchronize()
        same = torch.equal(ref, y.t)
        t_conv = timed(lambda: K.conv_fwd(g, node, w, b, y, 0, None, stats, False))
        t_act = timed(act2)
        t_a = timed(lambda: (act2(), K.conv_fwd(g, node, w, b, y, 0, None, stats, False)))
        t_b = timed(lambda: K.conv_fwd_nol(g, raw0, raw1, coef, 3, w, b, y, stats))
    print("C=4 %3d^3 d=%d %-6s: bit-identical %s | conv %.2f us, epilogue %.2f us, A = epilogue -> conv %.2f us | B = conv normalising on load %.2f us | "
          "chain saving per hand-off %.2f us (conv +%.2f us for -%.2f us of epilogue)" %
          (size, dil, "dense" if planar else "slice", same, t_conv, t_act, t_a, t_b, t_a - t_b, t_b - t_conv, t_act), flush=True)
    return same


if __name__ == "__main__":
    ok = True
    for size in (64, 128):
        for dil in (1, 2):
            for planar in (True, False):
                ok &= case(size, dil, planar=planar)
    print("all bit-identical:", ok)
