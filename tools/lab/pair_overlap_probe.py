"""How well do the two convs of a searched-cell node overlap when they run at the same time?  Upper bound for a one-launch pair kernel:
the two launches of n3d_conv_fwd2 back to back on one stream against the same two launches on two streams (graph-free, many repetitions
queued so that the launch overhead of the host is hidden).  usage: pair_overlap_probe.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K
from nas_3d_unet_amd.train import reserve_side_streams
dev = torch.device("cuda", 0)

def mk(c, size, stride, dil, transposed=False):
    so = size // stride
    x = K.as_view(K.empty_ndhwc(2, c, size if not transposed else so, size if not transposed else so, size if not transposed else so, dev).normal_())
    y = K.as_view(K.empty_ndhwc(2, c, so if not transposed else size, so if not transposed else size, so if not transposed else size, dev).normal_())
    w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.1
    b = torch.randn(c, device=dev) * 0.1
    g = K.conv_geom(2, size, size, size, c, c, 3, stride, dil, dil)
    rows = K.conv_stats_rows(g, transposed)
    st = torch.empty((2, rows, c, 2), dtype=torch.float64, device=dev) if rows > 0 else None
    return (g, x, w, b, y, 0, None, st, transposed)

def run(cases, label):
    ctx = K.StepContext(dev)
    with K.step_context(ctx):
        for c in cases: K.conv_fwd(*c)
        ctx.freeze(); ctx.pack_all()
        s1, s2 = reserve_side_streams(dev, 2)[:2]
        torch.cuda.synchronize()
        n = 100
        def cap(stream, fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(stream):
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    fn()
                finally:
                    g.capture_end()
            return g
        def loop(cs):
            def f():
                for _ in range(n):
                    for c in cs: K.conv_fwd(*c)
            return f
        from nas_3d_unet_amd.train import capture_stream
        cs_ = capture_stream(dev)
        g_seq = cap(cs_, loop(cases))
        g_a, g_b = cap(cs_, loop(cases[:1])), cap(cs_, loop(cases[1:]))
        t1, t2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        def t_seq():
            with torch.cuda.stream(t1): g_seq.replay()
        def t_par():
            with torch.cuda.stream(t1): g_a.replay()
            with torch.cuda.stream(t2): g_b.replay()
        def t_a():
            with torch.cuda.stream(t1): g_a.replay()
        def t_b():
            with torch.cuda.stream(t1): g_b.replay()
        out = []
        for fn in (t_seq, t_par, t_a, t_b):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t1.wait_event(e0); t2.wait_event(e0)
            for _ in range(3): fn()
            ea, eb = torch.cuda.Event(), torch.cuda.Event()
            ea.record(t1); eb.record(t2)
            torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
            e1.record(); torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / (3 * n))
        print("%-52s back to back %.2f us per pair; two chains side by side %.2f; alone: %.2f and %.2f" % (label, out[0], out[1], out[2], out[3]), flush=True)

run([mk(4, 64, 1, 1), mk(4, 64, 2, 1)], "C=4 64^3: conv s1 d1 + down conv s2 d1")
run([mk(4, 64, 1, 1), mk(4, 64, 1, 2)], "C=4 64^3: conv s1 d1 + dil conv s1 d2")
run([mk(4, 64, 1, 1), mk(4, 64, 2, 1, True)], "C=4 64^3: conv s1 d1 + up conv (32^3 -> 64^3)")
run([mk(8, 32, 1, 1), mk(8, 32, 2, 1)], "C=8 32^3: conv s1 d1 + down conv s2 d1")
run([mk(8, 32, 1, 1), mk(8, 32, 1, 2)], "C=8 32^3: conv s1 d1 + dil conv s1 d2")
