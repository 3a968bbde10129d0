"""vox_wgrad: stages in flight (N3D_VW_NS) x workgroup target (N3D_VW_WGS); one process per setting (the knobs are read once)"""
import sys, os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, R)
    import torch
    from nas_3d_unet_amd import kernels as K
    dev = torch.device("cuda")
    def timeit(fn, reps=20, rounds=5):
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
    out = []
    for (c, size, dil, dt) in [(4, 64, 1, "fp32"), (4, 64, 2, "fp32"), (8, 32, 1, "fp32"), (8, 32, 2, "fp32"), (4, 128, 1, "fp32"), (8, 64, 1, "fp32"), (4, 128, 1, "bf16"), (8, 64, 1, "bf16")]:
        with K.storage(torch.bfloat16 if dt == "bf16" else torch.float32):
            x = K.as_view(K.empty_ndhwc(2, c, size, size, size, dev)); x.t.normal_()
            dy = K.as_view(K.empty_ndhwc(2, c, size, size, size, dev)); dy.t.normal_()
        dw = torch.empty(c, c, 3, 3, 3, device=dev)
        g = K.conv_geom(2, size, size, size, c, c, 3, 1, dil, dil)
        if os.environ.get("WG_DEFER"):
            ctx = K.StepContext(dev)     # the slab reduction is deferred (never run here): the weight-gradient kernel alone
            def fn():
                with K.step_context(ctx):
                    K.conv_bwd_weight(g, x, dy, dw, None, 0, None, False)
                del ctx.final[:]; del ctx.keep[:]
        else:
            fn = lambda: K.conv_bwd_weight(g, x, dy, dw, None, 0, None, False)
        out.append("%d@%d^3d%d%s %.1f" % (c, size, dil, "b" if dt == "bf16" else "", timeit(fn)))
    print("NS=%s WGS=%s: " % (os.environ.get("N3D_VW_NS", "auto"), os.environ.get("N3D_VW_WGS", "384")) + "  ".join(out), flush=True)
else:
    for ns in ("0", "1", "2", "4", None):
        for wgs in ("384", "768", "1536"):
            env = dict(os.environ, N3D_VW_WGS=wgs)
            if ns is not None: env["N3D_VW_NS"] = ns
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
