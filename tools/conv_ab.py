"""A/B timing of the C in {4, 8} 3x3x3 conv launches through the C ABI: forward and data gradient (accumulate into the
destination, as the backward walk of a cell issues it), HIP-graph replay + HIP events (kernel time + the dependent-launch
boundary).  N3D_LIB=<another libn3d.so> times another build.   python tools/conv_ab.py [C size dil batch] ..."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K, _lib
from nas_3d_unet_amd._lib import ACCUMULATE
from nas_3d_unet_amd.train import capture_stream

dev = torch.device("cuda", 0)
DT = torch.bfloat16 if os.environ.get("CONV_AB_DT") == "bf16" else torch.float32     # CONV_AB_DT=bf16: bf16-storage kernels (priced against HBM)


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


def case(c, size, dil, batch):
    x = K.as_view(K.empty_ndhwc(batch, c, size, size, size, dev, DT).normal_())
    y = K.as_view(K.empty_ndhwc(batch, c, size, size, size, dev, DT).normal_())
    w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.1
    b = torch.randn(c, device=dev) * 0.1
    g = K.conv_geom(batch, size, size, size, c, c, 3, 1, dil, dil)
    rows = K.conv_stats_rows(g, False, 0, x, y)
    stats = torch.empty((batch, rows, c, 2), dtype=torch.float64, device=dev)
    ctx = K.StepContext(dev)
    with K.step_context(ctx):
        K.conv_fwd(g, x, w, b, y, 0, None, stats, False)
        K.conv_bwd_data(g, y, w, x, ACCUMULATE, None, None, False)
        ctx.freeze()
        ctx.pack_all()
        tf = timed(lambda: K.conv_fwd(g, x, w, b, y, 0, None, stats, False))
        if os.environ.get("CONV_AB_EXTRA"):
            tns = timed(lambda: K.conv_fwd(g, x, w, b, y, 0, None, None, False))
            tnb = timed(lambda: K.conv_fwd(g, x, w, None, y, 0, None, None, False))
            print("   fwd %.2f us, without statistics %.2f us, without statistics and bias %.2f us" % (tf, tns, tnb), flush=True)
        td = timed(lambda: K.conv_bwd_data(g, y, w, x, ACCUMULATE, None, None, False))
        tn = timed(lambda: K.conv_bwd_data(g, y, w, x, 0, None, None, False))
    fl = 2.0 * batch * size ** 3 * c * c * 27
    if DT == torch.bfloat16:
        by = 2.0 * batch * size ** 3 * c * 2      # input + output tensor, once each
        print("bf16 C=%d %d^3 d=%d B=%d: fwd %.2f us (%.3f of 8 TB/s)  dgrad+acc %.2f us (%.3f, 3 passes)  dgrad %.2f us (%.3f)" %
              (c, size, dil, batch, tf, by / tf / 1e3 / 8000, td, 1.5 * by / td / 1e3 / 8000, tn, by / tn / 1e3 / 8000), flush=True)
        return
    print("C=%d %d^3 d=%d B=%d: fwd %.2f us (%.3f of 157.3 TF)  dgrad+acc %.2f us (%.3f)  dgrad %.2f us" %
          (c, size, dil, batch, tf, fl / tf / 1e6 / 157.3, td, fl / td / 1e6 / 157.3, tn), flush=True)


if __name__ == "__main__":
    print("lib:", _lib.LIB_PATH)
    args = [int(a) for a in sys.argv[1:]]
    cases = [tuple(args[i:i + 4]) for i in range(0, len(args), 4)] or [(4, 64, 1, 2), (4, 64, 2, 2), (8, 32, 1, 2), (8, 32, 2, 2), (4, 128, 1, 2)]
    for cs in cases:
        case(*cs)
