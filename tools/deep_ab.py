"""Timing of the deep-level (C = 16 .. 64) 3x3x3 conv launches through the C ABI, one conv at a time: forward, data gradient, weight
gradient (partial slabs only: the fixed-order reduction is one batched launch per step), HIP-graph replay + HIP events.
N3D_LIB=<another libn3d.so> times another build.   python tools/deep_ab.py [C size stride dil] ..."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K, _lib
from conv_ab import timed

dev = torch.device("cuda", 0)
PEAK = 157.3


def case(c, size, stride, dil, batch=2):
    so = size // stride
    x = K.as_view(K.empty_ndhwc(batch, c, size, size, size, dev).normal_())
    y = K.as_view(K.empty_ndhwc(batch, c, so, so, so, dev).normal_())
    w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.1
    b = torch.randn(c, device=dev) * 0.1
    dw, db = torch.zeros_like(w), torch.zeros_like(b)
    g = K.conv_geom(batch, size, size, size, c, c, 3, stride, dil, dil)
    rows = K.conv_stats_rows(g, False, 0, x, y)
    stats = torch.empty((batch, max(rows, 1), c, 2), dtype=torch.float64, device=dev) if rows > 0 else None
    ctx = K.StepContext(dev)
    # DEEP_AB_MM=1: the bf16 configuration's form of these levels (N3D_MM_BF16: operands rounded to bf16 in registers)
    with K.step_context(ctx), K.storage(torch.float32, os.environ.get("DEEP_AB_MM") == "1"):
        K.conv_fwd(g, x, w, b, y, 0, None, stats, False)
        K.conv_bwd_data(g, y, w, x, 0, None, None, False)
        ctx.freeze()
        ctx.pack_all()
        tf = timed(lambda: K.conv_fwd(g, x, w, b, y, 0, None, stats, False))
        td = timed(lambda: K.conv_bwd_data(g, y, w, x, 0, None, None, False))

        def wg():
            K.conv_bwd_weight(g, x, y, dw, db, 0, None, False)
            ctx.final.clear(); ctx.keep.clear()
        tw = timed(wg)
    fl = 2.0 * batch * so ** 3 * c * c * 27
    fr = lambda t: fl / t / 1e6 / PEAK
    print("C=%d %d^3 s%d d%d B=%d: %6.1f MFLOP  fwd %6.2f us (%.3f)  dgrad %6.2f us (%.3f)  wgrad %6.2f us (%.3f)   of %.1f TF" %
          (c, size, stride, dil, batch, fl / 1e6, tf, fr(tf), td, fr(td), tw, fr(tw), PEAK), flush=True)


if __name__ == "__main__":
    print("lib:", _lib.LIB_PATH)
    args = [int(a) for a in sys.argv[1:]]
    cases = [tuple(args[i:i + 4]) for i in range(0, len(args), 4)] or [
        (16, 32, 1, 1), (16, 32, 1, 2), (16, 32, 2, 1), (16, 16, 1, 1), (16, 16, 2, 1),
        (32, 16, 1, 1), (32, 16, 1, 2), (32, 16, 2, 1), (32, 8, 1, 1), (32, 8, 2, 1),
        (64, 8, 1, 1), (64, 8, 1, 2), (64, 8, 2, 1), (64, 4, 1, 1), (64, 4, 2, 1), (64, 2, 1, 1)]
    for cs in cases:
        case(*cs)
