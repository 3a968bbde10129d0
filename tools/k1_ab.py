"""Timing of the 1x1x1 streaming conv launches (conv_k1_kernel) through the C ABI: forward with statistics and data gradient with its
ReLU mask source (the form the backward walk issues), HIP-graph replay + HIP events, priced against 8 TB/s.
N3D_LIB=<another libn3d.so> times another build.   python tools/k1_ab.py [Ci Co size] ...     K1_AB_DT=bf16: bf16 storage"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K, _lib
from nas_3d_unet_amd._lib import ACCUMULATE
from conv_ab import timed

dev = torch.device("cuda", 0)
DT = torch.bfloat16 if os.environ.get("K1_AB_DT") == "bf16" else torch.float32
ES = 2 if DT == torch.bfloat16 else 4


def case(ci, co, size, batch=2):
    x = K.as_view(K.empty_ndhwc(batch, ci, size, size, size, dev, DT).normal_())
    dx = K.as_view(K.empty_ndhwc(batch, ci, size, size, size, dev, DT).normal_())
    y = K.as_view(K.empty_ndhwc(batch, co, size, size, size, dev, DT).normal_())
    w = torch.randn(co, ci, 1, 1, 1, device=dev) * 0.1
    b = torch.randn(co, device=dev) * 0.1
    g = K.conv_geom(batch, size, size, size, ci, co, 1, 1, 1, 0)
    rows = K.conv_stats_rows(g, False, 0, x, y)
    stats = torch.empty((batch, max(rows, 1), co, 2), dtype=torch.float64, device=dev)
    ctx = K.StepContext(dev)
    nv = batch * size ** 3
    with K.step_context(ctx):
        K.conv_fwd(g, x, w, b, y, 0, None, stats, False)
        K.conv_bwd_data(g, y, w, dx, 0, x, None, False)
        ctx.freeze()
        ctx.pack_all()
        tf = timed(lambda: K.conv_fwd(g, x, w, b, y, 0, None, stats, False))
        td = timed(lambda: K.conv_bwd_data(g, y, w, dx, 0, x, None, False))
        ta = timed(lambda: K.conv_bwd_data(g, y, w, dx, ACCUMULATE, x, None, False))
    bf, bd, ba = nv * (ci + co) * ES, nv * (co + 2 * ci) * ES, nv * (co + 3 * ci) * ES
    fr = lambda by, t: by / t / 1e3 / 8000
    print("%s %d->%d %d^3 B=%d: fwd %.2f us (%.3f of 8 TB/s)  dgrad+mask %.2f us (%.3f)  dgrad+mask+acc %.2f us (%.3f)" %
          ("bf16" if ES == 2 else "f32", ci, co, size, batch, tf, fr(bf, tf), td, fr(bd, td), ta, fr(ba, ta)), flush=True)


if __name__ == "__main__":
    print("lib:", _lib.LIB_PATH)
    args = [int(a) for a in sys.argv[1:]]
    cases = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(12, 4, 128), (12, 8, 64), (4, 12, 128), (24, 4, 64), (8, 4, 64)]
    for cs in cases:
        case(*cs)
