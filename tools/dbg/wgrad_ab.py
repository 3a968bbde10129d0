"""A/B: weight gradient of the C=4 / C=8 3x3x3 convs: MFMA vox_wgrad vs the generic VALU kernel (N3D_NO_MFMA)."""
import sys, os, ctypes as C
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

for (ci, co, size, dil) in [(4, 4, 64, 1), (4, 4, 64, 2), (4, 4, 128, 1), (8, 8, 32, 1), (8, 8, 32, 2), (8, 8, 16, 1), (8, 8, 64, 1)]:
    for dt in ("fp32", "bf16"):
        with K.storage(torch.bfloat16 if dt == "bf16" else torch.float32):
            x = K.as_view(K.empty_ndhwc(2, ci, size, size, size, dev)); x.t.normal_()
            dy = K.as_view(K.empty_ndhwc(2, co, size, size, size, dev)); dy.t.normal_()
        w = torch.empty(co, ci, 3, 3, 3, device=dev)
        dw = torch.empty_like(w)
        g = K.conv_geom(2, size, size, size, ci, co, 3, 1, dil, dil)
        res = []
        for fl in (0, _lib.NO_MFMA):
            try:
                us = timeit(lambda: K.conv_bwd_weight(g, x, dy, dw, None, fl, None, False))
            except Exception as e:
                us = float("nan"); print("  ", e)
            res.append(us)
        flop = 2.0 * 2 * size ** 3 * 27 * ci * co
        print("%2d->%2d %3d^3 d%d %s: default %.1f us (%.1f TF)  NO_MFMA %.1f us (%.1f TF)" % (ci, co, size, dil, dt, res[0], flop / res[0] / 1e6, res[1], flop / res[1] / 1e6))
