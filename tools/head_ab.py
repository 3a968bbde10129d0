"""Timing of the two head passes (n3d_head_fwd with its Dice sums, n3d_head_bwd with the fused Dice gradient) with float and with
byte targets, HIP-graph replay + HIP events, priced against 8 TB/s.   python tools/head_ab.py [size ...]     HEAD_AB_DT=bf16: bf16 input"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K, _lib
from conv_ab import timed

dev = torch.device("cuda", 0)
DT = torch.bfloat16 if os.environ.get("HEAD_AB_DT") == "bf16" else torch.float32
ES = 2 if DT == torch.bfloat16 else 4


def case(size, batch=2, ci=12, co=3):
    x = K.as_view(K.empty_ndhwc(batch, ci, size, size, size, dev, DT).normal_(), bf16_ok=True)
    dx = K.as_view(K.empty_ndhwc(batch, ci, size, size, size, dev, DT), bf16_ok=True)
    w = torch.randn(co, ci, 1, 1, 1, device=dev) * 0.3
    b = torch.randn(co, device=dev) * 0.1
    tb = torch.rand(batch, co, size, size, size, device=dev) < 0.3
    dw, db = torch.empty_like(w), torch.empty_like(b)
    nv = batch * size ** 3
    for name, t, tes in (("float targets", tb.float(), 4), ("byte targets", tb.to(torch.uint8), 1)):
        _, _, sums, _ = K.head_fwd(x, w, b, None, t, want_p=False)
        tf = timed(lambda: K.head_fwd(x, w, b, None, t, want_p=False))
        tg = timed(lambda: K.head_bwd(x, w, b, None, dx, dw, db, t=t, sums=sums))
        bf, bg = nv * (ci * ES + co * tes), nv * (2 * ci * ES + co * tes)
        print("%s %d->%d %d^3 B=%d, %s: head_fwd (+ Dice sums launch) %.2f us (%.3f of 8 TB/s)  head_bwd %.2f us (%.3f)" %
              ("bf16" if ES == 2 else "f32", ci, co, size, batch, name, tf, bf / tf / 1e3 / 8000, tg, bg / tg / 1e3 / 8000), flush=True)


if __name__ == "__main__":
    print("lib:", _lib.LIB_PATH)
    for s in [int(a) for a in sys.argv[1:]] or [64, 128]:
        case(s)
