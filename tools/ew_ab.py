"""Timing of the three node-epilogue streaming launches of a searched-cell node (n3d_affine_act2, n3d_affine_act_bwd_reduce2,
n3d_affine_act_bwd_apply2) at (B, C, S^3) on dense tensors, HIP-graph replay + HIP events, priced against 8 TB/s.
N3D_LIB=<another libn3d.so> times another build.   python tools/ew_ab.py [C S B] ...     EW_AB_DT=bf16: bf16 storage"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import ctypes as C
import torch
from nas_3d_unet_amd import kernels as K, _lib
from conv_ab import timed

dev = torch.device("cuda", 0)
BF = os.environ.get("EW_AB_DT") == "bf16"
DT = torch.bfloat16 if BF else torch.float32
ES = 2 if BF else 4
DTC = _lib.BF16 if BF else 0


def case(c, s, b):
    mk = lambda: K.as_view(K.empty_ndhwc(b, c, s, s, s, dev, DT).normal_())
    raw0, raw1, draw0, draw1, node, dnode = mk(), mk(), mk(), mk(), mk(), mk()
    N = s ** 3
    lib = _lib.load()
    a = [torch.randn(b, c, device=dev) for _ in range(10)]
    rows = K.stats_rows(N, c)
    sums = [torch.empty((b, rows, c, 3), dtype=torch.float64, device=dev) for _ in range(2)]

    def fwd_terms():
        return [_lib.GnFwdTerm(r.p.value, r.ld, None, 0, 1, None, None, None, a[2 * i].data_ptr(), a[2 * i + 1].data_ptr(), None, None, DTC, 0)
                for i, r in enumerate((raw0, raw1))]

    def bwd_terms():
        return [_lib.GnBwdTerm(r.p.value, r.ld, a[2 * i].data_ptr(), a[2 * i + 1].data_ptr(), sums[i].data_ptr(), rows, 1, None, None, None, None,
                               d.p.value, d.ld, None, None, None, None, a[4 + 3 * i].data_ptr(), a[5 + 3 * i].data_ptr(), a[6 + 3 * i].data_ptr(), DTC, 0)
                for i, (r, d) in enumerate(((raw0, draw0), (raw1, draw1)))]

    def f_act():
        t = fwd_terms()
        _lib.check(lib.n3d_affine_act2(C.byref(t[0]), C.byref(t[1]), node.p, node.ld, None, 0, b, N, c, _lib.ACT_BF16 if BF else 0, K.stream_ptr()), "act2")

    def f_red():
        t = bwd_terms()
        _lib.check(lib.n3d_affine_act_bwd_reduce2(dnode.p, dnode.ld, None, 0, C.byref(t[0]), C.byref(t[1]), b, N, c, K.stream_ptr()), "reduce2")

    def f_app():
        t = bwd_terms()
        _lib.check(lib.n3d_affine_act_bwd_apply2(dnode.p, dnode.ld, None, 0, C.byref(t[0]), C.byref(t[1]), b, N, c, K.stream_ptr()), "apply2")

    one = b * N * c * ES
    ta, tr, tp = timed(f_act), timed(f_red), timed(f_app)
    fr = lambda passes, t: passes * one / t / 1e3 / 8000
    print("%s C=%d %d^3 B=%d: act2 %.2f us (%.3f of 8 TB/s, 3 passes)  bwd_reduce2 %.2f us (%.3f, 3 passes)  bwd_apply2 %.2f us (%.3f, 5 passes)" %
          ("bf16" if BF else "f32", c, s, b, ta, fr(3, ta), tr, fr(3, tr), tp, fr(5, tp)), flush=True)


if __name__ == "__main__":
    print("lib:", _lib.LIB_PATH)
    args = [int(x) for x in sys.argv[1:]]
    cases = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(4, 128, 2), (8, 64, 2), (4, 64, 2)]
    for cs in cases:
        case(*cs)
