"""round-6 probe for the GroupNorm coefficient launches at the large levels (review item 4): the node epilogue of two terms as the shipped
two launches (n3d_gn_coeffs2 over R partial rows + n3d_affine_act2) against the fused one-launch form (n3d_affine_act_gn2, per-wave
prologue) fed with rows already folded to 16 -- what a two-level reduction would leave to the consumer.   python tools/gn_fold_probe.py [C S rows] ..."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K
from conv_ab import timed
dev = torch.device("cuda", 0)


def case(c, s, rows, b=2, dt=torch.float32):
    mk = lambda: K.as_view(K.empty_ndhwc(b, c, s, s, s, dev, dt).normal_())
    raw0, raw1, out = mk(), mk(), mk()
    gam, bet = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    N = s ** 3

    def stats(r):
        st = torch.rand((b, r, c, 2), dtype=torch.float64, device=dev)
        st[..., 1] += N / r * 2.0
        return st
    big, small = [stats(rows), stats(rows)], [stats(16), stats(16)]

    def run(sts, r):
        K.affine_act_gn2([(raw0, sts[0], r, gam, bet, None, True), (raw1, sts[1], r, gam, bet, None, True)], 1, 1e-5, out)
    t_two = timed(lambda: run(big, rows))
    t_fused = timed(lambda: run(small, 16))
    print("%s C=%d %d^3 B=%d: coefficients over %d rows + apply (two launches) %.2f us;  fused apply with a 16-row prologue (one launch) %.2f us"
          % ("bf16" if dt == torch.bfloat16 else "f32", c, s, b, rows, t_two, t_fused), flush=True)


if __name__ == "__main__":
    args = [int(x) for x in sys.argv[1:]]
    cases = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(4, 128, 8192), (8, 64, 2048), (4, 64, 1024), (8, 32, 512)]
    for cs in cases:
        case(*cs)
        case(*cs, dt=torch.bfloat16)


def bwd_case(c, s, rows, b=2):
    """backward: n3d_gn_bwd_coeffs2 over `rows` partial rows + n3d_affine_act_bwd_apply2 (two launches) against the fused
    n3d_affine_act_bwd_apply_gn2 with a 16-row prologue (one launch)"""
    import ctypes as C
    from nas_3d_unet_amd import _lib
    lib = _lib.load()
    mk = lambda: K.as_view(K.empty_ndhwc(b, c, s, s, s, dev).normal_())
    raw0, raw1, d0, d1, dnode = mk(), mk(), mk(), mk(), mk()
    N = s ** 3
    f = lambda *sh: torch.rand(*sh, device=dev) + 0.5
    a, bb, gam, mr = f(b, c), f(b, c), f(c), f(b, 1, 2)
    coef = torch.empty((2, 3, b, c), device=dev)
    dg, db = torch.empty((2, c), device=dev), torch.empty((2, c), device=dev)

    def terms(sums, r, fused):
        out = []
        for i, (rw, dr) in enumerate(((raw0, d0), (raw1, d1))):
            cc = [None] * 3 if fused else [coef[i, j].data_ptr() for j in range(3)]
            out.append(_lib.GnBwdTerm(rw.p.value, rw.ld, a.data_ptr(), bb.data_ptr(), sums[i].data_ptr(), r, 1, gam.data_ptr(), mr.data_ptr(), None, None,
                                      dr.p.value, dr.ld, dg[i].data_ptr(), db[i].data_ptr(), None, None, *cc, 0, 0))
        return out
    big = torch.rand((2, b, rows, c, 3), dtype=torch.float64, device=dev)
    small = torch.rand((2, b, 16, c, 3), dtype=torch.float64, device=dev)

    def two():
        t = terms(big, rows, False)
        _lib.check(lib.n3d_gn_bwd_coeffs2(C.byref(t[0]), C.byref(t[1]), b, c, 1, N, K.stream_ptr()), "bwd_coeffs2")
        _lib.check(lib.n3d_affine_act_bwd_apply2(dnode.p, dnode.ld, None, 0, C.byref(t[0]), C.byref(t[1]), b, N, c, K.stream_ptr()), "apply2")

    def one():
        t = terms(small, 16, True)
        _lib.check(lib.n3d_affine_act_bwd_apply_gn2(dnode.p, dnode.ld, None, 0, C.byref(t[0]), C.byref(t[1]), b, N, c, 1, K.stream_ptr()), "apply_gn2")
    print("f32 C=%d %d^3 B=%d backward: coefficients over %d rows + apply (two launches) %.2f us;  fused apply with a 16-row prologue %.2f us"
          % (c, s, b, rows, timed(two), timed(one)), flush=True)


if __name__ == "__main__":
    for cs in [(4, 128, 1024), (8, 64, 1024), (4, 64, 1024), (8, 32, 128)]:
        bwd_case(*cs)
