"""GPU: the two batched helper launches of a step through the C ABI, bit-exact against numpy restatements of their contracts
(include/n3d.h): n3d_pack_batch (native (Co, Ci, taps) weights -> the packed layouts 0..5 of the conv kernels) and
n3d_wgrad_finalize_batch (fixed-order sum of the weight-gradient partial slabs, scattered to the native layout).
The reference has no counterpart (torch packs / reduces inside cuDNN / ATen); what is checked is that every layout, the
channel tiles that do not fill a 16 x 16 tile, every chunk-count regime and the multi-launch grouping move each element to
the documented place."""
import ctypes as C

import numpy as np
import pytest
import torch

from _util import dev

pytestmark = pytest.mark.gpu


def _pack_reference(w, layout, data_grad, cdp):
    """w: (Co, Ci, taps) float32 -> flat packed array (float32, or bfloat16 bit patterns for layouts 4 / 5)"""
    Co, Ci, taps = w.shape
    Cs, Cd = (Co, Ci) if data_grad else (Ci, Co)
    # ws[tap][cs][cd]
    ws = np.transpose(w, (2, 0, 1)) if data_grad else np.transpose(w, (2, 1, 0))
    if layout == 0:
        out = np.zeros((taps, Cs, cdp), np.float32)
        out[:, :, :Cd] = ws
        return out.reshape(-1)
    if layout == 1:   # [tap][cs/16][kk][cd][j], cs = 16*c16 + 4*kk + j
        v = ws.reshape(taps, Cs // 16, 4, 4, Cd)          # tap, c16, kk, j, cd
        return np.ascontiguousarray(np.transpose(v, (0, 1, 2, 4, 3))).reshape(-1)
    if data_grad and layout in (2, 4):
        ws = ws[::-1]
    out = np.ascontiguousarray(np.transpose(ws, (0, 2, 1))).reshape(-1)   # [tap][cd][cs]
    if layout >= 4:
        return torch.from_numpy(out.copy()).to(torch.bfloat16).view(torch.int16).numpy()
    return out


PACK_CASES = [
    # layout, Co, Ci, taps, cdp, data_grad
    (0, 12, 4, 27, 12, 0), (0, 12, 4, 27, 4, 1), (0, 24, 20, 27, 32, 0), (0, 40, 36, 27, 36, 1), (0, 8, 8, 27, 16, 0),
    (0, 16, 48, 1, 16, 0), (0, 16, 48, 1, 48, 1), (0, 4, 12, 8, 4, 0), (0, 3, 12, 1, 4, 0), (0, 64, 64, 27, 64, 0),
    (1, 16, 16, 27, 0, 0), (1, 32, 16, 27, 0, 1), (1, 64, 64, 27, 0, 0), (1, 64, 64, 27, 0, 1), (1, 48, 32, 27, 0, 0),
    (1, 32, 192, 1, 0, 0), (1, 64, 96, 1, 0, 1), (1, 16, 32, 8, 0, 0),
    (2, 4, 4, 27, 0, 0), (2, 4, 4, 27, 0, 1), (2, 8, 8, 27, 0, 0), (2, 8, 8, 27, 0, 1), (3, 8, 8, 27, 0, 1), (3, 4, 4, 27, 0, 0),
    (4, 4, 4, 27, 0, 0), (4, 4, 4, 27, 0, 1), (4, 8, 8, 27, 0, 1), (5, 8, 8, 27, 0, 1), (5, 4, 4, 27, 0, 0),
]


def _run_pack(cases, seed):
    from nas_3d_unet_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    jobs = (_lib.PackJob * len(cases))()
    keep, want = [], []
    for i, (lay, Co, Ci, taps, cdp, dg) in enumerate(cases):
        wn = rng.standard_normal((Co, Ci, taps)).astype(np.float32)
        ref = _pack_reference(wn, lay, bool(dg), cdp)
        w = dev(wn)
        n_f32 = ref.size if lay < 4 else (ref.size + 1) // 2
        dst = torch.full((n_f32 + 8,), 777.0, dtype=torch.float32, device="cuda")   # guard words behind the slot
        jobs[i] = _lib.PackJob(w.data_ptr(), dst.data_ptr(), Co, Ci, taps, dg, lay, cdp)
        keep.append((w, dst))
        want.append(ref)
    _lib.check(lib.n3d_pack_batch(jobs, len(cases), None), "n3d_pack_batch")
    torch.cuda.synchronize()
    for (lay, Co, Ci, taps, cdp, dg), (w, dst), ref in zip(cases, keep, want):
        what = "layout %d Co %d Ci %d taps %d cdp %d data_grad %d" % (lay, Co, Ci, taps, cdp, dg)
        if lay < 4:
            got = dst[:ref.size].cpu().numpy()
            tail = dst[ref.size:].cpu().numpy()
        else:
            got = dst.view(torch.int16)[:ref.size].cpu().numpy()
            tail = dst[(ref.size + 1) // 2:].cpu().numpy()
        assert np.array_equal(got, ref), what
        assert np.all(tail == 777.0), what + ": wrote behind the slot"


def test_pack_batch_every_layout_bit_exact():
    _run_pack(PACK_CASES, 1)


def test_pack_batch_many_jobs_several_launches():
    # more jobs than one argument block holds (160), in an order that interleaves big and small jobs
    cases = [PACK_CASES[(7 * i) % len(PACK_CASES)] for i in range(420)]
    _run_pack(cases, 2)


def _final_case(rng, nchunks, taps, tci, tco, ci_t, co_t, Ci, Co, with_dw=True, with_db=True):
    ntiles = taps * tci * tco
    T = ci_t * co_t
    nb = tco * co_t
    partial = rng.standard_normal((nchunks, ntiles, T)).astype(np.float32)
    pbias = rng.standard_normal((nchunks, nb)).astype(np.float32)
    s = partial.astype(np.float64).sum(0).reshape(taps, tci, tco, ci_t, co_t)
    full = np.transpose(s, (2, 4, 1, 3, 0)).reshape(tco * co_t, tci * ci_t, taps)     # co, ci, tap
    dw = full[:Co, :Ci]
    db = pbias.astype(np.float64).sum(0)[:Co]
    return dict(partial=partial, pbias=pbias, dw=dw, db=db, nchunks=nchunks, ntiles=ntiles, tci=tci, tco=tco, ci_t=ci_t, co_t=co_t, Ci=Ci, Co=Co,
                taps=taps, with_dw=with_dw, with_db=with_db)


def _run_final(cases):
    from nas_3d_unet_amd import _lib
    lib = _lib.load()
    jobs = (_lib.FinalJob * len(cases))()
    keep = []
    for i, c in enumerate(cases):
        partial, pbias = dev(c["partial"]), dev(c["pbias"])
        dw = torch.full((c["Co"] * c["Ci"] * c["taps"] + 8,), 555.0, dtype=torch.float32, device="cuda")
        db = torch.full((c["Co"] + 8,), 555.0, dtype=torch.float32, device="cuda")
        jobs[i] = _lib.FinalJob(partial.data_ptr(), pbias.data_ptr() if c["with_db"] else None, dw.data_ptr() if c["with_dw"] else None,
                                db.data_ptr() if c["with_db"] else None, c["nchunks"], c["ntiles"], c["tci"], c["tco"], c["ci_t"], c["co_t"],
                                c["Co"], c["Ci"], c["taps"], 0)
        keep.append((partial, pbias, dw, db))
    _lib.check(lib.n3d_wgrad_finalize_batch(jobs, len(cases), None), "n3d_wgrad_finalize_batch")
    torch.cuda.synchronize()
    for c, (_, _, dw, db) in zip(cases, keep):
        what = "chunks %d taps %d tiles %dx%d of %dx%d, Ci %d Co %d" % (c["nchunks"], c["taps"], c["tci"], c["tco"], c["ci_t"], c["co_t"], c["Ci"], c["Co"])
        n = c["Co"] * c["Ci"] * c["taps"]
        gw, gb = dw.cpu().numpy(), db.cpu().numpy()
        scale = np.sqrt(c["nchunks"])
        if c["with_dw"]:
            assert np.abs(gw[:n].reshape(c["Co"], c["Ci"], c["taps"]) - c["dw"]).max() <= 2e-6 * scale * 4, what
        else:
            assert np.all(gw == 555.0), what
        assert np.all(gw[n:] == 555.0), what + ": wrote behind dw"
        if c["with_db"]:
            assert np.abs(gb[:c["Co"]] - c["db"]).max() <= 2e-6 * scale * 4, what
        assert np.all(gb[c["Co"]:] == 555.0), what + ": wrote behind dbias"


def test_wgrad_finalize_batch_every_regime():
    rng = np.random.default_rng(5)
    cases = []
    # (ci tile, co tile) transposing path: <= 4 chunks, tile <= 256 positions; clipped channel counts; 1 tap and 27 taps
    for nch in (1, 2, 3, 4):
        cases.append(_final_case(rng, nch, 27, 4, 4, 16, 16, 64, 64))
        cases.append(_final_case(rng, nch, 27, 3, 2, 8, 16, 20, 24))
        cases.append(_final_case(rng, nch, 27, 1, 1, 4, 4, 4, 4))
        cases.append(_final_case(rng, nch, 1, 1, 1, 12, 3, 12, 3))
        cases.append(_final_case(rng, nch, 27, 1, 1, 1, 8, 1, 8))           # depthwise form
        cases.append(_final_case(rng, nch, 8, 2, 3, 4, 8, 7, 20))
    # few chunks, tile larger than a workgroup (1x1x1 convs: the tile is the whole matrix)
    cases.append(_final_case(rng, 3, 1, 1, 1, 192, 64, 192, 64))
    cases.append(_final_case(rng, 12, 1, 1, 1, 48, 32, 48, 32))
    # 5 .. 16 chunks: one thread per slab position
    for nch in (5, 10, 16):
        cases.append(_final_case(rng, nch, 27, 2, 2, 16, 16, 32, 32))
        cases.append(_final_case(rng, nch, 27, 1, 1, 8, 8, 8, 8))
    # many chunks: positions x chunk segments, every split
    for nch in (17, 37, 64, 65, 128, 256, 257, 586, 1024, 1500):
        cases.append(_final_case(rng, nch, 27, 1, 1, 4, 4, 4, 4))
        cases.append(_final_case(rng, nch, 1, 1, 1, 12, 3, 12, 3))
    cases.append(_final_case(rng, 40, 27, 1, 1, 8, 8, 8, 8, with_db=False))
    cases.append(_final_case(rng, 3, 27, 1, 2, 16, 16, 16, 32, with_dw=False))
    cases.append(_final_case(rng, 20, 27, 1, 1, 4, 4, 4, 4, with_dw=False))
    _run_final(cases)


def test_wgrad_finalize_batch_many_jobs_several_launches():
    rng = np.random.default_rng(6)
    shapes = [(2, 27, 2, 2, 16, 16, 32, 32), (64, 27, 1, 1, 4, 4, 4, 4), (9, 27, 1, 1, 8, 8, 8, 8), (300, 1, 1, 1, 12, 4, 12, 4), (1, 1, 1, 1, 24, 16, 24, 16)]
    cases = [_final_case(rng, *shapes[(3 * i) % len(shapes)]) for i in range(210)]
    _run_final(cases)


def test_wgrad_finalize_one_job_larger_than_the_workgroup_map():
    # 20 chunks of a 600k-position slab: more workgroups than the byte map covers -> the job travels alone
    rng = np.random.default_rng(7)
    _run_final([_final_case(rng, 20, 27, 12, 12, 16, 16, 192, 192), _final_case(rng, 2, 27, 1, 1, 4, 4, 4, 4)])
