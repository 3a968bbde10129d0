"""GPU parity of the convolution family called straight through the C ABI wrappers (kernels.py),
against torch-CPU F.conv3d / F.conv_transpose3d (the arithmetic the reference delegates to).
Sweeps channel counts, strides, dilations, transposed, odd sizes; covers the fused extras
(ReLU-on-load, input gate, GN statistics epilogue, accumulate, ReLU-mask / gate on the data gradient)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import assert_close

pytestmark = pytest.mark.gpu

# (Cin, Cout, k, stride, dil, transposed, B, spatial of the conv INPUT tensor)
CASES = [
    (4, 4, 3, 1, 1, False, 2, (8, 10, 12)),
    (4, 4, 3, 1, 2, False, 1, (8, 8, 8)),
    (4, 12, 3, 2, 1, False, 2, (8, 8, 12)),
    (8, 8, 3, 1, 1, False, 2, (6, 8, 10)),
    (8, 8, 3, 2, 2, False, 2, (8, 8, 8)),
    (8, 8, 3, 2, 1, True, 2, (4, 4, 6)),
    (8, 8, 3, 2, 2, True, 2, (4, 4, 6)),
    (16, 16, 3, 1, 1, False, 2, (8, 8, 8)),      # gemm16, KSPLIT=4
    (16, 16, 3, 1, 2, False, 2, (24, 24, 24)),   # gemm16, KSPLIT=1 (tiles > 1024)
    # tile16 (LDS halo tile + 16x16x4 MFMA): the gemm16 problem with many voxels (> 16k), W % 16 == 0, H % 4 == 0, D % 2 == 0
    (16, 16, 3, 1, 1, False, 2, (16, 20, 32)),
    (16, 16, 3, 1, 2, False, 1, (18, 24, 48)),
    (16, 16, 3, 1, 1, False, 2, (32, 32, 32)),   # the C = 16 level of a 128^3 patch (>= 256 tiles: LDS-tile weight gradient too)
    (16, 16, 3, 1, 2, False, 2, (16, 32, 32)),
    (16, 16, 3, 1, 1, False, 2, (16, 16, 16)),   # the C = 16 level of the benchmarked 64^3 patch: 64 tiles, the lower bound of the LDS-tile
    (16, 16, 3, 1, 2, False, 2, (16, 16, 16)),   # weight gradient (N3D_WGT16_MIN); data gradient / forward on the K-split plan (KSPLIT=4)
    (16, 16, 3, 1, 1, False, 3, (12, 32, 32)),   # 576 tiles over 256 workgroups: ragged tiles-per-workgroup, workgroups span samples
    (32, 32, 3, 1, 1, False, 2, (16, 16, 16)),   # 64 tiles x 4 channel tiles: the LDS-tile weight gradient on 32 channels
    (32, 32, 3, 1, 2, False, 2, (16, 16, 16)),
    # tile32 (round 6: LDS halo tile + weight columns, 1 x 4 x 16 voxels x 16 output channels per workgroup): 32 -> 32, stride 1, >= 128 tiles
    (32, 32, 3, 1, 1, False, 1, (16, 32, 32)),   # one sample, 256 tiles
    (32, 32, 3, 1, 2, False, 3, (6, 16, 32)),    # three samples, D not a multiple of anything: 144 tiles x 3
    (32, 32, 3, 1, 1, False, 1, (8, 16, 16)),    # 32 tiles: below the tile kernel's floor, the K-split plan
    (32, 16, 3, 1, 1, False, 2, (16, 16, 32)),   # Ci != Co
    (64, 64, 3, 1, 1, False, 2, (8, 8, 8)),      # the 8^3 level of 128^3 patches: LDS-tile weight gradient on 4 x 4 x 8 tiles (8 tiles x 16 channel tiles)
    (64, 64, 3, 1, 2, False, 2, (8, 8, 8)),
    (64, 32, 3, 1, 1, False, 3, (12, 8, 8)),     # ... ragged: 9 tiles x 8 channel tiles, workgroups span samples
    (16, 48, 3, 1, 1, False, 1, (16, 32, 32)),
    # tile16_up: transposed forward / stride-2 data gradient of the 16-channel level with >= 32k destination voxels
    (16, 16, 3, 2, 1, True, 2, (8, 16, 32)),
    (16, 16, 3, 2, 2, True, 2, (8, 16, 32)),
    (16, 16, 3, 2, 1, False, 2, (16, 32, 64)),
    (16, 16, 3, 2, 2, False, 1, (32, 32, 32)),
    (16, 16, 3, 2, 1, True, 3, (6, 8, 48)),
    (16, 16, 3, 2, 1, False, 2, (8, 8, 8)),
    (16, 16, 3, 2, 1, True, 2, (4, 4, 4)),
    (32, 32, 3, 1, 1, False, 2, (4, 4, 4)),
    (32, 32, 3, 2, 2, True, 3, (4, 2, 2)),
    (64, 64, 3, 1, 1, False, 2, (2, 2, 2)),      # 8 voxels per sample: a 16-row MFMA tile holds two samples, one statistics row each
    (64, 64, 3, 1, 1, False, 3, (2, 2, 2)),      # ... odd batch: the last tile holds one sample
    (32, 32, 3, 1, 2, False, 1, (2, 2, 2)),
    (64, 64, 3, 2, 1, False, 2, (4, 4, 4)),
    (64, 64, 3, 2, 1, True, 2, (2, 2, 2)),
    (12, 8, 1, 2, 1, False, 2, (8, 8, 8)),
    (24, 16, 1, 1, 1, False, 2, (4, 6, 8)),
    (48, 16, 1, 1, 1, False, 2, (8, 8, 8)),
    (192, 64, 1, 1, 1, False, 2, (2, 2, 2)),
    (96, 32, 1, 2, 1, False, 2, (8, 8, 8)),
    (12, 3, 1, 1, 1, False, 2, (8, 8, 8)),
    # vox64 (MFMA 4x4x1 + LDS halo tile) shapes: W in {16, 32, 64k}, D % 4 == 0, H % (256/W) == 0
    (4, 4, 3, 1, 1, False, 2, (8, 8, 64)),
    (4, 4, 3, 1, 2, False, 2, (4, 12, 64)),
    (4, 4, 3, 1, 1, False, 1, (4, 4, 128)),
    (8, 8, 3, 1, 1, False, 2, (8, 8, 32)),
    (8, 8, 3, 1, 2, False, 2, (4, 16, 32)),
    (8, 8, 3, 1, 1, False, 2, (4, 16, 16)),
    (8, 8, 3, 1, 2, False, 1, (8, 16, 16)),
    (4, 4, 3, 1, 2, False, 1, (4, 16, 16)),
    (8, 8, 3, 1, 1, False, 1, (4, 4, 64)),
    # vox_s2 (stride-2 MFMA kernel, parity-deinterleaved LDS tile): forward of strided convs, data gradient of transposed ones
    (8, 8, 3, 2, 1, False, 2, (8, 8, 32)),
    (8, 8, 3, 2, 2, False, 2, (4, 8, 32)),
    (4, 4, 3, 2, 1, False, 2, (8, 16, 32)),
    (4, 4, 3, 2, 2, False, 1, (6, 8, 64)),
    (8, 8, 3, 2, 1, True, 2, (4, 4, 16)),
    (4, 4, 3, 2, 2, True, 2, (4, 8, 16)),
    (4, 4, 3, 2, 1, True, 1, (32, 32, 32)),
    (8, 8, 3, 2, 1, False, 2, (32, 32, 64)),     # large enough for the stride-2 MFMA weight gradient (>= 32768 output voxels)
    (4, 4, 3, 2, 2, False, 2, (32, 32, 64)),
    (4, 4, 3, 1, 2, False, 2, (64, 64, 64)),     # dilation 2 at full size: two waves per workgroup on a shared 8-row halo tile
    (4, 4, 3, 1, 1, False, 2, (64, 64, 64)),     # the roofline shape itself (TD = 4 tiles, XCD-ordered)
    # 1x1x1 streaming kernel (>= 32768 voxels, few channels): stems / outer preprocess convs and their data gradients
    (4, 12, 1, 1, 1, False, 2, (32, 32, 32)),
    (12, 4, 1, 1, 1, False, 2, (32, 33, 32)),    # ragged: the last workgroup is partial
    (12, 8, 1, 1, 1, False, 1, (32, 32, 40)),
    (24, 4, 1, 1, 1, False, 2, (32, 32, 32)),    # (its data gradient, 4 -> 24, is the widest destination of the streaming kernel)
    (4, 24, 1, 1, 1, False, 1, (32, 32, 40)),
    (12, 8, 1, 2, 1, False, 2, (32, 32, 64)),    # stride 2: its data gradient is the zero-upsampling form of the same kernel
    (8, 8, 1, 1, 1, False, 2, (16, 16, 16)),     # ... from 16^3 on (round 4): the supernet's pointwise convs of the 16^3 level
    (12, 8, 1, 1, 1, False, 2, (16, 16, 16)),
    (8, 8, 1, 1, 1, False, 3, (16, 16, 24)),     # ragged: 6144 voxels, the last workgroup of a sample is partial
    (8, 4, 1, 1, 1, False, 2, (16, 20, 16)),
    # the stems' 4 -> 12 stride-2 conv (nas.py:29, searched.py:70): weight gradient as Co / 4 column tiles of the 4 -> 4 stride-2 MFMA kernel
    (4, 12, 3, 2, 1, False, 2, (64, 64, 64)),
    (4, 8, 3, 2, 1, False, 2, (32, 64, 64)),
    (4, 16, 3, 2, 2, False, 2, (64, 32, 64)),
]


def _mk(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


@pytest.mark.parametrize("cin,cout,k,stride,dil,transposed,B,shape", CASES)
def test_conv_family(cin, cout, k, stride, dil, transposed, B, shape):
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd.prim_ops import _padding
    pad = _padding(k, stride, dil)
    xn = _mk((B, cin) + shape, 1)
    wn = _mk((cin, cout, k, k, k) if transposed else (cout, cin, k, k, k), 2, 1.0 / np.sqrt(cin * k ** 3))
    bn = _mk((cout,), 3, 0.1)
    gate_n = np.abs(_mk((B, cin), 4)) + 0.5
    # ---------------- CPU reference
    xc = torch.from_numpy(xn).requires_grad_(True)
    wc = torch.from_numpy(wn).requires_grad_(True)
    bc = torch.from_numpy(bn).requires_grad_(True)
    if transposed:
        yc = F.conv_transpose3d(xc, wc, bc, stride=stride, padding=pad, output_padding=0 if stride == 1 else 1, dilation=dil)
    else:
        yc = F.conv3d(xc, wc, bc, stride=stride, padding=pad, dilation=dil)
    rn = _mk(tuple(yc.shape), 5)
    (yc * torch.from_numpy(rn)).sum().backward()
    # ---------------- HIP
    dev = torch.device("cuda")
    x = K.as_view(torch.from_numpy(xn).to(dev))
    w, b = torch.from_numpy(wn).to(dev), torch.from_numpy(bn).to(dev)
    Bn, Co = yc.shape[0], yc.shape[1]
    if transposed:
        g = K.conv_geom(B, yc.shape[2], yc.shape[3], yc.shape[4], cout, cin, k, stride, dil, pad)
    else:
        g = K.conv_geom(B, shape[0], shape[1], shape[2], cin, cout, k, stride, dil, pad)
    y = K.as_view(K.empty_ndhwc(Bn, Co, yc.shape[2], yc.shape[3], yc.shape[4], dev))
    rows = K.conv_stats_rows(g, transposed)
    stats = torch.zeros((B, max(rows, 1), Co, 2), dtype=torch.float64, device=dev) if rows > 0 else None
    K.conv_fwd(g, x, w, b, y, 0, None, stats, transposed)
    assert_close(y.t, yc, 2e-5, "y")
    if stats is not None and Co % 4 == 0:
        st = stats.sum(dim=1).cpu().numpy()
        yd = yc.detach().double()
        assert_close(st[..., 0], yd.sum(dim=(2, 3, 4)).numpy(), 1e-5, "stats sum")
        assert_close(st[..., 1], (yd * yd).sum(dim=(2, 3, 4)).numpy(), 1e-5, "stats sumsq")
    dy = K.as_view(torch.from_numpy(rn).to(dev))
    dx = K.as_view(K.empty_ndhwc(B, cin, *shape, dev))
    K.conv_bwd_data(g, dy, w, dx, 0, None, None, transposed)
    assert_close(dx.t, xc.grad, 5e-5, "dx")
    # accumulate flag
    K.conv_bwd_data(g, dy, w, dx, K.ACCUMULATE, None, None, transposed)
    assert_close(dx.t, 2 * xc.grad, 5e-5, "dx accumulate")
    dw, db = torch.empty_like(w), torch.empty_like(b)
    K.conv_bwd_weight(g, x, dy, dw, db, 0, None, transposed)
    assert_close(dw, wc.grad, 1e-4, "dw")
    assert_close(db, bc.grad, 1e-4, "db")
    # weight gradient alone (the hot path gets the bias gradient from the GroupNorm sums): MFMA vox64 kernel where eligible
    dw2 = torch.zeros_like(w)
    K.conv_bwd_weight(g, x, dy, dw2, None, 0, None, transposed)
    assert_close(dw2, wc.grad, 1e-4, "dw (no bias)")
    if transposed:
        dx3 = K.as_view(K.empty_ndhwc(B, cin, *shape, dev))
        dw3 = torch.zeros_like(w)
        K.conv_bwd_both(g, x, dy, w, dx3, dw3, None, 0, None, None, 0, None, True)
        assert_close(dx3.t, xc.grad, 5e-5, "dx (both, transposed)")
        assert_close(dw3, wc.grad, 1e-4, "dw (both, transposed)")
        return
    # ---------------- fused extras (non-transposed): relu on load + input gate
    gate = torch.from_numpy(gate_n).to(dev)
    xc2 = torch.from_numpy(xn).requires_grad_(True)
    wc2 = torch.from_numpy(wn).requires_grad_(True)
    u = F.relu(xc2) * torch.from_numpy(gate_n)[:, :, None, None, None]
    yc2 = F.conv3d(u, wc2, None, stride=stride, padding=pad, dilation=dil)
    (yc2 * torch.from_numpy(rn)).sum().backward()
    K.conv_fwd(g, x, w, None, y, K.RELU_IN, gate, None, False)
    assert_close(y.t, yc2, 2e-5, "y relu+gate")
    K.conv_bwd_data(g, dy, w, dx, 0, x, gate, False)
    assert_close(dx.t, xc2.grad, 5e-5, "dx relu-mask+gate")
    K.conv_bwd_weight(g, x, dy, dw, None, K.RELU_IN, gate, False)
    assert_close(dw, wc2.grad, 1e-4, "dw relu+gate")
    # ---------------- ReLU on load alone (the preprocess convs, 'act_weight_norm'): the 1x1x1 streaming kernels take this form
    if k == 1:
        xc3 = torch.from_numpy(xn).requires_grad_(True)
        wc3 = torch.from_numpy(wn).requires_grad_(True)
        yc3 = F.conv3d(F.relu(xc3), wc3, None, stride=stride, padding=pad, dilation=dil)
        (yc3 * torch.from_numpy(rn)).sum().backward()
        K.conv_fwd(g, x, w, None, y, K.RELU_IN, None, None, False)
        assert_close(y.t, yc3, 2e-5, "y relu")
        dx4 = K.as_view(torch.from_numpy(_mk((B, cin) + shape, 9)).to(dev))
        base = dx4.t.clone()
        K.conv_bwd_data(g, dy, w, dx4, K.ACCUMULATE, x, None, False)
        assert_close(dx4.t, base.cpu() + xc3.grad, 5e-5, "dx relu-mask + accumulate")
        dw4, db4 = torch.zeros_like(w), torch.zeros_like(b)
        K.conv_bwd_weight(g, x, dy, dw4, db4, K.RELU_IN, None, False)
        assert_close(dw4, wc3.grad, 1e-4, "dw relu")
        assert_close(db4, bc.grad, 1e-4, "db relu")
    # ---------------- combined backward (one launch on the deep-level shapes): same results as the two calls
    dx3 = K.as_view(K.empty_ndhwc(B, cin, *shape, dev))
    dw3, db3 = torch.zeros_like(w), torch.zeros_like(b)
    K.conv_bwd_both(g, x, dy, w, dx3, dw3, db3, 0, None, None, 0, None)
    assert_close(dx3.t, xc.grad, 5e-5, "dx (both)")
    assert_close(dw3, wc.grad, 1e-4, "dw (both)")
    assert_close(db3, bc.grad, 1e-4, "db (both)")
    K.conv_bwd_both(g, x, dy, w, dx3, dw3, None, K.ACCUMULATE, x, gate, K.RELU_IN, gate)
    assert_close(dx3.t, xc.grad + xc2.grad, 5e-5, "dx (both, accumulate + relu-mask + gate)")
    assert_close(dw3, wc2.grad, 1e-4, "dw (both, relu + gate)")


@pytest.mark.parametrize("c,stride,transposed,shape", [(4, 1, False, (6, 8, 10)), (8, 2, False, (8, 8, 8)), (16, 2, True, (4, 4, 6)),
                                                         (64, 1, False, (2, 2, 2)), (32, 2, True, (2, 2, 2)),
                                                         # LDS-tile weight gradient: W % 16 == 0, H % 4 == 0, D % 4 == 0, >= 64 (tile, quad) units
                                                         (4, 1, False, (16, 16, 32)), (8, 1, False, (8, 12, 32)), (16, 1, False, (16, 16, 16)),
                                                         (12, 1, False, (12, 8, 48))])
def test_depthwise_family(c, stride, transposed, shape):
    from nas_3d_unet_amd import kernels as K
    B = 2
    xn, wn, bn = _mk((B, c) + shape, 1), _mk((c, 1, 3, 3, 3), 2, 0.2), _mk((c,), 3, 0.1)
    xc, wc, bc = (torch.from_numpy(a).requires_grad_(True) for a in (xn, wn, bn))
    if transposed:
        yc = F.conv_transpose3d(xc, wc, bc, stride=stride, padding=1, output_padding=0 if stride == 1 else 1, groups=c)
    else:
        yc = F.conv3d(xc, wc, bc, stride=stride, padding=1, groups=c)
    rn = _mk(tuple(yc.shape), 5)
    (yc * torch.from_numpy(rn)).sum().backward()
    dev = torch.device("cuda")
    x, w, b = K.as_view(torch.from_numpy(xn).to(dev)), torch.from_numpy(wn).to(dev), torch.from_numpy(bn).to(dev)
    if transposed:
        g = K.conv_geom(B, yc.shape[2], yc.shape[3], yc.shape[4], c, c, 3, stride, 1, 1, True)
    else:
        g = K.conv_geom(B, *shape, c, c, 3, stride, 1, 1, True)
    y = K.as_view(K.empty_ndhwc(B, c, yc.shape[2], yc.shape[3], yc.shape[4], dev))
    K.conv_fwd(g, x, w, b, y, 0, None, None, transposed)
    assert_close(y.t, yc, 2e-5, "y")
    dy = K.as_view(torch.from_numpy(rn).to(dev))
    dx = K.as_view(K.empty_ndhwc(B, c, *shape, dev))
    K.conv_bwd_data(g, dy, w, dx, 0, None, None, transposed)
    assert_close(dx.t, xc.grad, 5e-5, "dx")
    dw, db = torch.empty_like(w), torch.empty_like(b)
    K.conv_bwd_weight(g, x, dy, dw, db, 0, None, transposed)
    assert_close(dw, wc.grad, 1e-4, "dw")
    assert_close(db, bc.grad, 1e-4, "db")


@pytest.mark.parametrize("specA,specB", [
    ((64, 64, 3, 1, 1, False, (4, 4, 4)), (64, 64, 3, 1, 2, False, (4, 4, 4))),      # both K-split-16 GEMMs: one launch
    ((32, 32, 3, 2, 1, True, (4, 4, 4)), (32, 32, 3, 1, 1, False, (8, 8, 8))),       # transposed + plain
    ((16, 16, 3, 1, 1, False, (16, 16, 16)), (16, 16, 3, 1, 2, False, (16, 16, 16))),  # K-split-4 plan
    ((96, 32, 1, 2, 1, False, (8, 8, 8)), (48, 32, 1, 1, 1, False, (4, 4, 4))),      # the two preprocess convs of a cell
    ((8, 8, 3, 1, 1, False, (8, 8, 16)), (64, 64, 3, 1, 1, False, (4, 4, 4))),       # not foldable: falls back to two launches
    ((8, 8, 3, 1, 2, False, (8, 8, 16)), (8, 8, 3, 2, 1, False, (8, 8, 32))),        # two one-wave-tile convs: one multi-conv launch
    ((4, 4, 3, 2, 2, False, (8, 8, 32)), (4, 4, 3, 1, 1, False, (8, 8, 16))),        # the same at C = 4 (two-plane stride-2 tile)
])
def test_conv_pairs_match_single_calls(specA, specB):
    """n3d_conv_fwd2 / n3d_conv_bwd_both2 == the two single calls (and torch CPU), whatever launch grouping libn3d picks."""
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd.prim_ops import _padding
    dev = torch.device("cuda")
    B = 2
    fwd_calls, bwd_calls, refs, outs = [], [], [], []
    for si, (cin, cout, k, stride, dil, transposed, shape) in enumerate((specA, specB)):
        pad = _padding(k, stride, dil)
        xn = _mk((B, cin) + shape, 11 + si)
        wn = _mk((cin, cout, k, k, k) if transposed else (cout, cin, k, k, k), 21 + si, 1.0 / np.sqrt(cin * k ** 3))
        bn = _mk((cout,), 31 + si, 0.1)
        xc, wc, bc = (torch.from_numpy(a).requires_grad_(True) for a in (xn, wn, bn))
        if transposed:
            yc = F.conv_transpose3d(xc, wc, bc, stride=stride, padding=pad, output_padding=0 if stride == 1 else 1, dilation=dil)
            g = K.conv_geom(B, yc.shape[2], yc.shape[3], yc.shape[4], cout, cin, k, stride, dil, pad)
        else:
            yc = F.conv3d(xc, wc, bc, stride=stride, padding=pad, dilation=dil)
            g = K.conv_geom(B, shape[0], shape[1], shape[2], cin, cout, k, stride, dil, pad)
        rn = _mk(tuple(yc.shape), 41 + si)
        (yc * torch.from_numpy(rn)).sum().backward()
        x = K.as_view(torch.from_numpy(xn).to(dev))
        w, b = torch.from_numpy(wn).to(dev), torch.from_numpy(bn).to(dev)
        y = K.as_view(K.empty_ndhwc(B, yc.shape[1], yc.shape[2], yc.shape[3], yc.shape[4], dev))
        rows = K.conv_stats_rows(g, transposed)
        stats = torch.zeros((B, rows, yc.shape[1], 2), dtype=torch.float64, device=dev) if rows > 0 else None
        fwd_calls.append((g, x, w, b, y, 0, None, stats, transposed))
        dy = K.as_view(torch.from_numpy(rn).to(dev))
        dx = K.as_view(K.empty_ndhwc(B, cin, *shape, dev))
        dw = torch.zeros_like(w)
        db = None if transposed else torch.zeros_like(b)
        bwd_calls.append((g, x, dy, w, dx, dw, db, 0, None, None, 0, None, transposed))
        refs.append((yc.detach(), xc.grad, wc.grad, bc.grad))
        outs.append((y, stats, dx, dw, db))
    K.conv_fwd2(fwd_calls)
    K.conv_bwd_both2(bwd_calls)
    for (yc, dxc, dwc, dbc), (y, stats, dx, dw, db) in zip(refs, outs):
        assert_close(y.t, yc, 2e-5, "y (pair)")
        if stats is not None:
            assert_close(stats.sum(dim=1).cpu().numpy()[..., 0], yc.double().sum(dim=(2, 3, 4)).numpy(), 1e-5, "stats (pair)")
        assert_close(dx.t, dxc, 5e-5, "dx (pair)")
        assert_close(dw, dwc, 1e-4, "dw (pair)")
        if db is not None:
            assert_close(db, dbc, 1e-4, "db (pair)")
    # data gradients only (n3d_conv_bwd_data2), accumulating into a pre-filled target behind the ReLU mask of the conv input
    # where the single call supports it: == what the two single calls give
    pair, single = [], []
    for (g, x, dy, w, dx, dw, db, _, _, _, _, _, transposed) in bwd_calls:
        mask_src = None if transposed else x
        base = torch.from_numpy(_mk(tuple(dx.t.shape), 77)).to(dev)
        t2 = K.as_view(K.empty_ndhwc(*dx.t.shape, dev)); t2.t.copy_(base)
        t1 = K.as_view(K.empty_ndhwc(*dx.t.shape, dev)); t1.t.copy_(base)
        pair.append((g, dy, w, t2, K.ACCUMULATE, mask_src, None, transposed))
        K.conv_bwd_data(g, dy, w, t1, K.ACCUMULATE, mask_src, None, transposed)
        single.append(t1)
    K.conv_bwd_data2(pair)
    for c, t1 in zip(pair, single):
        assert_close(c[3].t, t1.t.cpu().numpy(), 1e-6, "dx (data-gradient pair)")


@pytest.mark.parametrize("specs", [
    [(64, 64, 3, 1, 1, (4, 4, 4)), (64, 64, 3, 1, 2, (4, 4, 4)), (64, 64, 3, 1, 1, (4, 4, 4)), (64, 64, 3, 1, 2, (4, 4, 4))],   # one launch
    [(32, 32, 3, 1, 1, (8, 8, 8)), (32, 32, 3, 1, 2, (8, 8, 8)), (32, 32, 3, 2, 1, (16, 16, 16))],                            # three
    [(16, 16, 3, 1, 1, (16, 16, 16)), (8, 8, 3, 1, 1, (8, 8, 16)), (64, 64, 3, 1, 1, (4, 4, 4))],                             # not foldable
    [(8, 8, 3, 1, 2, (16, 16, 16)), (8, 8, 3, 2, 1, (16, 16, 32)), (8, 8, 3, 1, 1, (16, 16, 16)), (8, 8, 3, 2, 2, (16, 16, 32))],  # one multi-conv launch
    [(4, 4, 3, 1, 1, (8, 8, 16)), (4, 4, 3, 2, 1, (8, 8, 32)), (4, 4, 3, 2, 2, (7, 8, 32))],                                  # C = 4 kinds
])
def test_conv_fwdN_matches_torch(specs):
    """n3d_conv_fwdN (up to four forward convs, as few launches as they fold into) against torch CPU, statistics included"""
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd.prim_ops import _padding
    dev = torch.device("cuda")
    B = 2
    calls, refs, outs = [], [], []
    for si, (cin, cout, k, stride, dil, shape) in enumerate(specs):
        pad = _padding(k, stride, dil)
        xn, wn, bn = _mk((B, cin) + shape, 50 + si), _mk((cout, cin, k, k, k), 60 + si, 1.0 / np.sqrt(cin * k ** 3)), _mk((cout,), 70 + si, 0.1)
        yc = F.conv3d(torch.from_numpy(xn), torch.from_numpy(wn), torch.from_numpy(bn), stride=stride, padding=pad, dilation=dil)
        g = K.conv_geom(B, shape[0], shape[1], shape[2], cin, cout, k, stride, dil, pad)
        x = K.as_view(torch.from_numpy(xn).to(dev))
        y = K.as_view(K.empty_ndhwc(B, cout, yc.shape[2], yc.shape[3], yc.shape[4], dev))
        rows = K.conv_stats_rows(g, False)
        stats = torch.zeros((B, rows, cout, 2), dtype=torch.float64, device=dev) if rows > 0 else None
        calls.append((g, x, torch.from_numpy(wn).to(dev), torch.from_numpy(bn).to(dev), y, 0, None, stats, False))
        refs.append(yc)
        outs.append((y, stats))
    K.conv_fwdN(calls)
    for yc, (y, stats) in zip(refs, outs):
        assert_close(y.t, yc, 2e-5, "y (fwdN)")
        if stats is not None:
            assert_close(stats.sum(dim=1).cpu().numpy()[..., 0], yc.double().sum(dim=(2, 3, 4)).numpy(), 1e-5, "stats (fwdN)")


@pytest.mark.parametrize("C", [4, 8])
def test_multi_conv_launch_is_bit_identical_to_single_launches(C):
    """The one-wave-tile 3x3x3 convs of a supernet node folded into ONE launch (conv_vox_multi_kernel: each conv's kernel body on its
    own range of workgroups) give the single launches' outputs and statistics rows bit for bit -- forward (n3d_conv_fwdN / fwd2, one
    conv accumulating into a pre-filled tensor) and data gradients (n3d_conv_bwd_data2)."""
    from nas_3d_unet_amd import kernels as K
    dev = torch.device("cuda")
    B = 2
    specs = [(1, 2, (16, 16, 16), 0), (2, 1, (16, 16, 32), K.ACCUMULATE), (1, 1, (16, 16, 16), 0), (2, 2, (16, 16, 32), 0)]
    for n in (2, 3, 4):
        calls, singles = [], []
        for si, (stride, dil, shape, fl) in enumerate(specs[:n]):
            g = K.conv_geom(B, shape[0], shape[1], shape[2], C, C, 3, stride, dil, dil)
            x = K.as_view(torch.from_numpy(_mk((B, C) + shape, 150 + si)).to(dev).contiguous(memory_format=torch.channels_last_3d))
            w, b = torch.from_numpy(_mk((C, C, 3, 3, 3), 160 + si, 0.1)).to(dev), torch.from_numpy(_mk((C,), 170 + si, 0.1)).to(dev)
            so = tuple(d // stride for d in shape)
            base = torch.from_numpy(_mk((B, C) + so, 180 + si)).to(dev)
            rows = K.conv_stats_rows(g, False)
            ys, sts = [], []
            for _ in range(2):
                y = K.as_view(K.empty_ndhwc(B, C, *so, dev)); y.t.copy_(base)
                ys.append(y); sts.append(torch.zeros((B, rows, C, 2), dtype=torch.float64, device=dev))
            calls.append((g, x, w, b, ys[0], fl, None, sts[0], False))
            K.conv_fwd(g, x, w, b, ys[1], fl, None, sts[1], False)
            singles.append((ys[1], sts[1]))
        K.conv_fwdN(calls) if n > 2 else K.conv_fwd2(calls)
        for c, (y1, st1) in zip(calls, singles):
            assert torch.equal(c[4].t, y1.t) and torch.equal(c[7], st1), f"forward, {n} convs"
    # data gradients of two stride-1 convs (the kernel on mirrored, transposed weights), one accumulating
    pair, single = [], []
    for si, (dil, fl) in enumerate(((1, K.ACCUMULATE), (2, 0))):
        shape = (16, 16, 16)
        g = K.conv_geom(B, *shape, C, C, 3, 1, dil, dil)
        dy = K.as_view(torch.from_numpy(_mk((B, C) + shape, 190 + si)).to(dev).contiguous(memory_format=torch.channels_last_3d))
        w = torch.from_numpy(_mk((C, C, 3, 3, 3), 195 + si, 0.1)).to(dev)
        base = torch.from_numpy(_mk((B, C) + shape, 197 + si)).to(dev)
        t = []
        for _ in range(2):
            d = K.as_view(K.empty_ndhwc(B, C, *shape, dev)); d.t.copy_(base); t.append(d)
        pair.append((g, dy, w, t[0], fl, None, None, False))
        K.conv_bwd_data(g, dy, w, t[1], fl, None, None, False)
        single.append(t[1])
    K.conv_bwd_data2(pair)
    for c, t1 in zip(pair, single):
        assert torch.equal(c[3].t, t1.t), "data gradients"
