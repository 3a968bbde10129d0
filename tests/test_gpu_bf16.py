"""GPU: the bf16-storage configuration (BASELINE configs[4]: 4x128^3 patches; reference config.yml:58 patch_shape 128,
train.py:117-128).  Storage of the HBM-bound levels' activations / activation gradients is bfloat16, arithmetic is fp32.

Stated tolerances (bf16 has 8 significand bits: one rounding = 2^-9 relative; a value passes through ~25 stored tensors between
input and logits, a gradient through ~50):
  * kernel level: a bf16-storage kernel equals the fp32 kernel run on the SAME bf16-rounded inputs up to the rounding of its
    own output: |err| <= 2^-8 * max|ref| (outputs stored in bf16), 1e-5 relative where the output is fp32;
  * whole net vs the fp32 reference (golden vectors / CPU oracle): loss 2e-3 abs, logits 3e-2 * range, probabilities 3e-2
    abs, parameter-gradient vector 6e-2 relative in L2, every tensor within 1.5e-2 * ||g_total||;
  * run to run: bit-identical."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_common as gc
from _util import assert_close, dev
from oracle import ref_path as orc
from test_gpu_nets import build_net, keep_logits

pytestmark = pytest.mark.gpu

TOL_LOSS, TOL_LOGITS, TOL_PROBS, TOL_GVEC, TOL_GTENSOR = 2e-3, 3e-2, 3e-2, 6e-2, 1.5e-2
ULP = 2.0 ** -8


def _bf(a):
    return torch.from_numpy(np.ascontiguousarray(a)).bfloat16()


def _view(t16):
    from nas_3d_unet_amd import kernels as K
    B, C, D, H, W = t16.shape
    v = K.empty_ndhwc(B, C, D, H, W, torch.device("cuda"), torch.bfloat16)
    v.copy_(t16.cuda())
    return K.as_view(v)


@pytest.mark.parametrize("cin,cout,k,stride,dil,transposed,shape", [
    (4, 12, 1, 1, 1, False, (32, 32, 32)),     # stem0 (streaming 1x1x1 kernel)
    (12, 4, 1, 1, 1, False, (32, 32, 32)),     # preprocess of the top cell
    (12, 8, 1, 2, 1, False, (16, 16, 16)),     # stride-2 preprocess (gather)
    (4, 12, 3, 2, 1, False, (16, 16, 16)),     # stem1
    (4, 4, 3, 1, 1, False, (8, 16, 16)),       # conv
    (8, 8, 3, 1, 2, False, (8, 16, 16)),       # dil_conv
    (8, 8, 3, 2, 1, False, (8, 16, 16)),       # down_conv
    (4, 4, 3, 2, 1, True, (4, 8, 8)),          # up_conv
    (8, 8, 3, 2, 2, True, (4, 8, 8)),          # up_dil_conv
    (24, 16, 1, 1, 1, False, (8, 8, 8)),       # boundary conv: bf16 in, fp32 out (checked in both mixes below)
    (4, 4, 3, 1, 1, False, (64, 64, 64)),      # the bf16 MFMA kernel, 4-plane tiles (conv_vox64b_kernel<4,4,1,1>)
    (4, 4, 3, 1, 2, False, (64, 64, 64)),      # ... two waves per 8-row tile (dilation 2)
    (8, 8, 3, 1, 1, False, (64, 64, 32)),      # ... C = 8, 2-plane tiles
    (8, 8, 3, 1, 2, False, (32, 32, 32)),      # ... C = 8, dilation 2, single-plane tiles
    (4, 4, 3, 1, 1, False, (8, 12, 16)),       # ... ragged tile counts (H = 12)
    (4, 4, 3, 2, 1, False, (32, 32, 64)),      # stride 2 on the bf16 MFMA kernels: forward = conv_vox_s2b<4,2,1>, data gradient = conv_vox_upb<4,1>
    (4, 4, 3, 2, 2, False, (32, 32, 64)),      # ... dilation 2
    (8, 8, 3, 2, 1, False, (16, 16, 64)),      # ... C = 8
    (8, 8, 3, 2, 2, False, (8, 16, 32)),
    (4, 4, 3, 2, 1, True, (8, 8, 16)),         # transposed: forward = conv_vox_upb, data gradient = conv_vox_s2b
    (8, 8, 3, 2, 2, True, (4, 8, 16)),
    (4, 4, 3, 2, 2, True, (16, 16, 32)),
    (4, 4, 3, 2, 1, False, (64, 64, 64)),      # large enough for the MFMA stride-2 weight gradient (vox_wgrad_s2_kernel<.., bf16>)
    (8, 8, 3, 2, 2, False, (32, 64, 64)),
    (4, 4, 3, 2, 1, True, (32, 32, 32)),       # ... with the roles swapped (transposed conv)
    (4, 12, 3, 2, 1, False, (64, 64, 64)),     # stem1 at full size: co-tiled stride-2 MFMA weight gradient, fp32 input with a bf16 gradient ("f32->bf16")
])
@pytest.mark.parametrize("mix", ["bf16->bf16", "bf16->f32", "f32->bf16"])
def test_conv_family_bf16_storage(cin, cout, k, stride, dil, transposed, shape, mix):
    """forward, data gradient and weight gradient of every conv shape of the bf16 cells through the C ABI (N3D_SRC_BF16 /
    N3D_DST_BF16) against torch fp32 on the bf16-rounded operands"""
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd.prim_ops import _padding
    rng = np.random.default_rng(cin * 1000 + cout * 10 + k + stride)
    B = 2
    pad = _padding(k, stride, dil)
    opad = 0 if stride == 1 else 1
    x16 = _bf(rng.standard_normal((B, cin) + shape).astype(np.float32))
    # weights representable in bf16: the bf16 MFMA kernels (3x3x3, stride 1, C in {4, 8}, both tensors bf16) round them, the
    # streaming / gather kernels keep them in fp32 -- with representable weights both compute the same products
    w = torch.from_numpy((rng.standard_normal((cin, cout, k, k, k) if transposed else (cout, cin, k, k, k)) * 0.2).astype(np.float32)).bfloat16().float()
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    xin = x16.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    if transposed:
        yr = F.conv_transpose3d(xin, wr, b, stride=stride, padding=pad, output_padding=opad, dilation=dil)
    else:
        yr = F.conv3d(xin, wr, b, stride=stride, padding=pad, dilation=dil)
    dy16 = _bf(rng.standard_normal(tuple(yr.shape)).astype(np.float32))
    yr.backward(dy16.float())
    in16, out16 = mix.startswith("bf16"), mix.endswith("bf16")
    dev_ = torch.device("cuda")
    xv = _view(x16) if in16 else K.as_view(dev(x16.float().numpy()))
    if transposed:
        g = K.conv_geom(B, yr.shape[2], yr.shape[3], yr.shape[4], cout, cin, k, stride, dil, pad)
    else:
        g = K.conv_geom(B, *shape, cin, cout, k, stride, dil, pad)
    with K.storage(torch.bfloat16 if out16 else torch.float32):
        y = K.as_view(K.empty_ndhwc(*yr.shape, dev_))
    wd, bd = w.cuda(), b.cuda()
    K.conv_fwd(g, xv, wd, bd, y, 0, None, None, transposed)
    assert_close(y.t.float(), yr.detach(), ULP if out16 else 1e-5, "y")
    # data gradient: dy has the output's storage type, dx the input's
    dyv = _view(dy16) if out16 else K.as_view(dev(dy16.float().numpy()))
    dx = K.like(xv)
    K.conv_bwd_data(g, dyv, wd, dx, 0, None, None, transposed)
    assert_close(dx.t.float(), xin.grad, ULP if in16 else 1e-5, "dx")
    dw, db = torch.empty_like(wd), torch.empty_like(bd)
    K.conv_bwd_weight(g, xv, dyv, dw, None if transposed else db, 0, None, transposed)
    assert_close(dw, wr.grad, 2e-5, "dw")
    if not transposed:
        # without the bias gradient (the hot path derives it from the GroupNorm sums): the MFMA weight-gradient kernels
        dw2 = torch.empty_like(wd)
        K.conv_bwd_weight(g, xv, dyv, dw2, None, 0, None, False)
        assert_close(dw2, wr.grad, 2e-5, "dw (no bias gradient)")


@pytest.mark.parametrize("dil", [1, 2])
def test_dense_bf16_conv_8_plane_tiles_and_their_statistics_rows(dil):
    """round 3: a DENSE 4-channel bf16 source takes the two-voxels-per-slot LDS image and, with >= 4096 tiles, 8 output planes per
    tile -- which must still write the statistics rows of the 4-plane plan (n3d_conv_stats_rows does not know which form runs): the
    output against torch fp32 on the bf16-rounded operands, every statistics row against the sums of its own 4 x 4 x 16 voxel block"""
    from nas_3d_unet_amd import kernels as K
    rng = np.random.default_rng(90 + dil)
    B, C, shape = 2, 4, (64, 128, 128)
    x16 = _bf(rng.standard_normal((B, C) + shape).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((C, C, 3, 3, 3)) * 0.2).astype(np.float32)).bfloat16().float()
    b = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.1)
    yr = F.conv3d(x16.float(), w, b, padding=dil, dilation=dil)
    xv = _view(x16)
    assert xv.ld == 4                                             # dense: the pair-fetching image
    g = K.conv_geom(B, *shape, C, C, 3, 1, dil, dil)
    with K.storage(torch.bfloat16):
        y = K.as_view(K.empty_ndhwc(*yr.shape, torch.device("cuda")))
    rows = K.conv_stats_rows(g, False, 0, xv, y)
    assert rows == (shape[0] // 4) * (shape[1] // 4) * (shape[2] // 16)
    stats = torch.full((B, rows, C, 2), float("nan"), dtype=torch.float64, device="cuda")
    K.conv_fwd(g, xv, w.cuda(), b.cuda(), y, 0, None, stats, False)
    assert_close(y.t.float(), yr, ULP, "y")
    st = stats.cpu()
    assert not torch.isnan(st).any(), "a statistics row of the 4-plane plan was not written"
    # row (d4, h4, w16) = sums over that block of the fp32 result (before the bf16 rounding of the store)
    blk = yr.double().reshape(B, C, shape[0] // 4, 4, shape[1] // 4, 4, shape[2] // 16, 16)
    s1 = blk.sum(dim=(3, 5, 7)).permute(0, 2, 3, 4, 1).reshape(B, rows, C)
    s2 = (blk * blk).sum(dim=(3, 5, 7)).permute(0, 2, 3, 4, 1).reshape(B, rows, C)
    if dil == 1:
        assert float((st[..., 0] - s1).abs().max()) <= 1e-4 * float(s1.abs().max())
        assert float((st[..., 1] - s2).abs().max()) <= 1e-4 * float(s2.abs().max())
    else:
        # dilation 2 runs two-wave 8-row tiles, whose rows come in (tile, wave) order: the per-sample totals are what GroupNorm uses
        assert float((st[..., 0].sum(1) - s1.sum(1)).abs().max()) <= 1e-5 * float(s2.sum(1).max())
        assert float((st[..., 1].sum(1) - s2.sum(1)).abs().max()) <= 1e-5 * float(s2.sum(1).max())


@pytest.mark.parametrize("dil,shape", [(1, (128, 128, 128)), (1, (40, 128, 128)), (2, (64, 64, 128)), (1, (72, 64, 128))])
def test_dense_bf16_conv_8_plane_tiles_forward_and_accumulating_data_gradient(dil, shape):
    """round 5: the 8-plane tile form of the dense 4-channel bf16 3x3x3 conv fills its LDS tile through BUFFER loads (zero padding by the
    resource's bounds check) and has its accumulate flag compiled in: forward with statistics rows and the data gradient ACCUMULATING
    into its destination, full volumes and D not a multiple of 16, against torch fp32 on the bf16-rounded operands."""
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd._lib import ACCUMULATE
    rng = np.random.default_rng(190 + dil + shape[0])
    B, C = 2, 4
    x16 = _bf(rng.standard_normal((B, C) + shape).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((C, C, 3, 3, 3)) * 0.2).astype(np.float32)).bfloat16().float()
    b = torch.from_numpy(rng.standard_normal(C).astype(np.float32) * 0.1)
    yr = F.conv3d(x16.float(), w, b, padding=dil, dilation=dil)
    xv = _view(x16)
    assert xv.ld == 4
    g = K.conv_geom(B, *shape, C, C, 3, 1, dil, dil)
    with K.storage(torch.bfloat16):
        y = K.as_view(K.empty_ndhwc(*yr.shape, torch.device("cuda")))
    rows = K.conv_stats_rows(g, False, 0, xv, y)
    stats = torch.full((B, rows, C, 2), float("nan"), dtype=torch.float64, device="cuda")
    K.conv_fwd(g, xv, w.cuda(), b.cuda(), y, 0, None, stats, False)
    assert_close(y.t.float(), yr, ULP, "y")
    st = stats.cpu()
    assert not torch.isnan(st).any(), "a statistics row was not written"
    assert float((st[..., 0].sum(1) - yr.double().sum(dim=(2, 3, 4))).abs().max()) <= 1e-5 * float((yr.double() ** 2).sum(dim=(2, 3, 4)).max())
    assert float((st[..., 1].sum(1) - (yr.double() ** 2).sum(dim=(2, 3, 4))).abs().max()) <= 1e-5 * float((yr.double() ** 2).sum(dim=(2, 3, 4)).max())
    if dil == 1:      # one-wave tiles: every row against its own 4 x 4 x 16 block (dilation 2 runs two-wave tiles: rows in (tile, wave) order)
        assert rows == (shape[0] // 4) * (shape[1] // 4) * (shape[2] // 16)
        blk = yr.double().reshape(B, C, shape[0] // 4, 4, shape[1] // 4, 4, shape[2] // 16, 16)
        s1 = blk.sum(dim=(3, 5, 7)).permute(0, 2, 3, 4, 1).reshape(B, rows, C)
        assert float((st[..., 0] - s1).abs().max()) <= 1e-4 * float(s1.abs().max())
    # data gradient, accumulating (the backward walk's form): dx_prev + conv^T(dy)
    dy16 = _bf(rng.standard_normal((B, C) + shape).astype(np.float32))
    dx0 = _bf(rng.standard_normal((B, C) + shape).astype(np.float32))
    dxr = dx0.float() + F.conv_transpose3d(dy16.float(), w, None, padding=dil, dilation=dil)
    dxv = _view(dx0.clone())
    K.conv_bwd_data(g, _view(dy16), w.cuda(), dxv, ACCUMULATE, None, None, False)
    assert_close(dxv.t.float(), dxr, ULP, "dx (accumulating)")


@pytest.mark.parametrize("C,shape,B", [(4, (16, 16, 16), 2), (8, (32, 16, 16), 2), (8, (4, 4, 4), 2), (4, (48, 32, 32), 1)])
def test_node_epilogues_bf16_storage(C, shape, B):
    """GroupNorm -> ReLU -> node sum of two conv outputs (searched.py:45-50) and its backward with every tensor in bf16:
    against the same kernels in fp32 storage on the bf16-rounded operands (forward: the stored node rounds once more)"""
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd.programs import group_count
    rng = np.random.default_rng(C * 10 + B)
    G = group_count(C)
    raws16 = [_bf(rng.standard_normal((B, C) + shape).astype(np.float32) * 1.5 + 0.3) for _ in range(2)]
    dout16 = _bf(rng.standard_normal((B, C) + shape).astype(np.float32))
    gam = [torch.from_numpy(rng.uniform(0.5, 1.5, C).astype(np.float32)).cuda().requires_grad_(True) for _ in range(2)]
    bet = [torch.from_numpy(rng.standard_normal(C).astype(np.float32)).cuda().requires_grad_(True) for _ in range(2)]
    res = {}
    for tag in ("f32", "bf16"):
        if tag == "bf16":
            raws = [_view(r) for r in raws16]
            dout = _view(dout16)
        else:
            raws = [K.as_view(dev(r.float().numpy())) for r in raws16]
            dout = K.as_view(dev(dout16.float().numpy()))
        sts = [K.channel_stats(r) for r in raws]
        out = K.like(raws[0])
        terms = [(raws[i], sts[i][0], sts[i][1], gam[i], bet[i], None, True) for i in range(2)]
        sv = K.affine_act_gn2(terms, G, 1e-5, out, 0)
        tds = [dict(raw=raws[i], a=sv[i][0], b=sv[i][1], mr=sv[i][2], sumraw=sv[i][3], gamma=gam[i], beta=bet[i], wptr=None, relu=True,
                    conv_bias=None, draw=K.like(raws[i]), dalpha_ptr=None) for i in range(2)]
        gouts = K.affine_act_bwd_gn2(dout, tds, G)
        res[tag] = (out.t.float().clone(), [t["draw"].t.float().clone() for t in tds], [(a.clone(), b.clone()) for a, b, _ in gouts])
    assert_close(res["bf16"][0], res["f32"][0].cpu().numpy(), ULP, "node")
    for i in range(2):
        assert_close(res["bf16"][1][i], res["f32"][1][i].cpu().numpy(), ULP, "draw%d" % i)
        assert_close(res["bf16"][2][i][0], res["f32"][2][i][0].cpu().numpy(), 1e-5, "dgamma%d" % i)
        assert_close(res["bf16"][2][i][1], res["f32"][2][i][1].cpu().numpy(), 1e-5, "dbeta%d" % i)


def _run_bf16(net, x, t):
    with keep_logits() as k:
        l, p = net.forward_loss(x, t)
        logits = k.logits
    l.backward()
    return float(l), logits, p.detach()


def _check_against_fp32(l, logits, p, grads, lr, zr, pr, gr):
    assert abs(l - lr) <= TOL_LOSS, (l, lr)
    assert float((logits.cpu() - zr).abs().max()) <= TOL_LOGITS * float(zr.abs().max())
    assert float((p.cpu() - pr).abs().max()) <= TOL_PROBS
    tot = float(torch.sqrt(sum((g.double() ** 2).sum() for g in gr.values())))
    err2 = 0.0
    for n, g in grads.items():
        d = float((g.cpu().double() - gr[n].double()).norm())
        assert d <= TOL_GTENSOR * tot, (n, d / tot)
        err2 += d * d
    assert err2 ** 0.5 <= TOL_GVEC * tot, err2 ** 0.5 / tot


def test_searched_net_bf16_vs_golden_d4s64(golden):
    """net/searched/G_CONV/d4s64 (the golden 64^3 patch of the reference) through the bf16-storage path: loss, logits, probabilities"""
    from nas_3d_unet_amd import unet
    g = golden("nets")
    key, kind, gname, depth, size, batch, adam = [c for c in gc.net_cases() if c[0] == "net/searched/G_CONV/d4s64"][0]
    net, _ = build_net(kind, gname, depth)
    unet.set_storage(net, "bf16")
    xn, tn = gc.net_batch(key, batch, size)
    l, logits, p = _run_bf16(net, dev(xn), dev(tn))
    c = size // 2
    s = slice(c - 3, c + 3)
    assert abs(l - float(g[key + "/loss"])) <= TOL_LOSS
    assert np.abs(logits[:, :, s, s, s].cpu().numpy() - g[key + "/logits_crop"]).max() <= TOL_LOGITS * float(g[key + "/logits_absmax"])
    assert np.abs(p[:, :, s, s, s].cpu().numpy() - g[key + "/probs_crop"]).max() <= TOL_PROBS
    total = float(g[key + "/gnorm_total"])
    for n, q in net.named_parameters():
        ref = float(g[key + "/gnorm/" + n])
        assert abs(float(q.grad.double().norm()) - ref) <= TOL_GTENSOR * total + 0.05 * ref, n
    # the storage policy really is in force: stems and the node-width <= 8 cells hold bf16, the deep cells fp32
    dts = [pl.dt for pl in net._net_plan.cells]
    assert net._net_plan.stem_dt == torch.bfloat16 and dts[0] == torch.bfloat16 and dts[-1] == torch.bfloat16 and dts[3] == torch.float32


@pytest.mark.parametrize("size,batch,gname", [(64, 2, "G_CONV"), (128, 1, "G_CONV"), ((32, 64, 96), 3, "G_CONV"), (128, 1, "G_ALL"), (64, 2, "G_ALL")])
def test_searched_net_bf16_vs_cpu_oracle(size, batch, gname):
    """seeded 64^3 (batch 2) and 128^3 (the configuration's patch size) cases against the fp32 CPU oracle: loss, logits,
    probabilities, every parameter gradient; and a second run of the same step is bit-identical.  G_ALL (round 5): the genotype with
    depthwise-separable, SE, pooling and identity primitives -- their cells store bf16 too (prim_ops.py:119-174)"""
    from nas_3d_unet_amd import unet
    shape = (size, size, size) if isinstance(size, int) else size     # also a non-cubic patch with a batch of 3
    rng = np.random.default_rng(shape[0] + shape[2])
    xn = rng.standard_normal((batch, 4) + shape).astype(np.float32)
    tn = (rng.uniform(0, 1, (batch, 3) + shape) < 0.3).astype(np.float32)
    gene = getattr(orc, gname)
    P = orc.make_params(orc.searched_param_specs(orc.DEFAULT_CFG, gene), requires_grad=True)
    pr, zr = orc.searched_forward(P, torch.from_numpy(xn), gene, return_logits=True)
    lr = orc.dice_loss(pr, torch.from_numpy(tn))
    lr.backward()
    runs = []
    for _ in range(2):
        net, _ = build_net("searched", gname, 4)
        unet.set_storage(net, "bf16")
        l, logits, p = _run_bf16(net, dev(xn), dev(tn))
        runs.append((l, logits, p, {n: q.grad.clone() for n, q in net.named_parameters()}))
    l, logits, p, grads = runs[0]
    _check_against_fp32(l, logits, p, grads, float(lr), zr.detach(), pr.detach(), {n: q.grad for n, q in P.items()})
    assert runs[1][0] == l and torch.equal(runs[1][1], logits) and all(torch.equal(runs[1][3][n], grads[n]) for n in grads)


def test_trainer_bf16_storage_graph_replay():
    """Trainer(storage='bf16') under HIP-graph replay: losses of three Adam steps follow the fp32 trainer's within the bf16 tolerance"""
    from nas_3d_unet_amd.train import Trainer
    rng = np.random.default_rng(3)
    x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))
    out = {}
    for st in ("fp32", "bf16"):
        net, _ = build_net("searched", "G_CONV", 4)
        tr = Trainer(net, graph=True, storage=st)
        out[st] = [float(tr.step(x, t)) for _ in range(3)]
    np.testing.assert_allclose(out["bf16"], out["fp32"], rtol=0, atol=TOL_LOSS)
    assert out["bf16"][2] < out["bf16"][0]


@pytest.mark.parametrize("scale", [1.0 / 50.0, 1.0], ids=["scaled", "bench-range"])
def test_bf16_storage_trains_with_fp32_at_128(scale):
    """BASELINE configs[4] at its own size (config.yml:58; train.py:117-128): 20 HIP-graph-replayed Adam steps on a batch of
    4x128^3 patches, bf16 storage against the fp32 trainer from the same weights -- the loss curves must stay together
    (<= 2e-3 at EVERY step) and must actually move.  "bench-range": the synthetic batch exactly as bench.py feeds it (values in
    [10, 110] inside the ball, 0 outside: preprocess.py:87-93), not rescaled."""
    import bench
    from nas_3d_unet_amd.train import Trainer
    xn, tn = bench.synthetic_batch(2, 128, 77)
    x, t = bench.to_patch_layout(dev((xn * scale).astype(np.float32))), dev(tn)
    curves = []
    for storage in (None, "bf16"):
        net, _ = build_net("searched", "G_CONV", 4)       # closed-form weights, head Dropout3d off: the two runs see the same net
        tr = Trainer(net, graph=True, storage=storage)
        curves.append([float(tr.step(x, t)) for _ in range(20)])
        torch.cuda.synchronize()
        tr.check_sync()
        del tr, net
        torch.cuda.empty_cache()
    f32, b16 = np.array(curves[0]), np.array(curves[1])
    assert np.abs(f32 - b16).max() <= 2e-3, (f32, b16)
    assert f32[0] - f32[-1] > 5e-3, "the fp32 run did not train: the comparison would be vacuous"


def test_searched_net_bf16_with_non_conv_primitives_vs_oracle():
    """G_ALL (depthwise-separable, SE, pooling, identity primitives): since round 5 their cells store bf16 like the all-conv ones (the
    node-width <= 8 levels; the deep cells fp32).  Against the fp32 CPU oracle at the bf16 tolerance."""
    from nas_3d_unet_amd import unet
    rng = np.random.default_rng(5)
    xn = rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32)
    tn = (rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32)
    P = orc.make_params(orc.searched_param_specs(orc.DEFAULT_CFG, orc.G_ALL), requires_grad=True)
    pr, zr = orc.searched_forward(P, torch.from_numpy(xn), orc.G_ALL, return_logits=True)
    lr = orc.dice_loss(pr, torch.from_numpy(tn))
    lr.backward()
    net, _ = build_net("searched", "G_ALL", 4)
    unet.set_storage(net, "bf16")
    l, logits, p = _run_bf16(net, dev(xn), dev(tn))
    dts = [pl.dt for pl in net._net_plan.cells]
    assert net._net_plan.stem_dt == torch.bfloat16 and dts[0] == torch.bfloat16 and dts[-1] == torch.bfloat16 and dts[3] == torch.float32
    _check_against_fp32(l, logits, p, {n: q.grad for n, q in net.named_parameters()}, float(lr), zr.detach(), pr.detach(),
                        {n: q.grad for n, q in P.items()})


def test_foreign_bf16_tensor_without_slack_is_repacked():
    """the bf16 3x3x3 kernels read 16 bytes per 8-byte voxel, i.e. up to 8 bytes past a dense 4-channel tensor.
    Tensors the package allocates carry that slack; a caller-owned one that does not must be repacked by as_view, not
    handed to the kernels as it is (include/n3d.h, "READABLE SLACK")."""
    from nas_3d_unet_amd import kernels as K
    own = K.empty_ndhwc(2, 4, 8, 8, 8, torch.device("cuda"), torch.bfloat16)
    assert K.as_view(own).t.data_ptr() == own.data_ptr()                      # slack present: used in place
    z = K.zeros_ndhwc(2, 4, 8, 8, 8, torch.device("cuda"), torch.bfloat16)
    assert K.as_view(z).t.data_ptr() == z.data_ptr() and float(z.float().abs().sum()) == 0.0
    foreign = torch.randn(2, 8, 8, 8, 4, device="cuda").to(torch.bfloat16).permute(0, 4, 1, 2, 3)   # dense NDHWC, exact-size storage
    v = K.as_view(foreign)
    assert v.t.data_ptr() != foreign.data_ptr() and torch.equal(v.t, foreign)
    assert v.t.untyped_storage().nbytes() >= foreign.numel() * 2 + 16
    wide = torch.randn(2, 8, 8, 8, 8, device="cuda").to(torch.bfloat16).permute(0, 4, 1, 2, 3)      # 16-byte voxels: nothing is over-read
    assert K.as_view(wide).t.data_ptr() == wide.data_ptr()
    inner = K.empty_ndhwc(2, 12, 8, 8, 8, torch.device("cuda"), torch.bfloat16)[:, 4:8]               # a channel slice of a wider buffer
    assert K.as_view(inner).t.data_ptr() == inner.data_ptr()


# ------------------------------------------------------------------------------------------------ round 6: N3D_MM_BF16
# The C >= 16 levels of the bf16 configuration keep fp32 storage; their MFMA conv kernels round BOTH operands of the matrix products to
# bfloat16 in registers and accumulate in fp32 (include/n3d.h, N3D_MM_BF16).  The exact statement of that arithmetic is a conv of the
# bf16-ROUNDED tensors evaluated in fp64: per conv, forward / data gradient / weight gradient, every kernel form of the family.
MM_CASES = [
    # (Cin, Cout, stride, dil, transposed, B, spatial of the conv input): kernel form
    (16, 16, 1, 1, False, 2, (32, 32, 32)),   # tile16 + LDS-tile weight gradient (the C = 16 level of 128^3 patches)
    (16, 16, 1, 2, False, 2, (16, 32, 32)),   # ... dilation 2
    (16, 16, 1, 1, False, 2, (16, 16, 16)),   # gemm16 K-split 4 (forward / data gradient), LDS-tile weight gradient
    (16, 16, 2, 1, False, 2, (16, 32, 64)),   # stride 2: tile16_up data gradient
    (16, 16, 2, 2, True, 2, (8, 16, 32)),     # transposed: tile16_up forward
    (32, 32, 1, 1, False, 2, (16, 16, 16)),   # gemm16 K-split 4, LDS-tile weight gradient on 32 channels
    (32, 32, 1, 2, False, 2, (16, 16, 16)),
    (32, 32, 2, 1, False, 2, (16, 16, 16)),   # gemm16 K-split 16 + conv_wgrad16
    (64, 64, 1, 1, False, 2, (8, 8, 8)),      # K-split 16, 4 x 4 x 8 weight-gradient tiles
    (64, 64, 1, 1, False, 2, (4, 4, 4)),
    (64, 64, 2, 1, True, 2, (2, 2, 2)),
    (48, 32, 1, 1, False, 2, (16, 16, 16)),   # 1x1x1 preprocess conv (k = 1 below)
]


def _r16(a):
    return torch.from_numpy(a).to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("cin,cout,stride,dil,transposed,B,shape", MM_CASES)
def test_mm_bf16_conv_family_vs_rounded_operands(cin, cout, stride, dil, transposed, B, shape):
    import torch.nn.functional as F
    from nas_3d_unet_amd import kernels as K
    from nas_3d_unet_amd.prim_ops import _padding
    k = 1 if cin == 48 else 3
    pad = _padding(k, stride, dil)
    rng = np.random.default_rng(11)
    xn = rng.standard_normal((B, cin) + shape).astype(np.float32)
    wn = (rng.standard_normal((cin, cout, k, k, k) if transposed else (cout, cin, k, k, k)) / np.sqrt(cin * k ** 3)).astype(np.float32)
    # reference: the same conv on operands rounded to bf16, in fp64
    xc, wc = _r16(xn).requires_grad_(True), _r16(wn).requires_grad_(True)
    if transposed:
        yc = F.conv_transpose3d(xc, wc, None, stride=stride, padding=pad, output_padding=0 if stride == 1 else 1, dilation=dil)
    else:
        yc = F.conv3d(xc, wc, None, stride=stride, padding=pad, dilation=dil)
    rn = rng.standard_normal(tuple(yc.shape)).astype(np.float32)
    dev_ = torch.device("cuda")
    x = K.as_view(torch.from_numpy(xn).to(dev_))
    w = torch.from_numpy(wn).to(dev_)
    if transposed:
        g = K.conv_geom(B, yc.shape[2], yc.shape[3], yc.shape[4], cout, cin, k, stride, dil, pad)
    else:
        g = K.conv_geom(B, shape[0], shape[1], shape[2], cin, cout, k, stride, dil, pad)
    y = K.as_view(K.empty_ndhwc(B, cout, yc.shape[2], yc.shape[3], yc.shape[4], dev_))
    y32 = K.as_view(K.empty_ndhwc(B, cout, yc.shape[2], yc.shape[3], yc.shape[4], dev_))
    dy = K.as_view(torch.from_numpy(rn).to(dev_))
    dx = K.as_view(K.empty_ndhwc(B, cin, *shape, dev_))
    dw = torch.zeros_like(w)
    K.conv_fwd(g, x, w, None, y32, 0, None, None, transposed)
    with K.storage(torch.float32, True):
        K.conv_fwd(g, x, w, None, y, 0, None, None, transposed)
        K.conv_bwd_data(g, dy, w, dx, 0, None, None, transposed)
        K.conv_bwd_weight(g, x, dy, dw, None, 0, None, transposed)
    # the data / weight gradients round dy too
    (yc * _r16(rn)).sum().backward()
    tol = 2e-5     # fp32 accumulation of exactly representable products: the fp32 kernels' own tolerance
    assert_close(y.t, yc.detach().float(), tol, "y (bf16 operands)")
    assert_close(dx.t, xc.grad.float(), 5e-5, "dx (bf16 operands)")
    assert_close(dw, wc.grad.float(), 1e-4, "dw (bf16 operands)")
    if k == 3:
        # and the flag does select another arithmetic: the fp32 result differs from it at the bf16 rounding level
        d = float((y.t - y32.t).abs().max() / y32.t.abs().max())
        assert 1e-4 < d < 3e-2, d
