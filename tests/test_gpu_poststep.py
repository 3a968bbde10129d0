"""GPU: n3d_stitch / n3d_tumor_labels are bit-exact against the reference's own functions (golden) and the oracle."""
import numpy as np
import pytest
import torch

import golden_common as gc
from oracle import post_step as ps

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout", ["ncdhw", "ndhwc"])
def test_stitch_matches_reference(golden, layout):
    from nas_3d_unet_amd import poststep as hp
    g = golden("poststep")
    patches, corners, shape = gc.poststep_patches()
    t = torch.from_numpy(np.stack(patches)).cuda()
    if layout == "ndhwc":
        t = t.permute(0, 2, 3, 4, 1).contiguous().permute(0, 4, 1, 2, 3)   # the layout the net's head produces
    out = hp.stitch(t, corners, shape[1:])
    assert out.dtype == torch.float64 and np.array_equal(out.cpu().numpy(), g["stitch/out"])
    # placed inside a larger full image (prediction.py:141-147)
    full = hp.stitch(t, corners, shape[1:], full_shape=(20, 15, 16), origin=(3, 2, 1))
    ref = np.zeros((3, 20, 15, 16))
    ref[:, 3:16, 2:13, 1:13] = g["stitch/out"]
    assert np.array_equal(full.cpu().numpy(), ref)


def test_tumor_labels_match_reference(golden):
    from nas_3d_unet_amd import poststep as hp
    g = golden("poststep")
    pred = torch.from_numpy(gc.poststep_pred()).cuda()
    assert np.array_equal(hp.tumor_labels(pred, 0.5, True).cpu().numpy(), g["tumor/inclusive"])
    assert np.array_equal(hp.tumor_labels(pred, 0.5, False).cpu().numpy(), g["tumor/exclusive"])
    assert np.array_equal(hp.tumor_labels(pred, 0.3, False).cpu().numpy(), g["tumor/exclusive_t03"])


def test_stitch_large_vs_oracle():
    from nas_3d_unet_amd import poststep as hp
    rng = np.random.default_rng(8)
    shape = (3, 70, 61, 66)
    corners = [tuple(int(v) for v in rng.integers(-20, 60, 3)) for _ in range(40)]
    patches = [rng.uniform(0, 1, (3, 32, 32, 32)).astype(np.float32) for _ in corners]
    out = hp.stitch(torch.from_numpy(np.stack(patches)).cuda(), corners, shape[1:])
    assert np.array_equal(out.cpu().numpy(), ps.stitch(patches, corners, shape))
    lab = hp.tumor_labels(out, 0.5, False).cpu().numpy()
    assert np.array_equal(lab, ps.tumor_labels(out.cpu().numpy(), 0.5, False))


def test_predictor_batched_equals_patch_by_patch():
    """Whole-volume inference: batched forward + device stitch == the reference's patch-by-patch loop (prediction.py:120-148)
    driven through the same net, including its rule that all-zero patches predict zeros."""
    from nas_3d_unet_amd import searched
    from nas_3d_unet_amd.predict import Predictor, patching
    from _util import fill_module
    gene = searched.Genotype(down=[("down_conv", 0), ("down_dil_conv", 1), ("down_conv", 1), ("conv", 2), ("dil_conv", 2), ("conv", 3)],
                             up=[("conv", 0), ("up_conv", 1), ("up_conv", 1), ("dil_conv", 2), ("conv", 3), ("up_dil_conv", 1)])
    net = searched.SearchedNet(4, 4, 3, 2, 3, True, gene)
    fill_module(net)
    net = net.cuda().eval()
    rng = np.random.default_rng(5)
    vol = rng.standard_normal((4, 40, 21, 30)).astype(np.float32)
    vol[:, :, :, 18:] = 0     # some patches are entirely empty
    dv = torch.from_numpy(vol).cuda()
    P = 16
    pr = Predictor(net, patch=P, batch=5)
    out = pr.predict(dv, overlap=4).cpu().numpy()
    # reference-style loop: host crop, one patch at a time, host stitch (oracle)
    from oracle import data_step as ds
    corners = [tuple(int(v) for v in c) for c in patching((40, 21, 30), (P, P, P), overlap=4)]
    preds = []
    with torch.no_grad():
        for c in corners:
            data = ds.crop_zero_pad(vol, c, P)
            if np.all(data == 0):
                preds.append(np.zeros((3, P, P, P), dtype=np.float32))
                continue
            preds.append(net(torch.from_numpy(data[None]).cuda())[0].cpu().numpy())
    ref = ps.stitch(preds, corners, (3, 40, 21, 30))
    assert any(np.all(p == 0) for p in preds)
    assert np.abs(out - ref).max() < 2e-6     # batch-of-5 vs batch-of-1 forward: GroupNorm is per sample, kernels pick other tilings
    lab = pr.tumor(dv, overlap=4).cpu().numpy()
    assert np.mean(lab == ps.tumor_labels(ref, 0.5, True)) > 0.999
