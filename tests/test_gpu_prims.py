"""GPU parity: every registry primitive and every non-registry ConvOps use, HIP path vs
(a) the golden vectors produced by the reference and (b) the CPU oracle.  fp32; tolerances are
relative to max|ref| of each tensor (stated below)."""
import numpy as np
import pytest
import torch

import golden_common as gc
from _util import assert_close, dev, fill_module

pytestmark = pytest.mark.gpu

TOL_FWD = 2e-5   # forward outputs
TOL_DX = 5e-5    # input gradients
TOL_DW = 2e-4    # parameter gradients (long fp32 reductions, different summation order)


@pytest.mark.parametrize("name,c,shape", gc.prim_cases())
def test_prim_vs_golden(golden, name, c, shape):
    from nas_3d_unet_amd import prim_ops
    g = golden("prims")
    key = gc.prim_key(name, c)
    op = fill_module(prim_ops.OPS[name](c), key + ".").cuda()
    x = dev(gc.case_input(key, (gc.B, c) + shape), True)
    y = op(x)
    assert_close(y, g[key + "/y"], TOL_FWD, key + " y")
    r = dev(gc.case_cotangent(key, tuple(y.shape)))
    (y * r).sum().backward()
    assert_close(x.grad, g[key + "/dx"], TOL_DX, key + " dx")
    for n, p in op.named_parameters():
        assert p.grad is not None, n
        assert_close(p.grad, g[key + "/grad/" + key + "." + n], TOL_DW, key + " d" + n)


@pytest.mark.parametrize("key,kw,cin,shape", gc.convops_cases())
def test_convops_vs_golden(golden, key, kw, cin, shape):
    from nas_3d_unet_amd import prim_ops
    g = golden("convops")
    op = fill_module(prim_ops.ConvOps(cin, **kw), key + ".").cuda()
    x = dev(gc.case_input(key, (gc.B, cin) + shape), True)
    y = op(x)
    assert_close(y, g[key + "/y"], TOL_FWD, key + " y")
    r = dev(gc.case_cotangent(key, tuple(y.shape)))
    (y * r).sum().backward()
    assert_close(x.grad, g[key + "/dx"], TOL_DX, key + " dx")
    for n, p in op.named_parameters():
        assert_close(p.grad, g[key + "/grad/" + key + "." + n], TOL_DW, key + " d" + n)


@pytest.mark.parametrize("name", gc.ALL_PRIMS)
def test_prim_vs_oracle_channel_slices(name):
    """Inputs and outputs that are channel slices of wider NDHWC buffers (zero-copy concat path),
    odd batch, C=8 -- checked against the CPU oracle on the same seeded data."""
    from nas_3d_unet_amd import prim_ops
    from oracle import ref_path as orc
    c, B, shape = 8, 3, (6, 4, 8)
    key = "slice/" + name
    op = fill_module(prim_ops.OPS[name](c), key + ".").cuda()
    xn = np.random.default_rng(7).standard_normal((B, 3 * c) + shape).astype(np.float32)
    wide = torch.from_numpy(xn).cuda().contiguous(memory_format=torch.channels_last_3d)
    x = wide[:, c:2 * c].detach().requires_grad_(True)  # strided channel slice
    y = op(x)
    P = orc.make_params(orc.prim_param_specs(key + ".", name, c), requires_grad=True)
    xo = torch.from_numpy(xn[:, c:2 * c].copy()).requires_grad_(True)
    yo = orc.prim_forward(P, key + ".", name, xo)
    assert_close(y, yo, TOL_FWD, key + " y")
    rn = np.random.default_rng(8).standard_normal(tuple(yo.shape)).astype(np.float32)
    (y * torch.from_numpy(rn).cuda()).sum().backward()
    (yo * torch.from_numpy(rn)).sum().backward()
    assert_close(x.grad, xo.grad, TOL_DX, key + " dx")
    for n, p in op.named_parameters():
        assert_close(p.grad, P[key + "." + n].grad, TOL_DW, key + " d" + n)
