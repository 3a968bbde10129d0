"""GPU: op-level path used when the reference's unchanged searched.py drives this repo's registry ops --
torch's own `0 + a + b` and `torch.cat(dim=1)` between our autograd nodes (searched.py:45-51), channels-last
strides preserved -- must equal the fused SearchedCell path and the oracle."""
import numpy as np
import pytest
import torch

from _util import assert_close, fill_module
from oracle import ref_path as orc

pytestmark = pytest.mark.gpu


def test_unfused_cell_algebra_equals_fused():
    from nas_3d_unet_amd import searched
    gene = searched.Genotype(*orc.G_ALL)
    cell = fill_module(searched.SearchedCell(3, 12, 12, 8, gene, True), "c.").cuda()
    rng = np.random.default_rng(3)
    x0n = rng.standard_normal((2, 12, 8, 8, 8)).astype(np.float32)
    x1n = rng.standard_normal((2, 12, 4, 4, 4)).astype(np.float32)
    x0, x1 = torch.from_numpy(x0n).cuda().requires_grad_(True), torch.from_numpy(x1n).cuda().requires_grad_(True)
    y = cell(x0, x1)
    y.square().sum().backward()
    g_fused = {n: p.grad.clone() for n, p in cell.named_parameters()}
    dx0 = x0.grad.clone()
    for p in cell.parameters():
        p.grad = None
    # the reference's own forward body, verbatim semantics, on our ops
    a0, a1 = torch.from_numpy(x0n).cuda().requires_grad_(True), torch.from_numpy(x1n).cuda().requires_grad_(True)
    xs = [cell.preprocess0(a0), cell.preprocess1(a1)]
    i = 0
    for node in range(3):
        outs = []
        for _ in range(2):
            outs.append(cell._ops[i](xs[cell.genolist[i][1]]))
            i += 1
        xs.append(sum(outs))
    y2 = torch.cat(xs[-3:], dim=1)
    assert_close(y2, y, 1e-6, "forward")
    y2.square().sum().backward()
    assert_close(a0.grad, dx0, 2e-5, "dx0")
    for n, p in cell.named_parameters():
        assert_close(p.grad, g_fused[n], 2e-4, n)
