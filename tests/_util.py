"""Shared helpers for the parity tests."""
import numpy as np
import torch

from oracle import ref_path as orc


def fill_module(mod, prefix=""):
    """Overwrite every parameter with the closed-form fill keyed by its state-dict name."""
    with torch.no_grad():
        for name, p in mod.named_parameters():
            p.copy_(torch.from_numpy(orc.fill_value(prefix + name, tuple(p.shape))).to(p.dtype))
    return mod


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def assert_close(a, b, tol, what=""):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    if isinstance(b, torch.Tensor):
        b = b.detach().cpu().numpy()
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    e = rel_err(a, b)
    assert e <= tol, "%s: max|err|/max|ref| = %.3e > %.1e" % (what, e, tol)


def dev(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.requires_grad_(grad)
