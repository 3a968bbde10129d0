"""Step after the hot path (prediction.py:120-170): stitch the per-patch predictions of the searched net
into the brain-wide volume with mean blending (patches.py:172-207), place it in the full image, and fuse the three
sigmoid channels into a label volume -- on the device, from the layout the net produces (no host round trip per patch
as in prediction.py:132-138).  Host logic here: argument plumbing only."""
from __future__ import annotations

import torch

from . import _lib
from . import kernels as K
from ._lib import N3DError, check


def stitch(patches, corners, box_shape, full_shape=None, origin=(0, 0, 0)):
    """patches: (B, C, P, P, P) fp32 device tensor (any layout the ops produce); corners: B corners on the brain-wide grid;
    box_shape: (X, Y, Z) of the brain-wide box; full_shape/origin: optional full image the box is written into
    (prediction.py:141-147).  Returns a float64 (C, FX, FY, FZ) device tensor."""
    if not (isinstance(patches, torch.Tensor) and patches.is_cuda and patches.dtype == torch.float32 and patches.dim() == 5):
        raise N3DError("stitch: patches must be a (B, C, P, P, P) fp32 tensor on a HIP device")
    B, Cc, P = int(patches.shape[0]), int(patches.shape[1]), int(patches.shape[2])
    if patches.shape[3] != P or patches.shape[4] != P:
        raise N3DError("stitch: cubic patches expected")
    if len(corners) != B:
        raise N3DError("stitch: one corner per patch")
    st = K._bcv_strides(patches)
    if st is None:
        patches = patches.contiguous()
        st = K._bcv_strides(patches)
    sb, sc, sv = st
    X, Y, Z = (int(v) for v in box_shape)
    FX, FY, FZ = (int(v) for v in (full_shape if full_shape is not None else box_shape))
    cor = torch.tensor([[int(c) for c in cr] for cr in corners], dtype=torch.int32, device=patches.device)
    out = torch.zeros((Cc, FX, FY, FZ), dtype=torch.float64, device=patches.device)
    check(_lib.load().n3d_stitch(K.ptr(patches), sb, sc, sv, Cc, P, K.ptr(cor), B, X, Y, Z, K.ptr(out), FX, FY, FZ, int(origin[0]), int(origin[1]),
                                 int(origin[2]), K.stream_ptr()), "n3d_stitch")
    return out


def tumor_labels(pred, threshold=0.5, inclusive_label=False):
    """pred: (3, X, Y, Z) float64 device tensor (stitch output) -> uint8 (X, Y, Z) labels {0, 1, 2, 4} (prediction.py:150-170)"""
    if not (isinstance(pred, torch.Tensor) and pred.is_cuda and pred.dtype == torch.float64 and pred.dim() == 4 and pred.shape[0] == 3
            and pred.is_contiguous()):
        raise N3DError("tumor_labels: pred must be a contiguous (3, X, Y, Z) float64 tensor on a HIP device")
    out = torch.empty(tuple(pred.shape[1:]), dtype=torch.uint8, device=pred.device)
    check(_lib.load().n3d_tumor_labels(K.ptr(pred), out.numel(), float(threshold), 1 if inclusive_label else 0, K.ptr(out), K.stream_ptr()),
          "n3d_tumor_labels")
    return out
