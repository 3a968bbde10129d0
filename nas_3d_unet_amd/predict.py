"""Whole-volume inference (prediction.py:120-148): patch corners of the reference's two patching strategies
(patches.py:9-70), BATCHED forward passes of the searched net on patches cropped on the device (the reference runs them
one by one through the host, prediction.py:132-138), device stitching and label fusion.  Host logic only; the kernels
are n3d_patch_batch, the net's own ops, n3d_stitch and n3d_tumor_labels."""
from __future__ import annotations

import numpy as np
import torch

from . import datastep, poststep
from ._lib import N3DError


def _corner_lattice(first, pitch, count):
    """All corners first + i * pitch, i < count per axis (first axis slowest), truncated towards zero: the lattice np.mgrid walks for
    the reference (patches.py:72-74).  first / pitch may be fractional -- the reference spreads the patches evenly with a real-valued
    pitch and lets the integer cast place them."""
    axes = [first[d] + np.arange(int(count[d])) * pitch[d] for d in range(3)]
    return np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, 3).astype(np.int64)


def _even_cover(img, patch):
    """The auto-fitting strategy of patches.py:9-34 in closed form, all three axes at once.  An axis needs n = ceil(img / patch) patches.
    n = 1: the one patch is centred (it overhangs by patch - img, the odd voxel on the low side).  n > 1: the n patches share their
    total excess n * patch - img evenly, i.e. neighbours overlap by excess / (n - 1) and the pitch is patch - that; what is left
    over after the real-valued division (zero up to rounding, and the reference keeps the rounding) is split around the image like
    the overhang.  Returns (first corner, pitch, count) per axis."""
    img, patch = img.astype(np.float64), patch.astype(np.float64)
    n = np.ceil(img / patch)
    lone = n == 1
    gaps = np.where(lone, 1.0, n - 1.0)
    share = np.floor(n * patch - img) / gaps                      # overlap between neighbours (n > 1)
    rest = n * patch - (n - 1.0) * share - img                    # excess the even split leaves
    first = np.where(lone, (-(patch - img)) // 2, (-rest) // 2)
    pitch = np.where(lone, patch, patch - share)
    # the reference walks mgrid[first : first + n * pitch : pitch]: ceil(((first + n * pitch) - first) / pitch) corners
    count = np.ceil(((first + n * pitch) - first) / pitch)
    return first, pitch, count


def _fixed_overlap_cover(img, patch, ov):
    """The given-overlap strategy of patches.py:59-67: pitch patch - ov, as many patches as it takes, the overhang split around the image"""
    img, patch, ov = img.astype(np.float64), patch.astype(np.float64), ov.astype(np.float64)
    pitch = patch - ov
    n = np.ceil(img / pitch)
    first = (-(patch * n - (n - 1.0) * ov - img)) // 2
    count = np.ceil(((first + n * pitch) - first) / pitch)
    return first, pitch, count


def patching(img_shape, patch_shape, overlap=None, both_ps=False):
    """bottom-left patch corners as the reference's patches.patching lists them (patches.py:36-70): the evenly spread cover followed
    by the centred cube; with `overlap` the centred cube followed by the fixed-overlap cover; with both_ps the first list, then the second."""
    img, patch = np.asarray(img_shape), np.asarray(patch_shape)
    centre = ((img - patch) // 2).reshape(1, 3).astype(np.int64)
    even = np.concatenate((_corner_lattice(*_even_cover(img, patch)), centre))
    if overlap is None:
        return even
    ov = np.full(3, overlap) if isinstance(overlap, int) else np.asarray(overlap)
    fixed = np.concatenate((centre, _corner_lattice(*_fixed_overlap_cover(img, patch, ov))))
    return np.concatenate((even, fixed)) if both_ps else fixed


class Predictor:
    """prediction.py:120-170 on the device: `volume` is the brain-wide crop (C, X, Y, Z) resident in HBM."""

    def __init__(self, model, patch=64, batch=8):
        self.model, self.patch, self.batch = model, int(patch), int(batch)

    @torch.no_grad()
    def predict(self, volume, overlap=None, both_ps=False, full_shape=None, origin=(0, 0, 0)):
        """-> float64 (n_labels, FX, FY, FZ) probabilities (the brain-wide box stitched and placed in the full image)"""
        if not (isinstance(volume, torch.Tensor) and volume.is_cuda and volume.dim() == 4):
            raise N3DError("Predictor: volume must be a (C, X, Y, Z) tensor on a HIP device")
        P = self.patch
        box = tuple(int(s) for s in volume.shape[1:])
        corners = [tuple(int(v) for v in c) for c in patching(box, (P, P, P), overlap, both_ps)]
        was_training = self.model.training
        self.model.eval()
        preds = []
        try:
            for i in range(0, len(corners), self.batch):
                chunk = corners[i:i + self.batch]
                x, _ = datastep.patch_batch(volume, None, chunk, [None] * len(chunk), P)
                y = self.model(x)
                # an all-zero patch is not run through the model by the reference: its prediction is zeros (prediction.py:133-135)
                empty = (x.abs().amax(dim=(1, 2, 3, 4)) == 0).view(-1, 1, 1, 1, 1)
                preds.append(torch.where(empty, torch.zeros_like(y), y))
        finally:
            self.model.train(was_training)
        return poststep.stitch(torch.cat(preds), corners, box, full_shape, origin)

    def tumor(self, volume, threshold=0.5, inclusive_label=True, **kw):
        return poststep.tumor_labels(self.predict(volume, **kw), threshold, inclusive_label)
