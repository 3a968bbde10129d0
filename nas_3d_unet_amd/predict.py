"""Whole-volume inference (prediction.py:120-148): patch corners of the reference's two patching strategies
(patches.py:9-70), BATCHED forward passes of the searched net on patches cropped on the device (the reference runs them
one by one through the host, prediction.py:132-138), device stitching and label fusion.  Host logic only; the kernels
are n3d_patch_batch, the net's own ops, n3d_stitch and n3d_tumor_labels."""
from __future__ import annotations

import itertools

import numpy as np
import torch

from . import datastep, poststep
from ._lib import N3DError


def _grid(start, stop, step):
    """corner grid of np.mgrid[start:stop:step] per axis, first axis slowest, truncated to integers (patches.py:72-74)"""
    axes = [np.arange(start[d], stop[d], step[d]) for d in range(3)]
    return np.asarray(list(itertools.product(*axes)), dtype=np.float64).reshape(-1, 3).astype(np.int64)


def _autofit(img, patch):
    """least number of patches covering the image symmetrically + the central cube (patches.py:9-34)"""
    n = np.ceil(img / patch)
    start, step = np.zeros(3), np.zeros(3)
    for d in range(3):
        if n[d] == 1:
            start[d] = -(patch[d] - img[d]) // 2
            step[d] = patch[d]
        else:
            overlap = np.floor(n[d] * patch[d] - img[d]) / (n[d] - 1)
            overflow = n[d] * patch[d] - (n[d] - 1) * overlap - img[d]
            start[d] = -overflow // 2
            step[d] = patch[d] - overlap
    stop = start + n * step
    return np.vstack((_grid(start, stop, step), (img - patch) // 2))


def patching(img_shape, patch_shape, overlap=None, both_ps=False):
    """bottom-left patch corners, as patches.patching (patches.py:36-70)"""
    img, patch = np.asarray(img_shape), np.asarray(patch_shape)
    auto = _autofit(img, patch)
    if overlap is None:
        return auto
    ov = np.asarray([overlap] * 3) if isinstance(overlap, int) else np.asarray(overlap)
    n = np.ceil(img / (patch - ov))
    overflow = patch * n - (n - 1) * ov - img
    start = -overflow // 2
    step = patch - ov
    stop = start + n * step
    ol = np.vstack(((img - patch) // 2, _grid(start, stop, step)))
    return np.vstack((auto, ol)) if both_ps else ol


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
