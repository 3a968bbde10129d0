"""On-device data step: what the reference's generator hands to the train step (train.py:117-119) --
patch crop with zero padding (patches.py:99-115,152-169), one of the 48 cube isometries per patch
(augment.py:73-131, same key for data and truth) and label expansion (generator.py:230-248) -- produced by ONE
libn3d launch (n3d_patch_batch) from a volume that stays resident in HBM, directly in the layout the hot path reads
(NDHWC input, (B,3,P,P,P) targets: fp32, or the three boolean maps as bytes).  Host logic here: the key set, key -> signed axis permutation, random draws.
"""
from __future__ import annotations

import ctypes as C
import itertools
import random

import torch

from . import _lib
from . import kernels as K
from ._lib import N3DError, PatchDesc, check


def generate_permutation_keys():
    """the 48 keys ((rotate_y, rotate_z), flip_x, flip_y, flip_z, transpose) of augment.py:73-92 (as a set, like the reference)"""
    return set(itertools.product(itertools.combinations_with_replacement(range(2), 2), range(2), range(2), range(2), range(2)))


def random_permutation_key(rng=random):
    """augment.py:95-100"""
    return rng.choice(sorted(generate_permutation_keys()))


def isometry_of_key(key):
    """key -> (perm, flip) with out[i0,i1,i2] = in[s0,s1,s2], s_a = i[perm[a]] or P-1-i[perm[a]] (flip[a]).
    The steps of augment.py:105-131 in order: rot90 in the (x,z) plane, rot90 in the (y,z) plane, flips, transpose (.T)."""
    (rot_y, rot_z), flip_x, flip_y, flip_z, transpose = key
    # src[o] = (output axis whose index feeds original axis o, reversed?)
    src = [(0, False), (1, False), (2, False)]

    def rot90(a, b):  # new[i_a, i_b] = old[i_b, P-1-i_a]
        for o, (ax, rev) in enumerate(src):
            if ax == a:
                src[o] = (b, rev)
            elif ax == b:
                src[o] = (a, not rev)

    if rot_y:
        rot90(0, 2)
    if rot_z:
        rot90(1, 2)
    for a, f in enumerate((flip_x, flip_y, flip_z)):
        if f:
            for o, (ax, rev) in enumerate(src):
                if ax == a:
                    src[o] = (ax, not rev)
    if transpose:
        for o, (ax, rev) in enumerate(src):
            src[o] = ({0: 2, 2: 0}.get(ax, ax), rev)
    return [s[0] for s in src], [bool(s[1]) for s in src]


def patch_batch(vol, truth, corners, keys, patch, inclusive_label=True, target_dtype=torch.float32, out=None):
    """vol: (Cv, X, Y, Z) fp32 device tensor; truth: (X, Y, Z) or (1, X, Y, Z) uint8 device tensor or None;
    corners: B patch corners (may lie outside the volume: zero padding); keys: B isometry keys (None = identity).
    Returns (x, t): x (B, Cv, P, P, P) fp32 in NDHWC storage, t (B, 3, P, P, P) fp32 (None without truth) -- uint8 with
    target_dtype=torch.uint8: generator.py:230-248 yields booleans, and the head passes read bytes (a quarter of the target traffic).
    out=(x, t): write the batch into these tensors (a trainer's input_buffers(): no per-step input copy); t's dtype then decides."""
    if target_dtype not in (torch.float32, torch.uint8):
        raise N3DError("patch_batch: targets are float32 or uint8")
    if not (isinstance(vol, torch.Tensor) and vol.is_cuda and vol.dtype == torch.float32 and vol.dim() == 4 and vol.is_contiguous()):
        raise N3DError("patch_batch: vol must be a contiguous (C, X, Y, Z) fp32 tensor on a HIP device")
    B = len(corners)
    if len(keys) != B or B < 1:
        raise N3DError("patch_batch: need one isometry key per patch corner")
    Cv, X, Y, Z = (int(s) for s in vol.shape)
    P = int(patch)
    tr = None
    if truth is not None:
        tr = truth.reshape(X, Y, Z)
        if not (tr.is_cuda and tr.dtype == torch.uint8 and tr.is_contiguous()):
            raise N3DError("patch_batch: truth must be a contiguous uint8 label volume on the same device")
    descs = (PatchDesc * B)()
    for i, (corner, key) in enumerate(zip(corners, keys)):
        perm, flip = isometry_of_key(key) if key is not None else ([0, 1, 2], [False, False, False])
        descs[i] = PatchDesc((C.c_int32 * 3)(*[int(c) for c in corner]), (C.c_int32 * 3)(*perm), (C.c_int32 * 3)(*[int(f) for f in flip]))
    if out is not None:
        x, t = out
        if tr is None:
            t = None
        if tuple(x.shape) != (B, Cv, P, P, P) or x.dtype != torch.float32 or x.device != vol.device:
            raise N3DError(f"patch_batch: out[0] must be a ({B}, {Cv}, {P}, {P}, {P}) fp32 tensor on the volume's device")
        if t is not None:
            if tuple(t.shape) != (B, 3, P, P, P) or not t.is_contiguous() or t.dtype not in (torch.float32, torch.uint8) or t.device != vol.device:
                raise N3DError(f"patch_batch: out[1] must be a contiguous ({B}, 3, {P}, {P}, {P}) float32 or uint8 tensor on the volume's device")
            target_dtype = t.dtype
    else:
        x = K.empty_ndhwc(B, Cv, P, P, P, vol.device)
        t = torch.empty((B, 3, P, P, P), dtype=target_dtype, device=vol.device) if tr is not None else None
    xv = K.as_view(x)
    if out is not None and xv.t is not x:
        raise N3DError("patch_batch: out[0] must be in NDHWC (channels-last) storage, as K.empty_ndhwc / Trainer.input_buffers() give it")
    flags = (_lib.PATCH_INCLUSIVE if inclusive_label else 0) | (_lib.PATCH_T_U8 if target_dtype == torch.uint8 else 0)
    check(_lib.load().n3d_patch_batch(K.ptr(vol), Cv, K.ptr(tr), X, Y, Z, descs, B, P, flags, xv.p, xv.ld, K.ptr(t),
                                      K.stream_ptr()), "n3d_patch_batch")
    return x, t
