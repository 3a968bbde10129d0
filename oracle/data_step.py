"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the reference's per-batch data step
 -- the work train.py:117-119 receives from the generator just before the hot path:

  * patch crop with zero padding         patches.py:99-115 (get_patch_from_3d_data), :152-169 (fix_out_of_bound_patch_attempt)
  * one of the 48 cube isometries        augment.py:73-92 (keys), :105-131 (permute_data); same key for data and truth
  * label expansion to 3 channels        generator.py:230-248 (get_multi_class_labels), including the reference's quirk:
                                         `np.logical_or(a, b, c)` takes c as the OUT array, so the inclusive "WT" channel
                                         is labels {1, 2} (label 4 is not included)

Formulation (not the reference's): every isometry is a signed axis permutation, out[i0,i1,i2] = in[s0,s1,s2] with
s_a = i_{perm[a]} or P-1-i_{perm[a]}; `isometry_of_key` derives (perm, flip) by pushing coordinate grids through the
key's steps.  Parity status: PINNED -- tests/golden/datastep.npz holds the outputs of the reference's own functions
(executed from the reference source by tests/golden/make_golden.py) for all 48 keys, 8 crop positions and both label modes.
"""
import itertools

import numpy as np


def permutation_keys():
    """the 48 keys ((rotate_y, rotate_z), flip_x, flip_y, flip_z, transpose) -- augment.py:73-92 -- in sorted order"""
    return sorted(itertools.product(itertools.combinations_with_replacement(range(2), 2), range(2), range(2), range(2), range(2)))


def isometry_of_key(key):
    """(perm, flip): out[i] = in[src(i)], src_a(i) = i[perm[a]] if not flip[a] else P-1-i[perm[a]]   (augment.py:105-131)"""
    (rot_y, rot_z), flip_x, flip_y, flip_z, transpose = key
    # state: for every axis a of the CURRENT array, which original axis it shows and whether reversed
    # represented as src[a] = (orig_axis, reversed) meaning current[..i_a..] reads original at orig_axis index i_a or P-1-i_a
    # we instead track, for each ORIGINAL axis o, the expression in terms of output axes: (out_axis, reversed)
    expr = {0: (0, False), 1: (1, False), 2: (2, False)}  # original axis o index = out index of axis expr[o][0] (maybe reversed)

    def rot90(a, b):
        # new[i_a, i_b] = old[i_b, P-1-i_a]  (np.rot90 k=1 in the plane a -> b): old axis a reads new i_b, old axis b reads P-1-new i_a
        nonlocal expr
        new = {}
        for o, (ax, rev) in expr.items():
            if ax == a:
                new[o] = (b, rev)
            elif ax == b:
                new[o] = (a, not rev)
            else:
                new[o] = (ax, rev)
        expr = new

    def flip(a):
        nonlocal expr
        expr = {o: ((ax, not rev) if ax == a else (ax, rev)) for o, (ax, rev) in expr.items()}

    def swap(a, b):
        nonlocal expr
        m = {a: b, b: a}
        expr = {o: (m.get(ax, ax), rev) for o, (ax, rev) in expr.items()}

    # each step redefines the array; expr maps original axes to (current axis, reversed) -- update with the step's own map
    if rot_y:
        rot90(0, 2)       # spatial axes (1,3) of a (C,X,Y,Z) array = spatial 0 and 2
    if rot_z:
        rot90(1, 2)
    if flip_x:
        flip(0)
    if flip_y:
        flip(1)
    if flip_z:
        flip(2)
    if transpose:
        swap(0, 2)        # .T of a 3-D array reverses the axis order
    perm = [expr[o][0] for o in range(3)]
    flp = [bool(expr[o][1]) for o in range(3)]
    return perm, flp


def apply_isometry(patch, perm, flip):
    """patch (C,P,P,P) -> out[c, i0,i1,i2] = patch[c, s0,s1,s2]"""
    P = patch.shape[1]
    idx = np.meshgrid(np.arange(P), np.arange(P), np.arange(P), indexing="ij")
    src = []
    for a in range(3):
        s = idx[perm[a]]
        src.append(P - 1 - s if flip[a] else s)
    return patch[:, src[0], src[1], src[2]]


def crop_zero_pad(vol, corner, P):
    """vol (C,X,Y,Z), corner (3,) possibly outside -> (C,P,P,P) with zeros outside the volume (patches.py:99-115,152-169)"""
    out = np.zeros((vol.shape[0], P, P, P), dtype=vol.dtype)
    lo = [max(0, -int(c)) for c in corner]
    hi = [min(P, vol.shape[1 + a] - int(corner[a])) for a in range(3)]
    if all(h > l for l, h in zip(lo, hi)):
        out[:, lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]] = vol[:, corner[0] + lo[0]:corner[0] + hi[0], corner[1] + lo[1]:corner[1] + hi[1],
                                                            corner[2] + lo[2]:corner[2] + hi[2]]
    return out


def expand_labels(truth, inclusive=True):
    """truth (B,1,P,P,P) integer labels {0,1,2,4} -> (B,3,P,P,P) int8   (generator.py:230-248, incl. the logical_or quirk)"""
    t = truth[:, 0]
    if inclusive:
        chans = [(t == 1) | (t == 4), (t == 1) | (t == 2), (t == 4)]
    else:
        chans = [(t == 1), (t == 2), (t == 4)]
    return np.stack(chans, axis=1).astype(np.int8)


def data_step(vol, truth, corners, keys, P, inclusive=True):
    """One batch: x (B,C,P,P,P) float32, y (B,3,P,P,P) float32 (train.py:118-119 casts both to float)."""
    xs, ys = [], []
    for corner, key in zip(corners, keys):
        perm, flip = isometry_of_key(key)
        xs.append(apply_isometry(crop_zero_pad(vol, corner, P), perm, flip))
        ys.append(apply_isometry(crop_zero_pad(truth, corner, P), perm, flip))
    y = expand_labels(np.asarray(ys), inclusive)
    return np.asarray(xs, dtype=np.float32), y.astype(np.float32)
