"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the step AFTER the hot path:

  * stitch: mean-blend of overlapping patch predictions on the brain-wide grid     patches.py:172-207
            (patches may hang over the border and are clipped; voxels no patch covers stay 0)
  * tumor labels: threshold + label fusion of the 3 sigmoid channels                prediction.py:150-170

Formulation (not the reference's): a GATHER -- for every output voxel, the covering patches are visited in list order,
summed in float64 and divided by their count (the reference scatters patch by patch in the same order, float64 too, so
the sums are bit-identical).  Parity status: PINNED -- tests/golden/poststep.npz holds the outputs of the reference's
own functions run from the reference source (make_golden.py; `np.int`/`np.bool`, removed from numpy, are provided as
the builtins they aliased).
"""
import numpy as np


def stitch(patches, corners, shape):
    """patches: list of (C,P,P,P); corners: list of (3,); shape (C,X,Y,Z) -> float64 mean over covering patches"""
    C, X, Y, Z = shape
    P = patches[0].shape[-1]
    out = np.zeros(shape, dtype=np.float64)
    cnt = np.zeros((X, Y, Z), dtype=np.int64)
    gx, gy, gz = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    for patch, corner in zip(patches, corners):
        lx, ly, lz = gx - corner[0], gy - corner[1], gz - corner[2]
        inside = (lx >= 0) & (lx < P) & (ly >= 0) & (ly < P) & (lz >= 0) & (lz < P)
        vals = patch[:, np.clip(lx, 0, P - 1), np.clip(ly, 0, P - 1), np.clip(lz, 0, P - 1)].astype(np.float64)
        out += np.where(inside[None], vals, 0.0)
        cnt += inside
    return out / np.maximum(cnt, 1)[None]


def tumor_labels(pred, threshold=0.5, inclusive_label=False):
    """pred (3,X,Y,Z) -> uint8 label volume {0,1,2,4}   (prediction.py:150-170)"""
    a, b, c = pred[0] >= threshold, pred[1] >= threshold, pred[2] >= threshold
    if inclusive_label:
        t = np.zeros(pred[0].shape, dtype=np.uint8)
        t[b] = 2
        t[a] = 1
        t[c] = 4
        return t
    # exclusive: channels vote; ties between two channels go to the larger probability (the earlier channel on equality)
    t = np.where(a & b, np.where(pred[0] >= pred[1], 1, 2), a * 1 + b * 2).astype(np.int64)
    t5 = c & (t == 1)
    t6 = c & (t == 2)
    t = np.where(t5, np.where(pred[0] >= pred[2], 1, 4), np.where(t6, np.where(pred[1] >= pred[2], 2, 4), t + 4 * c))
    return t.astype(np.uint8)
