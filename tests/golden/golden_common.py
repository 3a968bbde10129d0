"""Case lists and seeded inputs shared by make_golden.py (reference side) and the tests.

Inputs are regenerated from (case key -> crc32 -> numpy PCG64 stream), so fixtures only
need to store expected outputs."""
import zlib

import numpy as np

B = 2  # batch used by all per-op cases

ALL_PRIMS = ["identity", "se_conv", "dil_conv", "dep_conv", "conv", "avg_pool", "max_pool",
             "down_se_conv", "down_dil_conv", "down_dep_conv", "down_conv",
             "up_se_conv", "up_dep_conv", "up_conv", "up_dil_conv"]
_WIDE_SUBSET = ["identity", "conv", "dil_conv", "down_conv", "up_conv", "down_se_conv", "up_dep_conv",
                "dep_conv", "max_pool"]


def _rng(key, salt=0):
    return np.random.default_rng([zlib.crc32(key.encode()), salt])


def case_input(key, shape):
    return _rng(key, 1).standard_normal(shape).astype(np.float32)


def case_cotangent(key, shape):
    return _rng(key, 2).standard_normal(shape).astype(np.float32)


def case_alpha(key, n):
    a = _rng(key, 3).standard_normal(n)
    e = np.exp(a - a.max())
    return (e / e.sum()).astype(np.float32)


def case_alpha_matrix(key, rows, n):
    a = _rng(key, 4).standard_normal((rows, n))
    e = np.exp(a - a.max(axis=1, keepdims=True))
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def case_probs(key, shape):
    return _rng(key, 5).uniform(0.01, 0.99, shape).astype(np.float32)


def case_targets(key, shape):
    return (_rng(key, 6).uniform(0, 1, shape) < 0.3).astype(np.float32)


def prim_key(name, c):
    return "prim/%s/c%d" % (name, c)


def prim_cases():
    """(name, C, input spatial shape).  Even, non-cubic shapes; C=32/64 exercise multi-group GN."""
    shapes = {4: (4, 6, 8), 8: (8, 6, 4), 16: (4, 4, 6), 32: (4, 4, 4), 64: (2, 2, 2)}
    cases = []
    for c in (4, 8, 16):
        for n in ALL_PRIMS:
            cases.append((n, c, shapes[c]))
    for c in (32, 64):
        for n in _WIDE_SUBSET:
            cases.append((n, c, shapes[c]))
    return cases


def convops_cases():
    """(key, ConvOps kwargs, Cin, input spatial shape) for stems / preprocess / head."""
    return [
        ("convops/stem0", dict(out_channels=12, kernel_size=1, ops_order="weight_norm"), 4, (4, 6, 8)),
        ("convops/stem1", dict(out_channels=12, kernel_size=3, stride=2, ops_order="weight_norm"), 4, (4, 6, 8)),
        ("convops/pre0_down", dict(out_channels=8, kernel_size=1, stride=2, ops_order="act_weight_norm"), 12, (4, 6, 8)),
        ("convops/pre1", dict(out_channels=16, kernel_size=1, ops_order="act_weight_norm"), 24, (4, 4, 6)),
        ("convops/pre_wide", dict(out_channels=64, kernel_size=1, ops_order="act_weight_norm"), 192, (2, 2, 2)),
        ("convops/pre_wide2", dict(out_channels=32, kernel_size=1, ops_order="act_weight_norm"), 96, (4, 2, 2)),
        ("convops/head", dict(out_channels=3, kernel_size=1, ops_order="weight"), 12, (4, 6, 8)),
        ("convops/plain_k3_4to8", dict(out_channels=8, kernel_size=3, ops_order="weight_norm_act"), 4, (4, 6, 8)),
    ]


def mixed_cases():
    """(key, C, stride, transposed, input spatial shape)"""
    return [
        ("mixed/norm/c8", 8, 1, False, (4, 6, 8)),
        ("mixed/down/c8", 8, 2, False, (4, 6, 8)),
        ("mixed/up/c8", 8, 2, True, (4, 6, 4)),
        ("mixed/norm/c4", 4, 1, False, (4, 4, 6)),
        ("mixed/up/c16", 16, 2, True, (2, 4, 4)),
        ("mixed/down/c32", 32, 2, False, (4, 4, 4)),
    ]


def cell_cases():
    """(key, c0, c1, c_node, downward, x0 spatial, x1 spatial).
    down: x0 is one level finer than x1 (preprocess0 has stride 2); up: x1 is one level coarser."""
    return [
        ("cell/down", 12, 12, 8, True, (8, 8, 8), (4, 4, 4)),
        ("cell/up", 24, 48, 8, False, (4, 4, 8), (2, 2, 4)),
        ("cell/down_wide", 24, 48, 32, True, (8, 8, 4), (4, 4, 2)),
    ]


def net_cases():
    """(key, kind, genotype name, depth, patch size, batch, adam steps)"""
    return [
        ("net/searched/G_CONV/d4s32", "searched", "G_CONV", 4, 32, 2, 3),
        ("net/searched/G_ALL/d4s32", "searched", "G_ALL", 4, 32, 2, 3),
        ("net/searched/G_CONV/d2s16", "searched", "G_CONV", 2, 16, 2, 0),
        ("net/supernet/d2s16", "supernet", None, 2, 16, 2, 3),
        ("net/supernet/d4s32", "supernet", None, 4, 32, 1, 0),
        ("net/searched/G_CONV/d4s64", "searched", "G_CONV", 4, 64, 1, 0),
    ]


def net_batch(key, batch, size):
    """Parity batches: N(0,1) inputs and Bernoulli(0.3) targets."""
    x = _rng(key, 7).standard_normal((batch, 4, size, size, size)).astype(np.float32)
    t = (_rng(key, 8).uniform(0, 1, (batch, 3, size, size, size)) < 0.3).astype(np.float32)
    return x, t


def case_drop_gate(key, batch, channels, p):
    """Dropout3d gate (prim_ops.py:66,72-73): Bernoulli(1 - p) per (sample, channel), kept channels scaled by 1/(1-p)"""
    keep = _rng(key, 9).uniform(0, 1, (batch, channels)) >= p
    return (keep / (1.0 - p)).astype(np.float32)


def net2_cases():
    """second group of whole-net cases (nets2.npz): (key, kind, genotype name, depth, patch size, batch, options)
    options: drop = head Dropout3d rate applied with case_drop_gate (train-mode head); wshare = normal_w_share (nas.py:109-113)"""
    return [
        ("net2/searched/G_CONV/d4s32/drop", "searched", "G_CONV", 4, 32, 2, dict(drop=0.5)),
        ("net2/supernet/d2s16/drop", "supernet", None, 2, 16, 2, dict(drop=0.1)),
        ("net2/supernet/d2s16/wshare", "supernet", None, 2, 16, 2, dict(wshare=True)),
        ("net2/supernet/d4s64", "supernet", None, 4, 64, 1, dict()),
    ]


def search_cases():
    """search-step trajectories (search.py:211-238 written out with the reference modules and torch.optim.Adam):
    (key, depth, patch size, batch, steps)"""
    return [("search/d4s32", 4, 32, 2, 2)]


def search_batches(key, batch, size):
    (x, t), (vx, vt) = net_batch(key + "/train", batch, size), net_batch(key + "/val", batch, size)
    return x, t, vx, vt


def search_bench_case():
    """BASELINE configs[2] as bench.py times it (search.py:211-238; nas.py:50-52 head Dropout3d(0.1) in train mode):
    (key, depth, patch size, batch per pass, steps, dropout rate) -- fixture search64.npz"""
    return ("search/d4s64/b2/drop", 4, 64, 2, 2, 0.1)


def search_drop_gates(key, steps, batch, channels, p):
    """the head's Dropout3d masks of a search run made explicit: gates[step][0 = architecture pass, 1 = weight pass], each a
    (batch, channels) case_drop_gate with one to four dropped channels (the first salt that has them: at p = 0.1 a fresh draw
    of 24 drops nothing 8 % of the time, and a mask that drops nothing tests nothing)"""
    out = []
    for s in range(steps):
        row = []
        for name in ("arch", "weight"):
            for salt in range(64):
                g = case_drop_gate("%s/step%d/%s/%d" % (key, s, name, salt), batch, channels, p)
                if 1 <= int((g == 0).sum()) <= 4:
                    break
            else:
                raise AssertionError("no usable gate")
            row.append(g)
        out.append(tuple(row))
    return out


def dice_cases():
    return [("dice/a", (2, 3, 4, 6, 8)), ("dice/b", (1, 3, 16, 16, 16)), ("dice/c", (3, 3, 2, 2, 2))]


def geno_cases():
    return ["geno/%d" % i for i in range(8)]


# ---- data step (patch crop + cube isometry + label expansion) ---------------------------------
DATASTEP_PATCH = (6, 6, 6)


def permutation_keys():
    """the 48 keys ((rotate_y, rotate_z), flip_x, flip_y, flip_z, transpose) of augment.py:73-92, in a fixed order"""
    import itertools
    return sorted(itertools.product(itertools.combinations_with_replacement(range(2), 2), range(2), range(2), range(2), range(2)))


def datastep_cube():
    """(2, 5, 5, 5) float32 cube with distinct entries"""
    return np.arange(2 * 5 * 5 * 5, dtype=np.float32).reshape(2, 5, 5, 5) * 0.5 - 7.0


def datastep_volume():
    """(4, 11, 9, 10) float32 image and (1, 11, 9, 10) uint8 truth with BraTS labels {0, 1, 2, 4}"""
    rng = np.random.default_rng(77)
    vol = rng.standard_normal((4, 11, 9, 10)).astype(np.float32)
    truth = rng.choice(np.array([0, 0, 1, 2, 4], dtype=np.uint8), size=(1, 11, 9, 10))
    return vol, truth


def datastep_corners():
    """patch corners inside, partly outside (both sides) and mostly outside the volume"""
    return [(0, 0, 0), (5, 3, 4), (-2, 1, 6), (7, -3, -1), (-4, -4, -4), (9, 7, 8), (2, 2, 2), (-1, 5, -2)]


def datastep_batch_keys():
    ks = permutation_keys()
    return [ks[i] for i in (0, 5, 11, 17, 23, 30, 41, 47)]


# ---- stitching + label fusion -----------------------------------------------------------------
def poststep_patches():
    """12 (3, 6, 6, 6) float32 patches on a (3, 13, 11, 12) brain-wide grid: overlapping, some hanging over the border on
    either side, one voxel region covered by nothing"""
    rng = np.random.default_rng(91)
    corners = [(0, 0, 0), (4, 0, 0), (7, 0, 0), (0, 5, 0), (4, 5, 0), (7, 5, 0), (0, 0, 6), (4, 0, 6), (-2, 3, 7), (9, 7, 8), (3, -1, -3), (5, 4, 5)]
    patches = [rng.uniform(0, 1, (3, 6, 6, 6)).astype(np.float32) for _ in corners]
    return patches, corners, (3, 13, 11, 12)


def poststep_pred():
    """(3, 9, 8, 7) probabilities incl. exact ties and values exactly at the threshold"""
    rng = np.random.default_rng(92)
    p = rng.uniform(0, 1, (3, 9, 8, 7))
    p[:, 0, 0, :] = 0.5
    p[0, 1, :, 0] = p[1, 1, :, 0] = 0.75   # ties between channels 0 and 1 above the threshold
    p[1, 2, :, 1] = p[2, 2, :, 1] = 0.9
    p[:, 3, 3, :] = 0.8                    # all three above the threshold
    return p


def patching_cases():
    """(img_shape, patch_shape, overlap, both_ps) for patches.patching"""
    return [((155, 173, 140), (64, 64, 64), None, False), ((155, 173, 140), (64, 64, 64), 16, False),
            ((155, 173, 140), (64, 64, 64), (8, 16, 32), True), ((60, 64, 130), (64, 64, 64), None, False),
            ((128, 128, 128), (128, 128, 128), 0, True), ((200, 90, 65), (128, 128, 128), 32, False)]
