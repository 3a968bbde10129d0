"""CPU: the data-step oracle (oracle/data_step.py) and the host-side key decoding of nas_3d_unet_amd.datastep against
the golden vectors produced by the reference's own functions (tests/golden/make_golden.py datastep)."""
import numpy as np

import golden_common as gc
from oracle import data_step as ds


def test_all_48_isometries(golden):
    g = golden("datastep")
    keys = gc.permutation_keys()
    assert len(keys) == 48 and len(set(keys)) == 48
    assert np.array_equal(g["perm/keys"], np.array([[k[0][0], k[0][1], k[1], k[2], k[3], k[4]] for k in keys]))
    cube = gc.datastep_cube()
    seen = set()
    for i, k in enumerate(keys):
        perm, flip = ds.isometry_of_key(k)
        assert np.array_equal(ds.apply_isometry(cube, perm, flip), g["perm/out"][i]), k
        seen.add((tuple(perm), tuple(flip)))
    assert len(seen) == 48  # the keys really are the 48 distinct isometries of the cube


def test_host_key_decoding_equals_oracle():
    from nas_3d_unet_amd import datastep as hd
    assert hd.generate_permutation_keys() == set(gc.permutation_keys())
    for k in gc.permutation_keys():
        assert hd.isometry_of_key(k) == ds.isometry_of_key(k)


def test_crop_zero_pad_and_labels(golden):
    g = golden("datastep")
    vol, truth = gc.datastep_volume()
    P = gc.DATASTEP_PATCH[0]
    for i, c in enumerate(gc.datastep_corners()):
        assert np.array_equal(ds.crop_zero_pad(vol, c, P), g["crop/out"][i]), c
    tp = np.stack([ds.crop_zero_pad(truth, c, P) for c in gc.datastep_corners()])
    assert np.array_equal(ds.expand_labels(tp, True), g["labels/inclusive"])
    assert np.array_equal(ds.expand_labels(tp, False), g["labels/exclusive"])
    # the reference's quirk: label 4 is NOT part of the inclusive whole-tumour channel
    assert ds.expand_labels(np.full((1, 1, 2, 2, 2), 4), True)[0, 1].sum() == 0


def test_whole_batch(golden):
    g = golden("datastep")
    vol, truth = gc.datastep_volume()
    x, y = ds.data_step(vol, truth, gc.datastep_corners(), gc.datastep_batch_keys(), gc.DATASTEP_PATCH[0])
    assert np.array_equal(x, g["batch/x"]) and np.array_equal(y, g["batch/y"].astype(np.float32))
