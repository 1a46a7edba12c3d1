"""Data step on the GPU (depgan_data_prep_subject through the C ABI) against the oracle's NumPy restatement of
GT:93-146, 667-760: bit-exact, like everything the reference computes in plain fp32 elementwise arithmetic."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _vols(rng, X, Y, Z, int_flair=True):
    p1 = rng.normal(0.2, 0.5, size=(X, Y, Z)).astype(np.float32)
    p2 = rng.normal(0.2, 0.5, size=(X, Y, Z)).astype(np.float32)
    f1 = rng.uniform(0, 3000, size=(X, Y, Z))
    f1 = f1.astype(np.int16) if int_flair else f1.astype(np.float64)
    i1 = (rng.uniform(size=(X, Y, Z)) > 0.3).astype(np.uint8)
    i2 = (rng.uniform(size=(X, Y, Z)) > 0.3).astype(np.float32)
    s1 = (rng.uniform(size=(X, Y, Z)) > 0.9).astype(np.uint8)
    s2 = (rng.uniform(size=(X, Y, Z)) > 0.9).astype(np.int16)
    return p1, f1, i1, s1, p2, i2, s2


@pytest.mark.parametrize("shape,nicg,sl", [((256, 256, 6), 2, (True, True)), ((256, 256, 3), 1, (True, False)),
                                           ((40, 50, 3), 2, (False, True)), ((33, 31, 2), 2, (False, False)),
                                           ((64, 96, 5), 1, (False, False))])
def test_prep_subject_bit_exact(shape, nicg, sl):
    import torch
    from dep_gan_im_amd import data as dgdata
    from oracle import data_oracle as do
    rng = np.random.default_rng(sum(shape) + nicg)
    p1, f1, i1, s1, p2, i2, s2 = _vols(rng, *shape, int_flair=(shape[0] % 2 == 0))
    s1 = s1 if sl[0] else None
    s2 = s2 if sl[1] else None
    ex, ey = do.prep_subject(p1, f1 if nicg == 2 else None, i1, s1, p2, i2, s2, nicg)
    gx, gy = dgdata.prep_subject(p1, f1 if nicg == 2 else None, i1, s1, p2, i2, s2, nicg=nicg)
    torch.cuda.synchronize()
    assert tuple(gx.shape) == ex.shape and tuple(gy.shape) == ey.shape
    assert np.array_equal(gx.cpu().numpy(), ex)
    assert np.array_equal(gy.cpu().numpy(), ey)


def test_training_set_from_files(tmp_path):
    """End to end: NIfTI files on disk -> stacked slices in HBM -> split; one subject without a stroke-lesion file,
    one whose wmh_prob_1tp file is missing (skipped, GT:666)."""
    import torch
    from dep_gan_im_amd import data as dgdata
    from dep_gan_im_amd import nifti
    from oracle import data_oracle as do
    rng = np.random.default_rng(7)
    subs, ex, ey = [], [], []
    for k, Z in enumerate((4, 6, 5, 3)):
        vols = _vols(rng, 64, 64, Z)
        names = []
        for tag, v in zip(("p1", "f1", "i1", "s1", "p2", "i2", "s2"), vols):
            path = str(tmp_path / ("%s_%d.nii.gz" % (tag, k)))
            skip = (k == 1 and tag == "s1") or (k == 2 and tag == "p1")
            if not skip:
                nifti.save(path, v)
            names.append(path)
        subs.append(dgdata.SubjectFiles(names[0], names[1], names[2], names[3], names[4], names[5], names[6]))
        if k != 2:
            p1, f1, i1, s1, p2, i2, s2 = vols
            a, b = do.prep_subject(p1, f1, i1, None if k == 1 else s1, p2, i2, s2, 2)
            ex.append(a)
            ey.append(b)
    ex, ey = np.concatenate(ex, 0), np.concatenate(ey, 0)
    seen = []
    gx, gy = dgdata.load_training_set(subs, nicg=2, progress=lambda s, shp: seen.append(shp))
    assert len(seen) == 3 and tuple(gx.shape) == ex.shape == (13, 64, 64, 2)
    assert np.array_equal(gx.cpu().numpy(), ex) and np.array_equal(gy.cpu().numpy(), ey)
    # split + shuffle (GT:738-760)
    oxt, oxv, oyt, oyv = do.split_and_shuffle(ex, ey, seed_shuffle=5)
    xt, xv, yt, yv = dgdata.split_and_shuffle(gx, gy, rng=np.random.RandomState(5))
    for g, o in ((xt, oxt), (xv, oxv), (yt, oyt), (yv, oyv)):
        assert np.array_equal(g.cpu().numpy(), o)


def test_bad_arguments():
    import torch
    from dep_gan_im_amd import _lib
    from dep_gan_im_amd import data as dgdata
    v = np.zeros((8, 8, 2), np.float32)
    with pytest.raises(ValueError):
        dgdata.prep_subject(v, None, v, None, v, v, None, nicg=2)          # nicg = 2 without FLAIR
    with pytest.raises(ValueError):
        dgdata.prep_subject(v, v, v[:4], None, v, v, None, nicg=2)         # shape mismatch
    lib = _lib.load()
    assert lib.depgan_data_prep_subject(None, None, None, None, None, None, None, 8, 8, 2, 2, None, None, None,
                                        None) != 0
    assert b"data_prep_subject" in lib.depgan_last_error()
