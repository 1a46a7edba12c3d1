"""Data step, CPU side: the NIfTI reader / writer, the oracle's restatement of the reference's
NumPy statements (GT:93-146, 667-760) and the host logic around the device call."""
import gzip
import os
import struct

import numpy as np
import pytest

from dep_gan_im_amd import data as dgdata
from dep_gan_im_amd import nifti
from oracle import data_oracle as do


def test_nifti_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    aff = np.array([[0.9, 0.1, 0, -80], [0, 1.1, 0.2, -100], [0.05, 0, 3.0, -40], [0, 0, 0, 1.0]])
    for dt in (np.float32, np.int16, np.uint8, np.float64):
        v = (rng.normal(size=(9, 7, 4)) * 50).astype(dt)
        for name in ("v.nii", "v.nii.gz"):
            p = str(tmp_path / name)
            nifti.save(p, v, affine=aff, pixdim=[1, 0.9, 1.1, 3.0, 2.5])
            w = nifti.load(p)
            assert w.image.dtype == np.dtype(dt) and np.array_equal(w.image, v)
            assert np.allclose(w.affine, aff.astype(np.float32))
            assert w.dt == pytest.approx(2.5)
            assert w.image.flags["F_CONTIGUOUS"]          # x fastest, like nibabel's array proxy
    with gzip.open(str(tmp_path / "v.nii.gz"), "rb") as f:
        assert struct.unpack("<i", f.read(4))[0] == 348


def test_nifti_scaling_and_big_endian(tmp_path):
    v = np.arange(24, dtype=np.int16).reshape(2, 3, 4, order="F")
    p = str(tmp_path / "s.nii")
    nifti.save(p, v)
    raw = bytearray(open(p, "rb").read())
    struct.pack_into("<2f", raw, 112, 0.5, 10.0)          # scl_slope, scl_inter
    open(p, "wb").write(bytes(raw))
    w = nifti.load(p)
    assert w.image.dtype == np.float64 and np.array_equal(w.image, v * 0.5 + 10.0)   # nibabel get_data() scaling
    # the same volume written big-endian by hand
    hdr = bytearray(352)
    struct.pack_into(">i", hdr, 0, 348)
    struct.pack_into(">8h", hdr, 40, 3, 2, 3, 4, 1, 1, 1, 1)
    struct.pack_into(">hh", hdr, 70, 4, 16)
    struct.pack_into(">8f", hdr, 76, 1, 1, 1, 1, 1, 1, 1, 1)
    struct.pack_into(">3f", hdr, 108, 352.0, 0.0, 0.0)
    hdr[344:348] = b"n+1\x00"
    open(p, "wb").write(bytes(hdr) + v.astype(">i2").tobytes(order="F"))
    w = nifti.load(p)
    assert np.array_equal(w.image, v) and w.image.dtype.byteorder in ("=", "<", "|")
    with pytest.raises(nifti.NiftiError):
        open(p, "wb").write(bytes(hdr)[:100])
        nifti.load(p)


def test_oracle_data_prep_semantics():
    rng = np.random.default_rng(1)
    v = rng.normal(size=(6, 5, 3))
    s = do.data_prep(v)
    assert s.shape == (3, 6, 5, 1) and s.dtype == np.float32
    for z in range(3):
        assert np.array_equal(s[z, :, :, 0], v[:, :, z].astype(np.float32))     # GT:109-113
    back = do.data_prep_save(s)
    assert np.array_equal(back, v.astype(np.float32))      # GT:121-127 undoes the slicing: (X, Y, Z) again
    assert np.array_equal(dgdata.data_prep_save(s), back)


def test_oracle_prep_subject_properties():
    rng = np.random.default_rng(2)
    X, Y, Z = 12, 10, 4
    p1 = rng.normal(0.2, 0.5, size=(X, Y, Z)).astype(np.float32)      # some negative probabilities
    p2 = rng.normal(0.2, 0.5, size=(X, Y, Z)).astype(np.float32)
    f1 = (rng.uniform(0, 3000, size=(X, Y, Z))).astype(np.int16)
    icv = (rng.uniform(size=(X, Y, Z)) > 0.3).astype(np.uint8)
    sl = (rng.uniform(size=(X, Y, Z)) > 0.9).astype(np.uint8)
    x, y2 = do.prep_subject(p1, f1, icv, sl, p2, icv, None, 2)
    assert x.shape == (Z, X, Y, 2) and y2.shape == (Z, X, Y, 1) and x.dtype == np.float32
    assert x[..., 0].min() >= 0 and y2.min() >= 0                      # GT:716-717
    assert x[..., 1].min() == 0.0 and x[..., 1].max() == 1.0           # GT:707, percentile 0
    dead = np.transpose((icv == 0) | (sl == 1), (2, 0, 1))
    assert np.all(x[..., 0][dead] == 0)
    x1, _ = do.prep_subject(p1, None, icv, sl, p2, icv, None, 1)
    assert x1.shape == (Z, X, Y, 1) and np.array_equal(x1[..., 0], x[..., 0])


def test_split_matches_sklearn():
    sk = pytest.importorskip("sklearn.model_selection")
    for n in (50, 101, 777):
        x, y = np.arange(n)[:, None], 2 * np.arange(n)[:, None]
        ref = sk.train_test_split(x, y, test_size=0.02, random_state=42)       # GT:738
        got = do.split_and_shuffle(x, y)
        assert all(np.array_equal(a, b) for a, b in zip(ref, got))


def test_file_lists(tmp_path):
    names = {"wmh_prob_1tp": "p1", "flair_1tp": "f1", "icv_1tp": "i1", "sl_cleaned_1tp": "s1", "wmh_prob_2tp": "p2",
             "icv_2tp": "i2", "sl_cleaned_2tp": "s2"}
    for stem, tag in names.items():
        (tmp_path / ("%s_fold3.txt" % stem)).write_text("".join("/d/%s_%d.nii.gz\n" % (tag, i) for i in range(4)))
    subs = dgdata.training_file_lists(str(tmp_path), 3)
    assert len(subs) == 4 and subs[2] == dgdata.SubjectFiles("/d/p1_2.nii.gz", "/d/f1_2.nii.gz", "/d/i1_2.nii.gz",
                                                            "/d/s1_2.nii.gz", "/d/p2_2.nii.gz", "/d/i2_2.nii.gz",
                                                            "/d/s2_2.nii.gz")
    (tmp_path / "icv_2tp_fold3.txt").write_text("/d/i2_0.nii.gz\n")
    with pytest.raises(ValueError):
        dgdata.training_file_lists(str(tmp_path), 3)


def test_keras_h5_name_mapping():
    """The HDF5 importer is a lookup by Keras layer / weight names (h5py itself is absent here: a nested dict stands in
    for the open file, which is all the mapping touches)."""
    from dep_gan_im_amd import Gen_UNet2D
    from dep_gan_im_amd.models import weights_from_keras_h5
    g = Gen_UNet2D((64, 64, 1), (32, 1), 32, 1, seed=3)
    names = [n for n, _, _ in g._named_table()]
    want = g.get_weights_dict()
    layers = {}
    for n in names:
        layer, w = n.split("/", 1)
        layers.setdefault(layer, {layer: {}})[layer][w + ":0"] = want[n]
    got = weights_from_keras_h5({"model_weights": layers}, names)           # model.save layout (GT:892)
    assert list(got) == names and all(np.array_equal(got[n], want[n]) for n in names)
    got = weights_from_keras_h5(layers, names)                              # save_weights layout
    assert all(np.array_equal(got[n], want[n]) for n in names)
    del layers["conv2d_gen_0"]
    with pytest.raises(KeyError):
        weights_from_keras_h5(layers, names)
    with pytest.raises(ImportError):
        g.load_weights("/nonexistent/netG.h5")                               # h5py is not installed in this image


def test_split_and_shuffle_host_logic():
    """dep_gan_im_amd.data.split_and_shuffle is pure indexing: on CPU tensors it must reproduce the oracle's
    restatement of GT:738-760 (and therefore scikit-learn's split) element for element."""
    import torch
    rng = np.random.default_rng(4)
    x = rng.normal(size=(123, 4, 4, 2)).astype(np.float32)
    y = rng.normal(size=(123, 4, 4, 1)).astype(np.float32)
    want = do.split_and_shuffle(x, y, seed_shuffle=11)
    got = dgdata.split_and_shuffle(torch.from_numpy(x), torch.from_numpy(y), rng=np.random.RandomState(11))
    assert [tuple(g.shape) for g in got] == [w.shape for w in want]
    assert all(np.array_equal(g.numpy(), w) for g, w in zip(got, want))
    assert got[1].shape[0] == 3                        # ceil(0.02 * 123) validation slices
