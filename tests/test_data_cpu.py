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
    with pytest.raises(OSError):
        g.load_weights("/nonexistent/netG.h5")                               # opened by h5py, or by h5lite without it


def test_h5lite_reads_a_real_hdf5_file_in_keras_layout():
    """dep_gan_im_amd/h5lite.py (pure Python) against a file the REAL HDF5 library wrote: tests/golden/
    keras_layout_small.h5 was produced by h5py 3.3.0 / HDF5 1.10.6 (tests/golden/make_keras_h5.py, run under the image's
    /opt/conda interpreter) from keras_layout_small.npz in the layout of keras.engine.saving.save_weights_to_hdf5_group
    (GT:892 `netG.save`): groups through B-tree + symbol nodes + local heap, nested "<layer>/<layer>/<weight>:0"
    datasets, fixed-length (h5py 2.x style) and variable-length (h5py 3.x style) string attributes, weight-less layers."""
    from dep_gan_im_amd import h5lite
    from dep_gan_im_amd.models import weights_from_keras_h5
    gold = os.path.join(os.path.dirname(__file__), "golden")
    with np.load(os.path.join(gold, "keras_layout_small.npz")) as z:
        want = {k: z[k] for k in z.files}
    with h5lite.File(os.path.join(gold, "keras_layout_small.h5")) as f:
        assert f.keys() == ["model_weights"] and f.attrs["backend"] == b"tensorflow" and f.attrs["keras_version"] == b"2.2.4"
        g = f["model_weights"]
        layers = [n.decode() for n in g.attrs["layer_names"]]
        assert sorted(layers) == sorted(g.keys()) and "activation_2" in layers
        assert list(g["activation_2"].keys()) == [] and len(g["activation_2"].attrs["weight_names"]) == 0
        assert [n.decode() for n in g["conv2d_gen_0"].attrs["weight_names"]] == ["conv2d_gen_0/kernel:0",
                                                                                 "conv2d_gen_0/bias:0"]
        d = g["conv2d_gen_0"]["conv2d_gen_0"]["kernel:0"]
        assert d.shape == (3, 3, 1, 32) and d.dtype == np.float32
        assert "conv2d_gen_0/conv2d_gen_0/kernel:0" in g and "conv2d_gen_0/nope" not in g
        got = weights_from_keras_h5(f, list(want))                       # the importer itself, on the real file
        assert list(got) == list(want)
        for k in want:
            assert got[k].dtype == np.float32 and np.array_equal(got[k], want[k]), k
        with pytest.raises(KeyError):
            weights_from_keras_h5(f, list(want) + ["conv2d_gen_1/kernel"])
    with pytest.raises(h5lite.H5Error):
        h5lite.File(os.path.join(gold, "keras_layout_small.npz"))         # not an HDF5 file


CONDA_PY = "/opt/conda/bin/python3.9"


def _conda_has_h5py():
    import subprocess
    if not os.path.exists(CONDA_PY):
        return False
    try:
        return subprocess.run([CONDA_PY, "-c", "import h5py"], capture_output=True, timeout=60).returncode == 0
    except Exception:
        return False


@pytest.mark.parametrize("layout", ["model", "weights"])
def test_load_weights_from_keras_h5_written_by_the_hdf5_library(tmp_path, layout):
    """Model.load_weights("....h5") (GE:383) end to end: every tensor of Gen_UNet2D and Dis_C2D_FCN1 (2.49 M + 1.80 M
    parameters, the reference's sizes) is written to a Keras-layout HDF5 file by h5py in a SECOND interpreter (the image's
    /opt/conda python; the main one has no h5py) and loaded back through dep_gan_im_amd.h5lite -- both `model.save` and
    `save_weights` layouts.  Skipped where that interpreter does not exist; the committed small file above always runs."""
    import subprocess
    from dep_gan_im_amd import Dis_C2D_FCN1, Gen_UNet2D
    if not _conda_has_h5py():
        pytest.skip("no interpreter with h5py in this environment (%s)" % CONDA_PY)
    script = os.path.join(os.path.dirname(__file__), "golden", "make_keras_h5.py")
    for make in (lambda seed: Gen_UNet2D((64, 64, 1), (32, 1), 32, 1, seed=seed), lambda seed: Dis_C2D_FCN1((256, 256, 1), seed=seed)):
        src = make(5)
        want = src.get_weights_dict()
        npz, h5 = str(tmp_path / "w.npz"), str(tmp_path / ("w_%s.h5" % layout))
        np.savez(npz, **want)
        r = subprocess.run([CONDA_PY, script, npz, h5, layout], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        dst = make(6)
        assert not np.array_equal(dst.get_weights_dict()[next(iter(want))], want[next(iter(want))])
        dst.load_weights(h5)
        got = dst.get_weights_dict()
        assert list(got) == list(want)
        for k in want:
            assert np.array_equal(got[k], want[k]), k


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
