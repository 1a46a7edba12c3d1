"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol declared
in include/depgan.h, and the host-side facade mirrors the reference surface."""
import os
import re

import numpy as np
import pytest

import dep_gan_im_amd as dg
from dep_gan_im_amd import _lib, models
from oracle import depgan_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "depgan.h")).read()
    declared = set(re.findall(r"\b(depgan_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"depgan_ctx", "depgan_config"}
    assert len(declared) >= 25
    for name in sorted(declared):
        assert hasattr(lib, name), "libdepgan.so does not export %s" % name
    assert set(_lib.EXPORTS) <= declared


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DepganError):
        _lib.load()


def test_no_gpu_means_no_silent_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = dg.Gen_UNet2D((256, 256, 1))
    with pytest.raises(_lib.DepganError):
        g.predict([np.zeros((1, 256, 256, 1), np.float32), np.zeros((1, 32, 1), np.float32)])


def test_weight_tables_follow_reference_layer_names():
    g = dg.Gen_UNet2D((256, 256, 1), (32, 1), 32, 1, seed=0)
    ref = O.init_generator(0)
    assert [(n, tuple(s)) for n, s, _ in g._static_table()] == [(k, v.shape) for k, v in ref.items()]
    assert [w.name for w in g.trainable_weights] == O.trainable_names(ref)
    d = dg.Dis_C2D_FCN1((256, 256, 1), seed=0)
    refd = O.init_critic(0)
    assert [(n, tuple(s)) for n, s, _ in d._static_table()] == [(k, v.shape) for k, v in refd.items()]
    assert g.count_params() == 2486145 + 2 * 2912 and d.count_params() == 1798002
    g2 = dg.Gen_UNet2D((256, 256, 2))
    assert dict((n, s) for n, s, _ in g2._static_table())["conv2d_gen_0/kernel"] == (3, 3, 2, 32)


def test_keras_default_initialisers():
    g = dg.Gen_UNet2D((256, 256, 1), seed=3)
    w = g.get_weights_dict()
    assert np.all(w["bn_gen_0/gamma"] == 1) and np.all(w["bn_gen_0/moving_variance"] == 1)
    assert np.all(w["conv2d_gen_0/bias"] == 0) and np.all(w["bn_gen_0/moving_mean"] == 0)
    k = w["conv2d_gen_1/kernel"]
    lim = np.sqrt(6.0 / (9 * 32 + 9 * 32))
    assert k.shape == (3, 3, 32, 32) and np.abs(k).max() <= lim and np.abs(k).max() > 0.9 * lim
    hk = w["dense_noise_2_mul/kernel"]
    # truncated normal whose RESULT has std sqrt(2/fan_in) (Keras >= 2.2.3; models.HE_NORMAL_TRUNC_STD), never beyond
    # two sampled standard deviations -- and the same definition as the oracle's initialiser
    assert hk.shape == (1024, 128) and abs(hk.std() / np.sqrt(2.0 / 1024) - 1.0) < 0.03
    assert np.abs(hk).max() <= 2.0 * np.sqrt(2.0 / 1024) / models.HE_NORMAL_TRUNC_STD + 1e-7
    ok = O.init_generator(5)["dense_noise_2_mul/kernel"]
    assert abs(ok.std() / hk.std() - 1.0) < 0.03


def test_host_save_load_and_errors(tmp_path):
    d = dg.Dis_C2D_FCN1((256, 256, 1), seed=1)
    p = str(tmp_path / "d.npz")
    d.save(p)
    d2 = dg.Dis_C2D_FCN1((256, 256, 1), seed=2)
    d2.load_weights(p)
    for a, b in zip(d.get_weights(), d2.get_weights()):
        np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError):
        d2.set_weights([np.zeros(3)])
    with pytest.raises(ValueError):
        d2.set_weights({"conv2d_dis_0a/kernel": np.zeros((3, 3, 1, 16), np.float32)})
    with pytest.raises(ValueError):
        dg.Dis_C2D_FCN1((256, 256, 2))
    with pytest.raises(ValueError):
        dg.Gen_UNet2D((256, 256, 1), (16, 1))
    lines = []
    d.summary(print_fn=lines.append)
    assert any("1798002" in ln for ln in lines)


def _header_config_fields():
    hdr = open(os.path.join(ROOT, "include", "depgan.h")).read()
    body = re.search(r"typedef struct depgan_config \{(.*?)\} depgan_config;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.split(None, 1)
        fields += [(n.strip(), ctype) for n in names.split(",")]
    return fields


def test_config_struct_matches_header_binding_and_integration_doc(lib):
    import ctypes as C
    fields = _header_config_fields()
    assert fields[0] == ("struct_size", "int")
    ctypes_of = {"int": C.c_int, "float": C.c_float}
    assert [(n, ctypes_of[t]) for n, t in fields] == list(_lib.Config._fields_)
    assert lib.depgan_config_size() == C.sizeof(_lib.Config) == 4 * len(fields)
    hdr = open(os.path.join(ROOT, "include", "depgan.h")).read()
    assert int(re.search(r"#define DEPGAN_ABI_VERSION (\d+)", hdr).group(1)) == lib.depgan_abi_version() \
        == _lib.ABI_VERSION
    # the stub a maintainer would copy out of INTEGRATION.md declares the same struct
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = doc[doc.index("class Config(C.Structure)"):]
    stub = stub[:stub.index("ctx = C.c_void_p()")]
    doc_fields = re.findall(r'\("([a-z0-9_A-Z]+)", C\.c_(int|float)\)', stub)
    assert doc_fields == fields
    assert "depgan_abi_version() == %d" % _lib.ABI_VERSION in stub
    kw = re.search(r"cfg = Config\((.*?)\)\n", stub, re.S).group(1)
    assert [k.strip().split("=")[0] for k in kw.split(",")] == [n for n, _ in fields]


def test_create_rejects_a_config_of_another_size(lib):
    """No GPU needed: the size check comes before any HIP call."""
    import ctypes as C
    cfg = _lib.Config(batch=2, height=64, width=64, nicg=1, first_fm=32, im_thresh=0.5, delta=10.0, lrD=1e-4, lrG=1e-4,
                      beta1=0.0, beta2=0.9, adam_eps=1e-7, nc_out=1)
    assert cfg.struct_size == C.sizeof(_lib.Config)
    cfg.struct_size -= 4                     # what a caller bound to the previous header would pass
    h = C.c_void_p()
    assert lib.depgan_create(C.byref(cfg), C.byref(h)) != 0 and not h.value
    assert b"struct_size" in lib.depgan_last_error()
    cfg.struct_size = 0                      # a caller that never heard of the field
    assert lib.depgan_create(C.byref(cfg), C.byref(h)) != 0


def test_library_carries_the_hash_of_the_sources_it_was_built_from(lib, tmp_path, monkeypatch):
    """The GPU box loads the library that was cross-compiled here: the binding refuses one whose embedded source hash
    is not the hash of the sources lying next to it."""
    from dep_gan_im_amd import build
    assert lib.depgan_source_hash().decode() == build.source_hash()
    real = build.source_hash
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 32)
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(_lib.DepganError, match="other sources"):
        _lib.load()
    monkeypatch.setattr(build, "source_hash", real)
    monkeypatch.setattr(_lib, "_lib", None)
    assert _lib.load() is not None
