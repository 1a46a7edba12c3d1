"""CPU: the oracle against its committed golden vectors, and the two independent
restatements (autograd graph vs hand-derived backward) against each other.
Parity vs the real Keras reference is UNPINNED (oracle/__init__.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import depgan_oracle as O
from oracle import manual as M

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _checks(P):
    return np.array([float(np.sum(np.asarray(v, np.float64))) for v in P.values()])


def _setup(g):
    img, B, seed = int(g["img"]), int(g["B"]), int(g["seed"])
    PG = O.init_generator(seed, bias_std=0.05)
    PD1 = O.init_critic(seed + 1, bias_std=0.05, img=img)
    PD2 = O.init_critic(seed + 2, bias_std=0.05, img=img)
    x, y2, z, ep = O.synth_batch(seed + 5, B, img, img)
    return PG, PD1, PD2, x, y2, z, ep


@pytest.mark.parametrize("name", ["small_64_b2", "full_256_b2"])
def test_oracle_matches_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    PG, PD1, PD2, x, y2, z, ep = _setup(g)
    # seeded constructors have not drifted
    np.testing.assert_allclose(_checks(PG), g["wsumG"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(_checks(PD1), g["wsumD1"], rtol=1e-12, atol=1e-9)
    assert abs(float(x.astype(np.float64).sum()) - float(g["xsum"])) < 1e-6
    attr = O.g_predict(PG, x, z)
    np.testing.assert_allclose(attr.reshape(-1)[g["attr_idx"]], g["attr_samples"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(O.d_predict(PD1, y2).reshape(-1), g["d_y2"], rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(O.g_eval(PG, PD1, PD2, x, y2, z), g["g_eval"], rtol=2e-4, atol=1e-5)
    if name.startswith("small"):
        outs, grads, aux = O.critic_grads(PD1, PG, y2, x, z, ep, "y2")
        np.testing.assert_allclose(outs, g["critic_y2_outs"], rtol=2e-4, atol=1e-5)
        gn = [float(np.sqrt((np.asarray(v, np.float64) ** 2).sum())) for v in grads.values()]
        np.testing.assert_allclose(gn, g["critic_y2_gnorm"], rtol=2e-3, atol=1e-7)


def test_manual_backward_matches_autograd_critic():
    img = 32
    PG = O.init_generator(1, bias_std=0.05)
    PD = O.init_critic(2, bias_std=0.05, img=img)
    x, y2, z, ep = O.synth_batch(5, 2, img, img)
    for which in ("y2", "dem"):
        outs, grads, aux = O.critic_grads(PD, PG, y2, x, z, ep, which, dtype=torch.float64)
        attr = aux["attr"]
        real, fake = (y2, x[..., 0:1] + attr) if which == "y2" else (y2 - x[..., 0:1], attr)
        outs2, grads2, aux2 = M.critic_grads_manual(PD, real, fake, ep)
        np.testing.assert_allclose(outs, outs2, rtol=1e-10)
        assert abs(aux["gp"] - aux2["gp"]) < 1e-10
        for n in grads:
            np.testing.assert_allclose(grads2[n], grads[n], rtol=1e-7, atol=1e-12 + 1e-9 * np.abs(grads[n]).max())


def test_manual_backward_matches_autograd_generator():
    img = 32
    PG = O.init_generator(1, bias_std=0.05)
    PD1 = O.init_critic(2, bias_std=0.05, img=img)
    PD2 = O.init_critic(3, bias_std=0.05, img=img)
    x, y2, z, ep = O.synth_batch(5, 2, img, img)
    outs, grads = O.g_grads(PG, PD1, PD2, x, y2, z, dtype=torch.float64)
    outs2, grads2 = M.g_grads_manual(PG, PD1, PD2, x, y2, z)
    np.testing.assert_allclose(outs, outs2, rtol=1e-10)
    for n in grads:
        np.testing.assert_allclose(grads2[n], grads[n], rtol=1e-7, atol=1e-12 + 1e-9 * np.abs(grads[n]).max())


def test_semantics_phase0_bn_is_affine_and_m3_m4_have_no_gradient():
    """BN inference affine; M3/M4 (GT:581-589) contribute no gradient."""
    img = 32
    PG = O.init_generator(4, bias_std=0.05)
    PD1 = O.init_critic(5, img=img)
    PD2 = O.init_critic(6, img=img)
    x, y2, z, ep = O.synth_batch(9, 2, img, img)
    _, g_lo = O.g_grads(PG, PD1, PD2, x, y2, z, thr=0.1)
    _, g_hi = O.g_grads(PG, PD1, PD2, x, y2, z, thr=0.9)
    for n in g_lo:   # the threshold only enters M3/M4
        np.testing.assert_array_equal(g_lo[n], g_hi[n])
    T = O.to_torch(PG)
    xin = torch.randn(2, 32, 4, 4)
    y = O._bn_infer(xin, T, "bn_gen_0")
    s = T["bn_gen_0/gamma"] / torch.sqrt(T["bn_gen_0/moving_variance"] + 1e-3)
    ref = (xin - T["bn_gen_0/moving_mean"].view(1, -1, 1, 1)) * s.view(1, -1, 1, 1) + T["bn_gen_0/beta"].view(1, -1, 1, 1)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)


def test_keras_adam_formula():
    P = {"w": np.array([1.0, -2.0], np.float32)}
    opt = O.KerasAdam(["w"], 1e-2, 0.0, 0.9)
    g = {"w": np.array([0.5, -0.25], np.float32)}
    opt.apply(P, g)
    v = 0.1 * g["w"] ** 2
    lr_t = 1e-2 * np.sqrt(1 - 0.9)
    np.testing.assert_allclose(P["w"], np.array([1.0, -2.0]) - lr_t * g["w"] / (np.sqrt(v) + 1e-7), rtol=1e-6)
    assert opt.iterations == 1


def test_param_counts_match_survey():
    PG = O.init_generator(0)
    assert sum(PG[n].size for n in O.trainable_names(PG)) == 2486145
    assert sum(v.size for v in O.init_critic(0).values()) == 1798002


# ---------------------------------------------------------------------------
# DEP-UResNet supervised path
# ---------------------------------------------------------------------------
def _uresnet_setup(g):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    img, B, seed = int(g["img"]), int(g["B"]), int(g["seed"])
    P = mg.uresnet_params(seed)
    x, z, lab = O.synth_uresnet_batch(seed + 3, B, img, img)
    return P, x, z, lab


def test_uresnet_oracle_matches_golden():
    g = np.load(os.path.join(GOLD, "uresnet_64_b4.npz"))
    P, x, z, lab = _uresnet_setup(g)
    np.testing.assert_allclose(_checks(P), g["wsum"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(lab.reshape(-1, 4).sum(0), g["labsum"])
    probs = O.uresnet_predict(P, x, z)
    np.testing.assert_allclose(probs.sum(-1), 1.0, atol=1e-5)
    np.testing.assert_allclose(probs.reshape(-1)[g["probs_idx"]], g["probs_samples"], rtol=2e-4, atol=2e-6)
    ds = int(g["drop_seed"])
    assert int(O.dropout_keep_mask(ds, (int(g["B"]), 16, 16, 96)).sum()) == int(g["keep_sum"])
    loss, grads, stats = O.uresnet_grads(P, x, z, lab, drop_seed=ds)          # fp32 against the stored fp64
    assert abs(loss - float(g["loss"])) < 1e-4
    gn = np.array([float(np.sqrt((np.asarray(v, np.float64) ** 2).sum())) for v in grads.values()])
    big = g["gnorm"] > 1e-6
    np.testing.assert_allclose(gn[big], g["gnorm"][big], rtol=2e-2)
    tr = O.OracleUResNet(P)
    losses = [tr.train_on_batch([x, z], lab, drop_seed=ds + k) for k in range(2)]
    np.testing.assert_allclose(losses, g["step_losses"], rtol=2e-3)
    np.testing.assert_allclose(_checks(P), g["post_wsum"], rtol=1e-3, atol=5e-2)


def test_uresnet_phase1_properties():
    """What learning phase 1 changes: biases in front of a batch-statistics BN get a zero gradient, the moving
    statistics move by (1 - 0.99) of the batch statistics with the keras variance corrections, dropout keeps
    ~75 % and rescales by 4/3, and the loss is keras' clipped cross-entropy."""
    P = O.init_generator(3, nc_out=4, randomize_bn=True, bias_std=0.05)
    x, z, lab = O.synth_uresnet_batch(8, 3, 32, 32)
    loss, grads, stats = O.uresnet_grads(P, x, z, lab, drop_seed=None, dtype=torch.float64)
    scale = max(float(np.abs(v).max()) for v in grads.values())
    for k, v in grads.items():
        if k.endswith("/bias") and not k.startswith("gen_segmentation"):
            assert float(np.abs(v).max()) < 1e-10 * scale, k
    assert float(np.abs(grads["gen_segmentation/bias"]).max()) > 1e-6 * scale
    name = "bn_gen_0"
    mean, var, n, fused = stats[name]
    assert fused and n == 3 * 32 * 32
    mean1, var1, n1, fused1 = stats["dense_bn_noise_2_mul"]
    assert not fused1 and n1 == 3
    P0 = {k: v.copy() for k, v in P.items()}
    O.OracleUResNet(P).train_on_batch([x, z], lab)
    np.testing.assert_allclose(P[name + "/moving_mean"], 0.99 * P0[name + "/moving_mean"] + 0.01 * mean.numpy(),
                               rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(P[name + "/moving_variance"],
                               0.99 * P0[name + "/moving_variance"] + 0.01 * var.numpy() * n / (n - 1.0),
                               rtol=1e-4, atol=1e-6)
    keep = O.dropout_keep_mask(99, (4, 16, 16, 96))
    assert abs(keep.mean() - 0.75) < 0.01
    assert not np.array_equal(keep, O.dropout_keep_mask(100, (4, 16, 16, 96)))
    p = torch.tensor([[0.25, 0.25, 0.25, 0.25], [1.0, 0.0, 0.0, 0.0]], dtype=torch.float64)
    t = torch.tensor([[0.0, 1.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]], dtype=torch.float64)
    want = 0.5 * (-np.log(0.25) - np.log(1e-7))
    assert abs(float(O.keras_categorical_crossentropy_t(p, t)) - want) < 1e-9


def test_bf16_weight_rounding_and_master_updates():
    """BASELINE config 4 in the oracle: round-to-nearest-even on the bit pattern equals torch's bfloat16 cast;
    only '/kernel' tensors are rounded; the closures read rounded copies but Adam moves the fp32 masters."""
    a = np.random.default_rng(0).standard_normal(20000).astype(np.float32) * 3
    np.testing.assert_array_equal(O.round_bf16(a), torch.from_numpy(a).to(torch.bfloat16).float().numpy())
    tie = np.array([1.0 + 2.0 ** -8, 1.0 + 3 * 2.0 ** -8], np.float32)          # exact ties: to even
    np.testing.assert_array_equal(O.round_bf16(tie), np.array([1.0, 1.0 + 2.0 ** -6], np.float32))
    PD = O.init_critic(2, bias_std=0.05, img=32)
    Q = O.round_kernels_bf16(PD)
    for k in PD:
        if k.endswith("/kernel"):
            assert not np.array_equal(Q[k], PD[k]) and np.abs(Q[k] - PD[k]).max() <= np.abs(PD[k]).max() * 2.0 ** -8
        else:
            assert Q[k] is PD[k]
    PG = O.init_generator(1, bias_std=0.05)
    x, y2, z, ep = O.synth_batch(5, 2, 32, 32)
    before = {k: v.copy() for k, v in PD.items()}
    tr = O.OracleTrainers(PG, PD, O.init_critic(3, img=32), weights_dtype="bfloat16")
    out_q = tr.netD_y2_train([y2, x, z, ep])
    moved = [k for k in PD if not np.array_equal(PD[k], before[k])]
    # masters took the Adam step; the two tail biases have an exactly zero WGAN gradient (+1/B per fake, -1/B per
    # real sample cancel and the penalty has no bias gradient), so they stay
    assert sorted(set(PD) - set(moved)) == ["dense_1/bias", "dis_9/bias"]
    assert any(np.abs(PD[k] - O.round_bf16(PD[k])).max() > 0 for k in PD if k.endswith("/kernel"))   # and stay fp32
    out_f = O.OracleTrainers(PG, dict(before), O.init_critic(3, img=32)).netD_y2_train([y2, x, z, ep])
    assert out_q != out_f
