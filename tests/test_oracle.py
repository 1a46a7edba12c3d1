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
