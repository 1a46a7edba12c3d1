"""GPU parity of the DEP-UResNet supervised path through the C ABI and the Keras-style
facade: phase-0 predict / evaluate, one learning-phase-1 gradient evaluation, and train_on_batch / fit steps
against the CPU oracle and the committed golden vectors.

Tolerances.  Forward quantities (probabilities, losses, batch statistics) meet north_star's 1e-3 with two
orders of margin.  The phase-1 *gradient* flows through 40 batch-statistics BatchNorms, each of which removes
the mean of the incoming gradient: the result is a small residual of large terms, and the CPU oracle's own
fp32 and fp64 evaluations differ by up to 1e-2 on single tensors (measured below, same inputs).  So gradients
are checked as (a) relative L2 error over the whole gradient < 1e-3, and (b) per tensor against the fp64 oracle
with the oracle's own fp32-vs-fp64 spread as the yardstick."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _mg():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def _l2(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _engine(img, B, P):
    from dep_gan_im_amd import Engine
    eng = Engine(B, img, img, 1, lrG=1e-4, beta1=0.9, beta2=0.999, nc_out=4)
    eng.set_weights("G", P)
    return eng


def test_predict_eval_and_grads_match_golden_and_oracle(lib):
    from oracle import depgan_oracle as O
    g = np.load(os.path.join(GOLD, "uresnet_64_b4.npz"))
    img, B, seed, ds = int(g["img"]), int(g["B"]), int(g["seed"]), int(g["drop_seed"])
    P = _mg().uresnet_params(seed)
    x, z, lab = O.synth_uresnet_batch(seed + 3, B, img, img)
    eng = _engine(img, B, P)
    probs = eng.g_forward(x, z).cpu().numpy()
    assert probs.shape == (B, img, img, 4)
    np.testing.assert_allclose(probs.sum(-1), 1.0, atol=1e-5)
    np.testing.assert_allclose(probs.reshape(-1)[g["probs_idx"]], g["probs_samples"], rtol=1e-3, atol=1e-6)
    assert abs(eng.uresnet(x, z, lab, "eval") - float(g["eval_loss"])) < 1e-4
    # phase 1, gradients only
    loss = eng.uresnet(x, z, lab, "grads", drop_seed=ds)
    assert abs(loss - float(g["loss"])) < 1e-4 * max(1.0, abs(float(g["loss"])))
    G = eng.get_grads("G")
    loss64, g64, stats = O.uresnet_grads(P, x, z, lab, drop_seed=ds, dtype=torch.float64)
    loss32, g32, _ = O.uresnet_grads(P, x, z, lab, drop_seed=ds)
    live = [k for k in g64 if float(np.abs(g64[k]).max()) > 1e-9]
    cat = lambda d: np.concatenate([np.asarray(d[k], np.float64).reshape(-1) for k in live])  # noqa: E731
    assert _l2(cat(G), cat(g64)) < 1e-3
    spread = max(_l2(g32[k], g64[k]) for k in live)
    worst = max(_l2(G[k], g64[k]) for k in live)
    assert worst <= 3.0 * spread + 1e-3, (worst, spread)
    # per-tensor gradient norms against the committed golden values: a FREE comparison (each side takes its own ReLU /
    # arg-max decisions), so a unit within rounding of its kink moves a small tensor by per cent on any change of
    # summation order (the one-pass batch moments of round 3 did: one tensor of 122 by 2.5 %) -- the bound every
    # evaluation must meet is test_phase1_gradients_under_hip_masks below; here: 2 % on all tensors but at most two
    gn = np.array([float(np.linalg.norm(np.float64(G[k]))) for k in g64])
    big = g["gnorm"] > 1e-6
    relerr = np.abs(gn[big] - g["gnorm"][big]) / g["gnorm"][big]
    assert (relerr > 2e-2).sum() <= 2 and relerr.max() < 0.1, np.sort(relerr)[-4:]
    # biases feeding a batch-statistics BN: the exact gradient is zero
    scale = float(np.abs(cat(g64)).max())
    for k in g64:
        if k not in live:
            assert float(np.abs(G[k]).max()) < 1e-5 * scale, k
    # the phase-1 forward pass moved the moving statistics (momentum 0.99, keras variance corrections)
    W = eng.get_weights("G")
    for name, (mean, var, n, fused) in stats.items():
        corr = n / (n - 1.0) if fused else n / (n - (1.0 + O.BN_EPS))
        np.testing.assert_allclose(W[name + "/moving_mean"], P[name + "/moving_mean"] * 0.99 + mean.numpy() * 0.01,
                                   rtol=1e-4, atol=1e-6, err_msg=name)
        np.testing.assert_allclose(W[name + "/moving_variance"],
                                   P[name + "/moving_variance"] * 0.99 + var.numpy() * corr * 0.01,
                                   rtol=1e-4, atol=1e-6, err_msg=name)
    # ... and phase-0 predict now uses them
    P2 = dict(P)
    for k in W:
        if "moving_" in k:
            P2[k] = W[k]
    np.testing.assert_allclose(eng.g_forward(x, z).cpu().numpy(), O.uresnet_predict(P2, x, z), atol=2e-5)
    eng.close()


@pytest.mark.parametrize("ds", [0, 77], ids=["no_dropout", "dropout"])
def test_phase1_gradients_under_hip_masks(lib, ds):
    """The learning-phase-1 gradient (UT:423-427, 602-606: batch-statistics BatchNorm differentiated through, Dropout,
    softmax + cross-entropy) against the fp64 oracle evaluated under the ReLU signs / pool arg-maxes / FiLM signs the
    HIP pass took (tests/test_gpu_masked.py): per tensor, every evaluation.  Biases in front of a batch-statistics BN
    have an exactly zero gradient; they must be zero to the rounding of the largest gradient entry."""
    import test_gpu_masked as TM
    from oracle import depgan_oracle as O
    img, B = 64, 4
    P = _mg().uresnet_params(5)
    x, z, lab = O.synth_uresnet_batch(6, B, img, img)
    eng = _engine(img, B, P)
    loss = eng.uresnet(x, z, lab, "grads", drop_seed=ds)
    G = eng.get_grads("G")
    masks = TM.hip_uresnet_masks(eng, B)
    loss64, g64, _ = O.uresnet_grads(P, x, z, lab, drop_seed=ds or None, dtype=torch.float64, masks=masks)
    _, g32, _ = O.uresnet_grads(P, x, z, lab, drop_seed=ds or None, dtype=torch.float32, masks=masks)
    assert abs(loss - loss64) < 1e-5 * max(1.0, abs(loss64))
    errs, errs32 = TM.tensor_errors(G, g64), TM.tensor_errors(g32, g64)
    worst = max(errs, key=errs.get)
    print("DEP-UResNet phase-1 gradient under HIP's masks (%s): worst tensor %s %.2e (the oracle's own fp32 run under the "
          "same masks: %.2e on it, %.2e at its worst)" % ("dropout" if ds else "no dropout", worst, errs[worst],
                                                          errs32[worst], max(errs32.values())))
    # With the decisions pinned what is left is conditioning: every batch-statistics BN subtracts the mean of the incoming
    # gradient (the 14 noise-head BNs over a batch of FOUR rows), so some tensors are small residuals of large terms in
    # any fp32 arithmetic (measured: HIP 1.2e-4 / 3.7e-4 at its worst tensor, the CPU's fp32 evaluation of the same masked
    # graph 1.2e-4 / 1.3e-4 at ITS worst; the tensors differ -- both are noise-MLP weights behind three such BNs).  Per
    # tensor: 1e-4, or within the CPU fp32 evaluation's own worst error x 4, and only a handful of tensors above 1e-4.
    cap = max(1e-4, 4.0 * max(errs32.values()))
    for k in errs:
        assert errs[k] < cap, (k, errs[k], errs32[k])
    assert sum(e > 1e-4 for e in errs.values()) <= 8, sorted(errs.items(), key=lambda kv: -kv[1])[:10]
    eng.close()


def test_dropout_mask_is_the_oracles(lib):
    """drop_seed selects the same keep mask as oracle.dropout_keep_mask: with dropout on, the loss moves exactly
    as the oracle's does, and seed 0 means no dropout."""
    from oracle import depgan_oracle as O
    P = _mg().uresnet_params(5)
    x, z, lab = O.synth_uresnet_batch(6, 4, 64, 64)
    eng = _engine(64, 4, P)
    for ds in (0, 1, 77):
        eng.set_weights("G", P)
        got = eng.uresnet(x, z, lab, "grads", drop_seed=ds)
        want, _, _ = O.uresnet_grads(P, x, z, lab, drop_seed=ds or None)
        assert abs(got - want) < 2e-5 * max(1.0, abs(want)), (ds, got, want)
    eng.close()


def test_train_on_batch_steps_follow_the_oracle(lib):
    from oracle import depgan_oracle as O
    g = np.load(os.path.join(GOLD, "uresnet_64_b4.npz"))
    img, B, seed, ds = int(g["img"]), int(g["B"]), int(g["seed"]), int(g["drop_seed"])
    P = _mg().uresnet_params(seed)
    x, z, lab = O.synth_uresnet_batch(seed + 3, B, img, img)
    eng = _engine(img, B, P)
    tr = O.OracleUResNet(P)
    for k in range(2):
        got = eng.uresnet(x, z, lab, "step", drop_seed=ds + k)
        want = tr.train_on_batch([x, z], lab, drop_seed=ds + k)
        assert abs(got - float(g["step_losses"][k])) < 2e-3 * abs(float(g["step_losses"][k]))
        assert abs(got - want) < 2e-3 * abs(want)
    W = eng.get_weights("G")
    lr = 1e-4
    for k in P:
        if "moving_" in k:
            np.testing.assert_allclose(W[k], P[k], rtol=2e-3, atol=1e-5, err_msg=k)
        else:   # Adam's first steps are +-lr per element whatever the gradient's size
            assert float(np.abs(W[k] - P[k]).max()) <= 2 * 2.05 * lr, k
    live = [k for k in P if "moving_" not in k and not (k.endswith("/bias") and "segmentation" not in k)]
    moved = np.concatenate([(W[k] - P[k]).reshape(-1) for k in live])
    assert float(np.mean(np.abs(moved) < 0.2 * lr)) > 0.9      # the bulk of the weights agree far below one step
    np.testing.assert_allclose(eng.g_forward(x, z).cpu().numpy(), tr.predict([x, z]), atol=5e-3)
    eng.close()


def test_facade_fit_history_and_short_last_batch(lib):
    """Gen_UNet2D(..., nc_out=4).fit([flair, noise], onehot, epochs, batch_size, shuffle, validation_data)
    (UT:583-606): per-epoch loss / val_loss, a short last batch, save / load round trip."""
    from dep_gan_im_amd import Gen_UNet2D
    from oracle import depgan_oracle as O
    net = Gen_UNet2D((64, 64, 1), (32, 1), 32, 4, seed=3)
    assert net.count_params() == Gen_UNet2D((64, 64, 1), (32, 1), 32, 1).count_params() + 3 * 33
    x, z, lab = O.synth_uresnet_batch(12, 7, 64, 64)
    vx, vz, vlab = O.synth_uresnet_batch(13, 3, 64, 64)
    np.random.seed(0)
    lines = []
    h = net.fit([x, z], lab, epochs=3, batch_size=4, shuffle=True, validation_data=([vx, vz], vlab),
                print_fn=lines.append)
    assert len(h.history["loss"]) == 3 and len(h.history["val_loss"]) == 3 and len(lines) == 3
    assert all(np.isfinite(h.history["loss"])) and all(np.isfinite(h.history["val_loss"]))
    assert h.history["loss"][-1] < h.history["loss"][0]           # Adam at 1e-4 on a fixed set: loss goes down
    assert abs(net.evaluate([vx, vz], vlab, batch_size=4) - h.history["val_loss"][-1]) < 1e-6
    p = net.predict([vx, vz])
    assert p.shape == (3, 64, 64, 4) and abs(float(p.sum(-1).mean()) - 1.0) < 1e-5
    with pytest.raises(RuntimeError):
        Gen_UNet2D((64, 64, 1)).fit([x, z], lab)                    # the tanh generator is not compiled
