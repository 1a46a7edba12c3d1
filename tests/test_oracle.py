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


# ---------------------------------------------------------------------------
# Mask-pinned evaluation (oracle/manual.py `masks`): what the GPU gradient tests compare against
# ---------------------------------------------------------------------------
def _own_critic_masks(PD, img_nhwc):
    """critic_masks() from the activations of an fp32 forward of the oracle itself (stands in for the HIP tensors)."""
    taps = {}
    with torch.no_grad():
        O.d_forward_t(O.to_torch(PD), torch.tensor(img_nhwc), taps=taps)
    return M.critic_masks({k: v.permute(0, 2, 3, 1).numpy() for k, v in taps.items()})


def _flip_some(masks, rng, frac=0.02):
    """Flips a few ReLU signs and moves a few arg-maxes inside their window: masks NO evaluation would produce."""
    out = {}
    for name, (m, idx) in masks.items():
        m = m.clone()
        flip = torch.from_numpy(rng.uniform(size=tuple(m.shape)) < frac)
        m[flip] = ~m[flip]
        if idx is not None:
            idx = idx.clone()
            W = m.shape[3]
            move = torch.from_numpy(rng.uniform(size=tuple(idx.shape)) < frac)
            # toggle the column inside the 2x2 window: even x -> x + 1, odd x -> x - 1
            idx[move] = idx[move] + 1 - 2 * (idx[move] % W % 2)
        out[name] = (m, idx)
    return out


def test_first_argmax_rule_is_torchs_and_handles_ties():
    a = np.zeros((1, 2, 4, 4), np.float32)
    a[0, 0, 0, 1] = a[0, 0, 1, 0] = 2.0                 # tie inside window (0,0): the first in row-major order wins
    a[0, 1, 3, 3] = 1.0
    idx = M.first_argmax_idx(a)
    assert idx[0, 0, 0, 0] == 1 and idx[0, 0, 1, 1] == 2 * 4 + 2 and idx[0, 1, 1, 1] == 3 * 4 + 3
    r = np.random.default_rng(0).integers(0, 3, size=(2, 3, 8, 8)).astype(np.float32)      # many ties
    _, tidx = torch.nn.functional.max_pool2d(torch.from_numpy(r), 2, return_indices=True)
    np.testing.assert_array_equal(M.first_argmax_idx(r), tidx.numpy())


def test_masked_manual_critic_own_masks_reproduce_and_flipped_masks_match_autograd():
    img, B, delta = 32, 2, 10.0
    PD = O.init_critic(2, bias_std=0.05, img=img)
    x, y2, z, ep = O.synth_batch(5, B, img, img)
    rng = np.random.default_rng(1)
    real = (y2 + 0.02 * rng.uniform(size=y2.shape)).astype(np.float32)
    fake = (x + 0.02 * rng.uniform(size=x.shape)).astype(np.float32)
    mixed = (ep * real + (1 - ep) * fake).astype(np.float32)
    own = tuple(_own_critic_masks(PD, a) for a in (real, fake, mixed))
    outs0, g0, aux0 = M.critic_grads_manual(PD, real, fake, ep, delta)
    outs1, g1, aux1 = M.critic_grads_manual(PD, real, fake, ep, delta, masks=own)
    # tie-free inputs: the fp32 forward's masks ARE the fp64 evaluation's masks -> identical results
    np.testing.assert_allclose(outs1, outs0, rtol=1e-12)
    for n in g0:
        np.testing.assert_allclose(g1[n], g0[n], rtol=1e-9, atol=1e-14)
    # masks no forward pass would produce: the hand-derived backward must still be the derivative of the masked forward
    masks = tuple(_flip_some(m, rng) for m in own)
    outs2, g2, aux2 = M.critic_grads_manual(PD, real, fake, ep, delta, masks=masks)
    T = O.to_torch(PD, torch.float64, requires_grad=True)
    r, f = (torch.tensor(a, dtype=torch.float64).permute(0, 3, 1, 2) for a in (real, fake))
    e = torch.tensor(ep, dtype=torch.float64).reshape(-1, 1, 1, 1)
    mx = (e * r + (1 - e) * f).requires_grad_(True)
    out_r, _ = M.d_forward_store(T, r, masks[0])
    out_f, _ = M.d_forward_store(T, f, masks[1])
    out_m, _ = M.d_forward_store(T, mx, masks[2])
    (gm,) = torch.autograd.grad(out_m.sum(), mx, create_graph=True)
    gp = ((torch.sqrt((gm ** 2).sum((1, 2, 3))) - 1.0) ** 2).mean()
    loss = out_f.mean() - out_r.mean() + delta * gp
    names = O.trainable_names(PD)
    gs = torch.autograd.grad(loss, [T[n] for n in names], allow_unused=True)
    assert abs(float(gp.detach()) - aux2["gp"]) < 1e-12 and abs(aux2["gp"] - aux0["gp"]) > 1e-9      # the flips are visible
    for n, g in zip(names, gs):
        want = np.zeros_like(g2[n]) if g is None else g.detach().numpy()
        np.testing.assert_allclose(g2[n], want, rtol=1e-7, atol=1e-12 + 1e-9 * np.abs(want).max(), err_msg=n)


def test_masked_manual_generator_matches_autograd_of_the_masked_forward():
    img, B = 32, 2
    # (seed chosen so that the fp32 and the fp64 forward take the same side of every kink: with most seeds one of the
    # ~3e5 units does not -- 1 / 2 / 3 / 4 / 6 give 2 / 0 / 1 / 1 / 1 differing decisions -- which is the very reason
    # the GPU gradient tests pin the masks)
    PG = O.init_generator(2, bias_std=0.05)
    PD1 = O.init_critic(12, bias_std=0.05, img=img)
    PD2 = O.init_critic(13, bias_std=0.05, img=img)
    x, y2, z, ep = O.synth_batch(5, B, img, img)
    rng = np.random.default_rng(3)
    x = (x + 0.02 * rng.uniform(size=x.shape)).astype(np.float32)
    y2 = (y2 + 0.02 * rng.uniform(size=y2.shape)).astype(np.float32)
    # the tensors depgan_debug_tensor hands out, from an fp32 oracle forward
    taps = {}
    T32 = O.to_torch(PG)
    with torch.no_grad():
        attr = O.g_forward_t(T32, torch.tensor(x), torch.tensor(z), taps=taps).numpy()
        heads = O.noise_mlp(T32, torch.tensor(z))
        st32 = M.g_forward_store(T32, torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(z))[1]
    outs = {k: v.permute(0, 2, 3, 1).numpy() for k, v in taps.items() if k.startswith(("gen_", "de_gen"))}
    for k in ("de_gen_9", "de_gen_11", "de_gen_15"):        # taps hold the concatenation; the deconv part comes first
        outs[k] = outs[k][..., :st32[k].shape[1]]
    us = {k[:-2]: v.permute(0, 2, 3, 1).numpy() for k, v in st32.items() if k.endswith("/u")}
    hcat = np.concatenate([heads["noise_2_" + sfx].numpy() for sfx, _ in O.NOISE_HEADS], axis=1)
    mg = M.generator_masks(outs, us, hcat, st32["noise"]["a0"].numpy(), st32["noise"]["a1"].numpy(), attr, x, y2)
    md1 = _own_critic_masks(PD1, (x[..., 0:1] + attr).astype(np.float32))
    md2 = _own_critic_masks(PD2, attr)
    o0, g0 = M.g_grads_manual(PG, PD1, PD2, x, y2, z)
    o1, g1 = M.g_grads_manual(PG, PD1, PD2, x, y2, z, masks=(mg, md1, md2))
    np.testing.assert_allclose(o1, o0, rtol=1e-9)
    worst = max(float(np.abs(g1[n] - g0[n]).max() / (np.abs(g0[n]).max() + 1e-30)) for n in g0)
    assert worst < 1e-9, worst       # own masks: the fp32 forward's decisions are the fp64 evaluation's here
    # perturbed masks against autograd through the masked forward
    mg2 = dict(mg)
    for k, v in mg.items():
        if v.dtype == torch.bool:
            v = v.clone()
            flip = torch.from_numpy(rng.uniform(size=tuple(v.shape)) < 0.02)
            v[flip] = ~v[flip]
            mg2[k] = v
    md1b, md2b = _flip_some(md1, rng), _flip_some(md2, rng)
    o2, g2 = M.g_grads_manual(PG, PD1, PD2, x, y2, z, masks=(mg2, md1b, md2b))
    TG = O.to_torch(PG, torch.float64, requires_grad=True)
    TD1, TD2 = O.to_torch(PD1, torch.float64), O.to_torch(PD2, torch.float64)
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2)
    y2t = torch.tensor(y2, dtype=torch.float64).permute(0, 3, 1, 2)
    at, _ = M.g_forward_store(TG, xt, torch.tensor(z, dtype=torch.float64), masks=mg2)
    d1, _ = M.d_forward_store(TD1, xt[:, 0:1] + at, md1b)
    d2, _ = M.d_forward_store(TD2, at, md2b)
    diff = at - (y2t - xt[:, 0:1])
    loss = -d1.mean() - d2.mean() + 100.0 * (mg2["sign"].to(torch.float64) * diff).mean()
    names = O.trainable_names(PG)
    gs = torch.autograd.grad(loss, [TG[n] for n in names], allow_unused=True)
    for n, g in zip(names, gs):
        want = np.zeros_like(g2[n]) if g is None else g.detach().numpy()
        np.testing.assert_allclose(g2[n], want, rtol=1e-7, atol=1e-12 + 1e-9 * np.abs(want).max(), err_msg=n)


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
