"""GPU gradient parity under pinned masks: a bound EVERY evaluation must meet (no best-of-N, no loose caps).

The two-critic WGAN-GP step (GT:540-549, 562-568, 594) is piecewise linear in ~1e6 (64x64) to ~5e8 (256x256, batch
32) ReLU signs, max-pool arg-maxes and L1 signs.  An fp32 evaluation and an fp64 one that disagree on a single one of
them -- a unit within rounding of its kink -- differ by 1e-3..5e-2 on single tensors, so a free
comparison can only be statistical.  Here the fp64 oracle (oracle/manual.py) is evaluated under the decisions the HIP
path actually took, read back through depgan_debug_tensor: with them fixed the step is multilinear in weights and
inputs, and the HIP gradients must match per tensor to 1e-4 on every seed, at 64x64 and 256x256, for both critics
(first-order and penalty terms) and the generator.  The free evaluation is still run: it reports how many decisions
differ between the HIP path and fp64, and that number is bounded."""
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

D_LAYERS = ["dis_0a", "dis_0b", "dis_1a", "dis_1b", "dis_2", "dis_3", "dis_4", "dis_5", "dis_6", "dis_7", "dis_8"]


def setup(img, B, seed, noisy=True, trained_regime=False, nicg=1):
    from oracle import depgan_oracle as O
    PG = O.init_generator(seed, nicg=nicg, bias_std=0.05)
    PD1 = O.init_critic(seed + 1, bias_std=0.05, img=img)
    PD2 = O.init_critic(seed + 2, bias_std=0.05, img=img)
    x, y2, z, ep = O.synth_batch(seed + 5, B, img, img, nicg=nicg)
    if noisy:
        rng = np.random.default_rng(seed)
        x = (x + 0.02 * rng.uniform(size=x.shape)).astype(np.float32)
        y2 = (y2 + 0.02 * rng.uniform(size=y2.shape)).astype(np.float32)
    if trained_regime:      # penalty in the regime WGAN-GP trains in (norm ~ 2), see tests/test_gpu_steps.py::_setup
        for PD, key in ((PD1, "y2"), (PD2, "dem")):
            _, _, aux = O.critic_grads(PD, PG, y2, x, z, ep, key, dtype=torch.float32)
            PD["dense_1/kernel"] = (PD["dense_1/kernel"] * np.float32(2.0 / float(np.mean(aux["norm"])))).astype(np.float32)
    return PG, PD1, PD2, x, y2, z, ep


def engine(img, B, PG, PD1, PD2, **kw):
    from dep_gan_im_amd import Engine
    eng = Engine(B, img, img, PG["conv2d_gen_0/kernel"].shape[2], **kw)
    for n, P in (("G", PG), ("D_y2", PD1), ("D_dem", PD2)):
        eng.set_weights(n, P)
    eng.debug_capture(True)
    return eng


def hip_critic_masks(eng, B):
    """Decisions of the last critic closure: (real, fake, mixed) passes."""
    from oracle import manual as M
    acts = {n: eng.debug_tensor("d/act/" + n) for n in D_LAYERS}
    return (M.critic_masks({n: a[:B] for n, a in acts.items()}), M.critic_masks({n: a[B:2 * B] for n, a in acts.items()}),
            M.critic_masks({n: eng.debug_tensor("d/mixed/" + n) for n in D_LAYERS}))


def hip_generator_masks(eng, x, y2, B, nicg=1):
    """Decisions of the last generator training closure: generator, D_y2(fake_y2), D_dem(attr)."""
    from oracle import depgan_oracle as O
    from oracle import manual as M
    outs, us = {}, {}
    for ent in O.gen_trunk(nicg, 32, 1):
        if ent[0] in ("conv", "deconv"):
            outs[ent[1]] = eng.debug_tensor("g/out/" + ent[1])
        elif ent[0] == "film":
            us[ent[1]] = eng.debug_tensor("g/u/" + ent[1])
    mg = M.generator_masks(outs, us, eng.debug_tensor("g/heads").reshape(B, 1024), eng.debug_tensor("g/noise_a0"),
                           eng.debug_tensor("g/noise_a1"), eng.debug_tensor("g/out/gen_segmentation"), x, y2, nicg=nicg)
    acts = {n: eng.debug_tensor("d/act/" + n) for n in D_LAYERS}
    return (mg, M.critic_masks({n: a[:B] for n, a in acts.items()}), M.critic_masks({n: a[B:2 * B] for n, a in acts.items()}))


def hip_uresnet_masks(eng, B):
    """Decisions of the last DEP-UResNet training pass (learning phase 1: the same trunk, no critics, no L1 sign)."""
    from oracle import depgan_oracle as O
    from oracle import manual as M
    outs, us = {}, {}
    for ent in O.gen_trunk(1, 32, 4):
        if ent[0] in ("conv", "deconv"):
            outs[ent[1]] = eng.debug_tensor("g/out/" + ent[1])[:B]
        elif ent[0] == "film":
            us[ent[1]] = eng.debug_tensor("g/u/" + ent[1])[:B]
    return M.generator_masks(outs, us, eng.debug_tensor("g/heads")[:B].reshape(B, 1024), eng.debug_tensor("g/noise_a0")[:B],
                             eng.debug_tensor("g/noise_a1")[:B])


def tensor_errors(got, want):
    """Per tensor: max |got - want| / max |want|; tensors whose exact gradient is identically zero (the critics' two
    tail biases: +1/B per fake and -1/B per real sample cancel, the penalty has no bias gradient) must be zero to the
    rounding of the LARGEST gradient entry of the network."""
    gmax = max(float(np.abs(v).max()) for v in want.values())
    errs = {}
    for k in want:
        w = np.asarray(want[k], np.float64)
        scale = float(np.abs(w).max())
        errs[k] = float(np.abs(np.asarray(got[k], np.float64) - w).max()) / (scale if scale > 1e-7 * gmax else gmax)
    return errs


def critic_real_fake(which, x, y2, attr):
    y1 = x[..., 0:1]
    return (y2, y1 + attr) if which == "D_y2" else (y2 - y1, attr)


def check_critic(eng, which, PD, PG, x, y2, z, ep, B, tol=1e-4, nicg=1, dtype=torch.float64, attr=None):
    """One critic closure on the HIP path against the oracle under the HIP path's own decisions."""
    from oracle import depgan_oracle as O
    from oracle import manual as M
    t0 = time.time()
    out = eng.critic(which, y2, x, z, ep, update=False)
    gg = eng.get_grads(which)
    masks = hip_critic_masks(eng, B)
    t1 = time.time()
    if attr is None:
        attr = O.g_predict(PG, x, z, nicg=nicg, dtype=dtype)                   # the oracle's own generator forward
    real, fake = critic_real_fake(which, x.astype(attr.dtype), y2.astype(attr.dtype), attr)
    outs, grads, aux = M.critic_grads_manual(PD, real, fake, ep, dtype=dtype, masks=masks)
    errs = tensor_errors(gg, grads)
    worst = max(errs, key=errs.get)
    gp_hip = eng.last_sums()[2] / eng.last_sums()[3]
    print("%s: worst tensor %s %.2e, outputs %.1e, penalty %.6f vs %.6f  (HIP + read-back %.1f s, oracle %.1f s)" % (
        which, worst, errs[worst], max(abs(a - b) / (abs(b) + 1e-3) for a, b in zip(out, outs)), gp_hip, aux["gp"],
        t1 - t0, time.time() - t1))
    assert max(abs(a - b) / (abs(b) + 1e-3) for a, b in zip(out, outs)) < 1e-4, (which, out, outs)
    assert abs(gp_hip - aux["gp"]) < 1e-4 * (abs(aux["gp"]) + 1e-3), (which, gp_hip, aux["gp"])
    assert errs[worst] < tol, (which, worst, errs[worst])
    return masks, grads


def check_generator(eng, PG, PD1, PD2, x, y2, z, B, tol=1e-4, nicg=1, dtype=torch.float64, thr=0.5):
    from oracle import manual as M
    t0 = time.time()
    out = eng.generator(x, y2, z, "grads")
    gg = eng.get_grads("G")
    masks = hip_generator_masks(eng, x, y2, B, nicg)
    t1 = time.time()
    outs, grads = M.g_grads_manual(PG, PD1, PD2, x, y2, z, thr=thr, nicg=nicg, dtype=dtype, masks=masks)
    errs = tensor_errors(gg, grads)
    worst = max(errs, key=errs.get)
    print("G: worst tensor %s %.2e; scalars %s vs %s  (HIP + read-back %.1f s, oracle %.1f s)"
          % (worst, errs[worst], [round(v, 6) for v in out], [round(v, 6) for v in outs], t1 - t0, time.time() - t1))
    # M3 / M4 are functions of voxel COUNTS at the threshold (GT:581-589): compared where they are smooth
    assert max(abs(a - b) / (abs(b) + 1e-3) for a, b in zip(out[1:4], outs[1:4])) < 1e-4, (out, outs)
    assert errs[worst] < tol, (worst, errs[worst])
    return masks, grads


CASES_64 = [(31, 0), (33, 0), (131, 0), (137, 0), (151, 0), (31, 6), (131, 6)]


@pytest.mark.parametrize("seed,split", CASES_64, ids=["%d-%s" % (s, "split6" if m else "native") for s, m in CASES_64])
def test_gradients_64_under_hip_masks_every_seed(lib, seed, split):
    """64x64, batch 2: both critics and the generator on every seed -- seeds the free comparison used to pick the best
    of (31, 33: random-init critics; 131, 137, 151: penalty in the trained regime; 37 and 149 were measured too and
    dropped for the suite's run time: DESIGN.md section 2 has all nine).  Per tensor 1e-4.
    split = 6: the same bound for the opt-in f32_split mode (fp32 operands split exactly into bf16 terms, six products on
    the bf16 matrix pipe), which the bench line reports next to the headline."""
    from oracle import manual as M
    img, B = 64, 2
    PG, PD1, PD2, x, y2, z, ep = setup(img, B, seed, trained_regime=seed > 100)
    eng = engine(img, B, PG, PD1, PD2, f32_split=split)
    assert eng.f32_split == split
    flips = {}
    report = split == 0 and seed in (31, 131, 151)      # the free evaluations double the test's time: three seeds
    for which, PD in (("D_y2", PD1), ("D_dem", PD2)):
        masks, _ = check_critic(eng, which, PD, PG, x, y2, z, ep, B)
        if not report:
            continue
        # the free fp64 evaluation, for the report: how many decisions did fp32 arithmetic take differently?
        real, fake = critic_real_fake(which, x, y2, eng.debug_tensor("g/out/gen_segmentation"))
        _, _, aux = M.critic_grads_manual(PD, real, fake, ep)
        f = [M.count_decision_flips(a, b) for a, b in zip(masks, aux["decisions"])]
        flips[which] = (sum(t[0] for t in f), sum(t[1] for t in f))
    masks, _ = check_generator(eng, PG, PD1, PD2, x, y2, z, B)
    if not report:
        eng.close()
        return
    dec = {}
    M.g_grads_manual(PG, PD1, PD2, x, y2, z, decisions=dec)
    f = [M.count_decision_flips(a, dec[k]) for a, k in zip(masks, ("G", "D_y2", "D_dem"))]
    flips["G"] = (sum(t[0] for t in f), sum(t[1] for t in f))
    print("seed %d: decisions taken differently from the free fp64 evaluation: %s" % (seed, flips))
    for k, (n, tot) in flips.items():
        # a unit flips when its pre-activation is within fp32 rounding of zero (or two pool candidates within rounding of
        # each other): ~1e-6 of the units.  A wrong kernel moves per cent of them.
        assert n <= max(8, 2e-5 * tot), (k, n, tot)
    eng.close()


@pytest.mark.parametrize("seed,noisy,trained", [(231, True, True), (3, False, False)])
def test_gradients_256_under_hip_masks(lib, seed, noisy, trained):
    """256x256 (BASELINE's resolution), batch 2: tie-free inputs with the penalty in its trained regime, and
    reference-like inputs (exactly flat regions outside the brain mask: max-pool ties and ReLU kinks everywhere) at the
    critics' random initialisation.  Per tensor 1e-4 on both."""
    from oracle import manual as M
    img, B = 256, 2
    PG, PD1, PD2, x, y2, z, ep = setup(img, B, seed, noisy=noisy, trained_regime=trained)
    eng = engine(img, B, PG, PD1, PD2)
    # the oracle's convolutions run in float32 and its parameter-gradient reductions in float64 here, as in the batch-32
    # test (float64 throughout, as at 64x64, is a minute of CPU per case: measured once, 3.9e-6 / 7.3e-6 / 1.2e-5 on the
    # tie-free case and 5.9e-6 / 3.8e-6 / 7.9e-6 on the reference-like one)
    dt = torch.float32
    for which, PD in (("D_y2", PD1), ("D_dem", PD2)):
        masks, _ = check_critic(eng, which, PD, PG, x, y2, z, ep, B, dtype=dt)
        if which == "D_y2" and noisy:
            # the free fp64 evaluation: how many of the 2.8e7 decisions did fp32 arithmetic take differently?  (At this
            # size an evaluation without any is the exception -- for the CPU oracle's own fp32 run as well.)
            real, fake = critic_real_fake(which, x, y2, eng.debug_tensor("g/out/gen_segmentation"))
            _, _, aux = M.critic_grads_manual(PD, real, fake, ep)
            f = [M.count_decision_flips(a, b) for a, b in zip(masks, aux["decisions"])]
            n, tot = sum(t[0] for t in f), sum(t[1] for t in f)
            print("256x256 %s: %d of %d decisions differ from the free fp64 evaluation" % (which, n, tot))
            # reference-like inputs are flat (exactly 0) outside the brain mask: whole regions of exact pool ties and
            # pre-activations that differ from zero by rounding only -- there the count says nothing
            if noisy:
                assert n <= 2e-5 * tot, (n, tot)
    check_generator(eng, PG, PD1, PD2, x, y2, z, B, dtype=dt)
    eng.close()


def test_headline_config_backward_batch32_256(lib):
    """BASELINE configs[1] at its own size (batch 32, 256x256x1, fp32): critic("D_y2"), critic("D_dem") and
    generator("grads") against the oracle on the whole batch -- the launches the benchmark times (8-channel-chunk
    igemm_conv_kernel<32,3,8,9,true> from 1536 items up, the persistent 5x5 kernels in backward-data and u-forward, the
    fused transposed-convolution kernels).  The oracle's convolutions run in fp32 here (a float64 batch-32 step is
    minutes of CPU) and its parameter-gradient reductions in float64 (oracle/manual.py _es / _rs), under the HIP path's
    decisions; per-tensor bound printed and asserted."""
    from oracle import depgan_oracle as O
    img, B = 256, 32
    t0 = time.time()
    PG, PD1, PD2, x, y2, z, ep = setup(img, B, 5, noisy=False)
    eng = engine(img, B, PG, PD1, PD2)
    attr = O.g_predict(PG, x, z)
    print("setup %.1f s" % (time.time() - t0))
    for which, PD in (("D_y2", PD1), ("D_dem", PD2)):
        check_critic(eng, which, PD, PG, x, y2, z, ep, B, dtype=torch.float32, attr=attr)
    check_generator(eng, PG, PD1, PD2, x, y2, z, B, dtype=torch.float32)
    eng.close()
