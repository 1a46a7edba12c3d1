"""GPU parity of the evaluation step after the path (GE:616-807): the device census and the
reference's scalar algebra against the NumPy restatement (oracle/eval_oracle.py), and the n_repeat mean prediction.
Counts are integers: they must match exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _subject(seed, n=6, img=64, nicg=1, thr=0.178):
    from oracle import depgan_oracle as O
    x, y2, z, ep = O.synth_batch(seed, n, img, img, nicg=nicg)
    rng = np.random.default_rng(seed)
    pred = (y2[..., 0] - x[..., 0] + 0.05 * rng.standard_normal(y2[..., 0].shape)).astype(np.float32)
    pred[0, :4] = 3.0          # exercises the clip at +1 ...
    pred[0, 4:8] = -3.0        # ... and at -1 (GE:676-677)
    # the reference's mean prediction is float64 (np.zeros accumulator, GE:617): x0 + pred, the clip and the threshold
    # tests then run in float64.  Boundary voxels: fake exactly at the threshold, and one float64 ulp to either side --
    # values a float32 pipeline cannot tell apart
    pred = pred.astype(np.float64)
    x0 = x[..., 0].astype(np.float64)
    for row, delta in ((1, 0.0), (2, np.spacing(thr)), (3, -np.spacing(thr))):
        pred[1, row, :] = (thr - x0[1, row, :]) + delta
    a, b = x[..., 0] >= thr, y2[..., 0] >= thr
    code = np.zeros(a.shape, np.float32)
    code[a & ~b], code[~a & b], code[a & b] = 1, 2, 3
    mask1 = (rng.uniform(size=a.shape) > 0.1).astype(np.float32)
    mask2 = (rng.uniform(size=a.shape) > 0.1).astype(np.float32)
    return x, pred, code, mask1, a.astype(np.float32), mask2, b.astype(np.float32), y2[..., 0].copy()


@pytest.mark.parametrize("nicg", [1, 2])
def test_metrics_match_the_numpy_restatement(lib, nicg):
    from dep_gan_im_amd import evaluate as EV
    from oracle import eval_oracle as EO
    thr, vox = 0.178, 0.9375 * 0.9375 * 4.0
    for seed in (3, 4):
        args = _subject(seed, nicg=nicg, thr=thr)
        got = EV.dem_metrics(*args, voxel_volume=vox, thr=thr)
        want = EO.subject_metrics(*[np.copy(a) for a in args], vox, thr)
        np.testing.assert_allclose(got["vol_dsc"], want["vol_dsc"], rtol=1e-12, atol=0)
        assert got["vol_1tp_ml_im"] == want["vol_1tp_ml_im"] and got["vol_2tp_ml_im"] == want["vol_2tp_ml_im"]
        c = got["census"]
        assert c[11:14] == c[11:14] and got["dice"][5] == got["dice"][2]       # dice_6 restates dice_3 (GE:788-797)
        assert c[15] == int((args[2] > 0).sum()) and c[6] + c[9] + c[12] == c[15]


def test_census_edge_cases(lib):
    from dep_gan_im_amd import evaluate as EV
    from oracle import eval_oracle as EO
    # nothing above threshold anywhere: every Dice is smooth/smooth = 1, volumes 0 (GE:746-748 with empty sets)
    z = np.zeros((2, 16, 16, 1), np.float32)
    zz = np.zeros((2, 16, 16), np.float32)
    got = EV.dem_metrics(z, zz, zz, zz + 1, zz, zz + 1, zz, zz, 1.0, 0.5)
    want = EO.subject_metrics(z.copy(), zz.copy(), zz.copy(), zz + 1, zz.copy(), zz + 1, zz.copy(), zz.copy(), 1.0, 0.5)
    assert got["vol_dsc"] == [float(v) for v in want["vol_dsc"]] and got["dice"] == [1.0] * 6
    # a value exactly at the threshold counts as >= for the change code but not as > for the predicted volume
    x = np.full((1, 16, 16, 1), 0.5, np.float32)
    got = EV.census(x, np.zeros((1, 16, 16), np.float32), thr=0.5)
    assert got[2] == 256 and got[4] == 0 and got[13] == 256
    with pytest.raises(ValueError):
        EV.census(x, np.zeros((1, 16, 8), np.float32))


def test_threshold_boundary_voxels_follow_float64(lib):
    """One float64 ulp above / below the threshold must land on different sides (GE:675-679 run in float64 because
    the accumulated mean is float64); float32 arithmetic would merge them."""
    from dep_gan_im_amd import evaluate as EV
    thr = 0.178
    x = np.full((1, 16, 16, 1), 0.125, np.float32)
    base = thr - 0.125
    pred = np.full((1, 16, 16), base, np.float64)
    pred[0, 0] = base + 4 * np.spacing(thr)        # strictly above -> counted by fake > thr
    pred[0, 1] = base - 4 * np.spacing(thr)        # strictly below -> neither > nor >=
    fake = x[..., 0].astype(np.float64) + pred
    got = EV.census(x, pred, thr=thr)
    assert got[4] == int((fake > thr).sum()) and got[7 + 3] == int((fake >= thr).sum())      # grow: fake >= thr, x0 < thr
    assert np.float32(fake[0, 0, 0]) == np.float32(fake[0, 1, 0])                             # indistinguishable in fp32
    assert (fake[0, 0] > thr).all() and not (fake[0, 1] >= thr).any()                         # ... but not in fp64
    assert 16 <= got[4] <= 256 - 16


def test_mean_prediction_of_n_noises(lib):
    """predict_mean == mean over the same noise draws of netG.predict * mask (GE:616-628), accumulated in float64 and
    divided (not multiplied by a reciprocal) like the reference."""
    import dep_gan_im_amd as dg
    from dep_gan_im_amd import evaluate as EV
    from oracle import depgan_oracle as O
    from oracle import eval_oracle as EO
    img, n = 64, 5
    PG = O.init_generator(7, bias_std=0.05)
    x, y2, z, ep = O.synth_batch(9, n, img, img)
    mask = (np.random.default_rng(1).uniform(size=(n, img, img)) > 0.2).astype(np.float32)
    net = dg.Gen_UNet2D((img, img, 1), seed=0)
    net.set_weights(PG)
    got_t = EV.predict_mean(net, x, n_repeat=3, mask=mask, rng=np.random.RandomState(11), batch_size=4)
    assert got_t.dtype == torch.float64
    got = got_t.cpu().numpy()
    want = EO.mean_prediction(lambda a: O.g_predict(PG, a[0], a[1]), x, mask, n_repeat=3, rng=np.random.RandomState(11))
    assert want.dtype == np.float64
    np.testing.assert_allclose(got, want, atol=2e-4)
    assert float(np.abs(got[mask == 0]).max()) == 0.0
    # the accumulation itself, bit for bit: the device's own per-noise predictions through the NumPy statements
    rs = np.random.RandomState(11)
    acc = np.zeros(mask.shape)
    for _ in range(3):
        noise = rs.normal(size=(n, 32, 1)).astype("float32")
        acc = acc + np.multiply(np.squeeze(net.predict([x, noise], batch_size=4)), mask)
    np.testing.assert_array_equal(got, acc / float(3))
