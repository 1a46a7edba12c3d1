"""CPU: the host half of the evaluation step (dep_gan_im_amd/evaluate.metrics_from_census, GE:640-808) against the
NumPy restatement of the whole step (oracle/eval_oracle.py), with the census itself formed in NumPy here."""
import numpy as np

from dep_gan_im_amd.evaluate import NCOUNT, metrics_from_census
from oracle import depgan_oracle as O
from oracle import eval_oracle as EO


def numpy_census(x, pred, code, mask1, wmh1, mask2, wmh2, prob2, thr):
    x0 = x[..., 0]
    fake = np.clip(x0 + pred, -1, 1)
    fhi, xhi = fake >= thr, x0 >= thr
    fc = np.zeros(fake.shape, np.int64)
    fc[~fhi & xhi], fc[fhi & ~xhi], fc[fhi & xhi] = 1, 2, 3
    c = [np.count_nonzero(mask1 * wmh1), np.count_nonzero(mask2 * wmh2), np.count_nonzero(x >= thr),
         np.count_nonzero(prob2 >= thr), np.count_nonzero((fake > thr) & (mask2 != 0))]
    for k in (1, 2, 3):
        c += [np.count_nonzero((fc == k) & (code == k)), np.count_nonzero(code == k), np.count_nonzero(fc == k)]
    c += [np.count_nonzero((fc > 0) & (code > 0)), np.count_nonzero(code > 0), np.count_nonzero(fc > 0)]
    ch_f, ch_r = (fc == 1) | (fc == 2), (code == 1) | (code == 2)
    c += [np.count_nonzero(ch_f & ch_r), np.count_nonzero(ch_r), np.count_nonzero(ch_f)]
    assert len(c) == NCOUNT
    return [int(v) for v in c]


def test_scalar_algebra_follows_the_reference_statements():
    thr, vox = 0.178, 3.5
    for seed, flip in ((1, 1.0), (2, -1.0)):                      # growing and shrinking subjects (GE:697-707)
        x, y2, z, ep = O.synth_batch(seed, 4, 64, 64)
        rng = np.random.default_rng(seed)
        pred = (flip * (y2[..., 0] - x[..., 0]) + 0.03 * rng.standard_normal(y2[..., 0].shape)).astype(np.float32)
        a, b = x[..., 0] >= thr, y2[..., 0] >= thr
        code = np.zeros(a.shape, np.float32)
        code[a & ~b], code[~a & b], code[a & b] = 1, 2, 3
        m1 = (rng.uniform(size=a.shape) > 0.2).astype(np.float32)
        m2 = (rng.uniform(size=a.shape) > 0.2).astype(np.float32)
        args = (x, pred, code, m1, a.astype(np.float32), m2, b.astype(np.float32), y2[..., 0])
        got = metrics_from_census(numpy_census(*args, thr), vox)
        want = EO.subject_metrics(*[np.copy(v) for v in args], vox, thr)
        np.testing.assert_allclose(got["vol_dsc"], want["vol_dsc"], rtol=1e-12)
        assert got["prog"] + got["regg"] == 1 and got["true_pred"] == got["true_prog"] + got["true_regg"]
