"""CPU: the host-side schedule (GT:779-894) against a recording double of the four closures."""
import numpy as np

from dep_gan_im_amd.schedule import ScheduleState, train_epoch


class Recorder:
    def __init__(self):
        self.calls = []

    def netD_y2_train(self, inp):
        y2, x, z, ep = inp
        assert z.dtype == np.float64 and ep.dtype == np.float64 and z.shape[1:] == (32, 1) and ep.shape[1:] == (1, 1, 1)
        self.calls.append(("y2", float(x[0, 0, 0, 0])))
        return [0.5, 0.25]

    def netD_dem_train(self, inp):
        y2, x, z, ep = inp
        self.calls.append(("dem", float(x[0, 0, 0, 0])))
        return [0.125, 0.0625]

    def netG_no_update(self, inp):
        x, y2, z = inp
        assert z.dtype == np.float32
        self.calls.append(("eval", float(x[0, 0, 0, 0]), float(z[0, 0, 0])))
        return [float(z[0, 0, 0]), 0, 0, 0, 0, 0]       # total loss = first noise value -> arg-min is checkable

    def netG_train(self, inp):
        x, y2, z = inp
        self.calls.append(("train", float(x[0, 0, 0, 0]), float(z[0, 0, 0])))
        return [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]


def _data(n, bs):
    x = np.zeros((n * bs, 4, 4, 1), np.float32)
    for b in range(n):
        x[b * bs:(b + 1) * bs] = b          # batch id readable from any pixel
    return x, x.copy()


def test_warmup_uses_100_critic_iterations_and_separate_cursors():
    bs = 2
    x, y = _data(7, bs)
    rec, st, log = Recorder(), ScheduleState(), []
    train_epoch(rec, x, y, batchSize=bs, Diters=5, state=st, rng=np.random.RandomState(0), on_gen_iteration=log.append,
                shuffle=False)
    kinds = [c[0] for c in rec.calls]
    # gen_iterations 0 < 25: both critic loops run min(100, batches) = 7 steps, then 10 evals + 1 train; epoch ends (i == batches)
    assert kinds == ["y2"] * 7 + ["dem"] * 7 + ["eval"] * 10 + ["train"]
    assert [c[1] for c in rec.calls[:7]] == list(range(7)) and [c[1] for c in rec.calls[7:14]] == list(range(7))
    assert st.gen_iterations == 1 and st.crit_iterations == 7 and st.crit_dem_iterations == 7
    assert log[0]["Diters"] == 100 and log[0]["errD"] == 0.25 and log[0]["errD_dem"] == 0.0625


def test_steady_state_5_5_10_1_and_generator_uses_last_dem_batch_and_argmin_noise():
    bs = 2
    x, y = _data(12, bs)
    rec, st = Recorder(), ScheduleState()
    st.gen_iterations = 26                                    # past the warm-up, not a multiple of 500
    train_epoch(rec, x, y, batchSize=bs, Diters=5, state=st, rng=np.random.RandomState(1), shuffle=False)
    kinds = [c[0] for c in rec.calls]
    block = ["y2"] * 5 + ["dem"] * 5 + ["eval"] * 10 + ["train"]
    # 12 batches: i runs 5,10,12 -> three generator iterations; the last Y2 loop has only 2 batches left
    assert kinds == block + block + ["y2"] * 2 + ["dem"] * 2 + ["eval"] * 10 + ["train"]
    # cursor ii lags i: second DEM loop uses batches 5..9, third 10..11
    dem_batches = [c[1] for c in rec.calls if c[0] == "dem"]
    assert dem_batches == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]
    # generator sees the batch last used by the DEM loop (4, then 9, then 11)
    trains = [c for c in rec.calls if c[0] == "train"]
    assert [t[1] for t in trains] == [4.0, 9.0, 11.0]
    # and is trained with the arg-min noise of the 10 evaluations just before it
    idx = [k for k, c in enumerate(rec.calls) if c[0] == "train"]
    for k in idx:
        evals = rec.calls[k - 10:k]
        assert rec.calls[k][2] == min(e[2] for e in evals)
    assert st.gen_iterations == 29


def test_every_500th_iteration_is_a_long_critic_phase_and_shuffle_keeps_pairs():
    bs = 1
    x, y = _data(6, bs)
    y = y + 100
    rec, st = Recorder(), ScheduleState()
    st.gen_iterations = 500
    xs, ys = train_epoch(rec, x, y, batchSize=bs, Diters=2, state=st, rng=np.random.RandomState(3), shuffle=True)
    assert [c[0] for c in rec.calls][:12] == ["y2"] * 6 + ["dem"] * 6
    np.testing.assert_array_equal(ys - xs, np.full_like(xs, 100))          # pairs stay aligned
    assert sorted(xs[:, 0, 0, 0].tolist()) == [0, 1, 2, 3, 4, 5]
