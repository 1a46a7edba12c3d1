"""Host-side training schedule of the reference (GT:779-894), driving the four closures.

Reproduces, quirk for quirk:
  * `_Diters = 100` while `gen_iterations < 25` or every 500th generator iteration, else `Diters`
    (GT:792-797); the counters are not reset between folds (GT:47-50, 894);
  * the critic-Y2 loop consumes batches through cursor `i` (which also ends the epoch), the
    critic-DEM loop through its own cursor `ii` that restarts at 0 every epoch (GT:781-782, 802-829);
  * the generator is evaluated with `k_noise = 10` noises on the batch LAST USED BY THE DEM LOOP,
    trained once with the arg-min noise of the TOTAL loss (GT:868-878);
  * noise ~ N(0,1) (batch,32,1) and ep ~ U[0,1) (batch,1,1,1) drawn in float64 (GT:807-808, 822-823),
    the 10 generator noises as float32 (GT:870);
  * the per-epoch shuffle (GT:783-787).
Beyond the reference: `rank` / `world` shard every global batch by sample index, and with
`fused` a whole generator iteration -- both critic loops, the best-of-k search with its arg-min and the
generator update -- is one library call with one host synchronisation (depgan_gen_iteration).
Out of scope here: TensorBoard logging, validation images, HDF5 saves --
the `on_gen_iteration` callback receives every scalar the reference logs.
"""
from __future__ import annotations

import numpy as np


class ScheduleState:
    """Module-level counters of the reference script (GT:47-50)."""

    def __init__(self):
        self.gen_iterations = 0
        self.crit_iterations = 0
        self.crit_dem_iterations = 0
        self.errG = 0.0


def _is_tensor(a):
    return type(a).__module__.startswith("torch")


def _take(data, order):
    """data[order] for NumPy arrays and (device-resident) torch tensors alike."""
    if _is_tensor(data):
        import torch
        return data[torch.as_tensor(order, device=data.device)]
    return data[order]


def _rank_batches(data, first, n, batchSize, rank, world):
    """The n consecutive batches `first .. first+n-1` of this rank as (array, batch_stride): global batch g holds the
    samples [g*world*batchSize, (g+1)*world*batchSize), rank r its r-th slice of batchSize samples.
    A device-resident tensor is viewed in place (stride world*batchSize); a host array is gathered densely."""
    gb = batchSize * world
    if n == 0:
        return None, batchSize
    if world == 1:
        return data[first * batchSize:(first + n) * batchSize], batchSize
    if _is_tensor(data):
        lo = first * gb + rank * batchSize
        return data[lo:(first + n - 1) * gb + (rank + 1) * batchSize], gb
    return np.concatenate([data[(first + j) * gb + rank * batchSize:(first + j) * gb + (rank + 1) * batchSize]
                           for j in range(n)]), batchSize


def train_epoch(trainers, data_1tp, data_2tp, batchSize=16, Diters=5, k_noise=10, noiseSize=32, state=None,
                rng=None, on_gen_iteration=None, shuffle=True, rank=0, world=1, fused=None):
    """One pass of the `for epoch in range(niter)` body (GT:780-894).

    trainers: object with netD_y2_train / netD_dem_train / netG_no_update / netG_train (and optionally
    netG_no_update_many / gen_iteration).
    data_1tp: (N,H,W,nicg) baseline maps (+FLAIR), data_2tp: (N,H,W,1) follow-up maps; NumPy arrays or torch tensors
    already resident in HBM (a 288-GB device holds the whole training set: no per-batch host copies then).
    rank, world: data parallelism.  batchSize stays the PER-RANK batch; a schedule step consumes a global
    batch of world*batchSize samples of which this rank takes its slice, and every random draw (shuffle, noise, ep)
    is made for the global batch from `rng` -- which therefore must be seeded identically on all ranks -- and sliced,
    so that the union of the ranks' inputs is exactly what one process with batch world*batchSize would have fed.
    fused: True = every generator iteration is ONE enqueue with one host synchronisation (trainers.gen_iteration);
    None = do so when the trainers offer it; False = closure by closure like the reference.
    Returns (data_1tp, data_2tp) in the order used (the reference re-assigns the shuffled arrays).
    """
    state = state if state is not None else ScheduleState()
    rng = rng if rng is not None else np.random
    if fused is None:
        fused = hasattr(trainers, "gen_iteration")
    gb = batchSize * world
    lo, hi = rank * batchSize, (rank + 1) * batchSize
    i = 0
    ii = 0
    if shuffle:                                                     # GT:783-787
        indices = np.arange(data_1tp.shape[0])
        rng.shuffle(indices)
        data_1tp = _take(data_1tp, indices)
        data_2tp = _take(data_2tp, indices)
    batches = data_1tp.shape[0] // gb                                # GT:789
    errD_real = errD_fake = errD_real_dem = errD_fake_dem = 0.0
    real_data_1tp = real_data_2tp = None

    def batch(k):
        return (data_1tp[k * gb + lo:k * gb + hi], data_2tp[k * gb + lo:k * gb + hi])

    def draw():                                                     # GT:807-808, 822-823 (float64, global batch)
        noise = rng.normal(size=(gb, noiseSize, 1))[lo:hi]
        ep = rng.uniform(size=(gb, 1, 1, 1))[lo:hi]
        return noise, ep

    while i < batches:                                               # GT:791
        if state.gen_iterations < 25 or state.gen_iterations % 500 == 0:   # GT:792-797
            _Diters = _Diters_dem = 100
        else:
            _Diters = _Diters_dem = Diters
        if fused:
            n_y2, n_dem = min(_Diters, batches - i), min(_Diters_dem, batches - ii)
            d1 = [draw() for _ in range(n_y2)]                       # same draw order as the loops below
            d2 = [draw() for _ in range(n_dem)]
            noises = rng.normal(size=(k_noise, gb, noiseSize, 1)).astype("float32")[:, lo:hi]
            xa, stride = _rank_batches(data_1tp, i, n_y2, batchSize, rank, world)
            ya, _ = _rank_batches(data_2tp, i, n_y2, batchSize, rank, world)
            xb, stride_b = _rank_batches(data_1tp, ii, n_dem, batchSize, rank, world)
            yb, _ = _rank_batches(data_2tp, ii, n_dem, batchSize, rank, world)
            if n_y2 and n_dem and stride != stride_b:
                raise AssertionError("inconsistent batch strides")
            last = (ii + n_dem - 1) if n_dem else (i + n_y2 - 1)     # the batch `real_data_*` names at GT:868
            real_data_1tp, real_data_2tp = batch(last)
            stack = lambda d, j: np.stack([t[j] for t in d]) if d else None   # noqa: E731
            cy, cd, evals, tr_out, best = trainers.gen_iteration(
                (xa, ya, stack(d1, 0), stack(d1, 1), n_y2), (xb, yb, stack(d2, 0), stack(d2, 1), n_dem),
                (real_data_1tp, real_data_2tp, noises), stride if n_y2 else stride_b)
            i += n_y2
            ii += n_dem
            state.crit_iterations += n_y2
            state.crit_dem_iterations += n_dem
            if cy:
                errD_real, errD_fake = cy[-1]
            if cd:
                errD_real_dem, errD_fake_dem = cd[-1]
            losses_errG = [o[0] for o in evals]
            errG, errG_CY2, errG_DEM, errG_MSE, errG_VOL, errG_WMH = tr_out
        else:
            j = jj = 0
            while j < _Diters and i < batches:                          # GT:802-814
                j += 1
                real_data_1tp, real_data_2tp = batch(i)
                i += 1
                noise, ep = draw()
                errD_real, errD_fake = trainers.netD_y2_train([real_data_2tp, real_data_1tp, noise, ep])
                state.crit_iterations += 1
            while jj < _Diters_dem and ii < batches:                    # GT:817-829
                jj += 1
                real_data_1tp, real_data_2tp = batch(ii)
                ii += 1
                noise, ep = draw()
                errD_real_dem, errD_fake_dem = trainers.netD_dem_train([real_data_2tp, real_data_1tp, noise, ep])
                state.crit_dem_iterations += 1
            # generator: best of k_noise on the batch the DEM loop used last (GT:868-878)
            noises = rng.normal(size=(k_noise, gb, noiseSize, 1)).astype("float32")[:, lo:hi]
            if hasattr(trainers, "netG_no_update_many"):                # one enqueue, one host sync for the k calls
                losses_errG = [o[0] for o in trainers.netG_no_update_many([real_data_1tp, real_data_2tp, noises])]
            else:
                losses_errG = []
                for k in range(k_noise):
                    out = trainers.netG_no_update([real_data_1tp, real_data_2tp, noises[k]])
                    losses_errG.append(out[0])
            best = int(np.array(losses_errG).argmin(0))
            errG, errG_CY2, errG_DEM, errG_MSE, errG_VOL, errG_WMH = trainers.netG_train(
                [real_data_1tp, real_data_2tp, noises[best]])
        state.errG = errG
        if on_gen_iteration is not None:
            on_gen_iteration(dict(gen_iterations=state.gen_iterations, i=i, ii=ii, batches=batches,
                                  errD=errD_real - errD_fake, errD_real=errD_real, errD_fake=errD_fake,
                                  errD_dem=errD_real_dem - errD_fake_dem, errD_real_dem=errD_real_dem,
                                  errD_fake_dem=errD_fake_dem, errG=errG, errG_CY2=errG_CY2, errG_DEM=errG_DEM,
                                  errG_MSE=errG_MSE, errG_VOL=errG_VOL, errG_WMH=errG_WMH, best_noise=best,
                                  losses_errG=list(losses_errG), Diters=_Diters))
        state.gen_iterations += 1                                    # GT:894
    return data_1tp, data_2tp
