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


def train_epoch(trainers, data_1tp, data_2tp, batchSize=16, Diters=5, k_noise=10, noiseSize=32, state=None,
                rng=None, on_gen_iteration=None, shuffle=True):
    """One pass of the `for epoch in range(niter)` body (GT:780-894).

    trainers: object with netD_y2_train / netD_dem_train / netG_no_update / netG_train
    data_1tp: (N,H,W,nicg) baseline maps (+FLAIR), data_2tp: (N,H,W,1) follow-up maps.
    Returns (data_1tp, data_2tp) in the order used (the reference re-assigns the shuffled arrays).
    """
    state = state if state is not None else ScheduleState()
    rng = rng if rng is not None else np.random
    i = 0
    ii = 0
    if shuffle:                                                     # GT:783-787
        indices = np.arange(data_1tp.shape[0])
        rng.shuffle(indices)
        data_1tp = data_1tp[indices]
        data_2tp = data_2tp[indices]
    batches = data_1tp.shape[0] // batchSize                         # GT:789
    errD_real = errD_fake = errD_real_dem = errD_fake_dem = 0.0
    real_data_1tp = real_data_2tp = None
    while i < batches:                                               # GT:791
        if state.gen_iterations < 25 or state.gen_iterations % 500 == 0:   # GT:792-797
            _Diters = _Diters_dem = 100
        else:
            _Diters = _Diters_dem = Diters
        j = jj = 0
        while j < _Diters and i < batches:                          # GT:802-814
            j += 1
            real_data_1tp = data_1tp[i * batchSize:(i + 1) * batchSize]
            real_data_2tp = data_2tp[i * batchSize:(i + 1) * batchSize]
            i += 1
            noise = rng.normal(size=(batchSize, noiseSize, 1))
            ep = rng.uniform(size=(batchSize, 1, 1, 1))
            errD_real, errD_fake = trainers.netD_y2_train([real_data_2tp, real_data_1tp, noise, ep])
            state.crit_iterations += 1
        while jj < _Diters_dem and ii < batches:                    # GT:817-829
            jj += 1
            real_data_1tp = data_1tp[ii * batchSize:(ii + 1) * batchSize]
            real_data_2tp = data_2tp[ii * batchSize:(ii + 1) * batchSize]
            ii += 1
            noise = rng.normal(size=(batchSize, noiseSize, 1))
            ep = rng.uniform(size=(batchSize, 1, 1, 1))
            errD_real_dem, errD_fake_dem = trainers.netD_dem_train([real_data_2tp, real_data_1tp, noise, ep])
            state.crit_dem_iterations += 1
        # generator: best of k_noise on the batch the DEM loop used last (GT:868-878)
        noises = rng.normal(size=(k_noise, batchSize, noiseSize, 1)).astype("float32")
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
