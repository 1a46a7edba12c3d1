"""Evaluation step after the hot path (DEP-GAN_testing_4fold.py "GE":616-807).

    pred = predict_mean(netG, brain_prob__1tp, n_repeat=10, mask=icv_and_sl_mask_2tp)        # GE:616-628
    m = dem_metrics(brain_prob__1tp, pred, brain_code_2tp, icv_and_sl_mask_1tp, brain_wmh_1tp,
                    icv_and_sl_mask_2tp, brain_wmh_2tp, brain_prob__2tp, voxel_volume, TRSH_VAL) # GE:637-790
    m["vol_dsc"]   # the 18-entry row the script appends per subject (GE:806-808)

The mean over the n_repeat noise draws is accumulated on the device in float64 like the reference's np.zeros
accumulator (GE:617), and the volumes / Dice figures come from one integer census kernel (exact counts) that thresholds
that float64 mean in float64; only the two dozen integers travel to the host, where the reference's own scalar
algebra is applied.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

NCOUNT = 20


def _torch():
    import torch
    return torch


def _dev(a, device, dtype=None):
    torch = _torch()
    if a is None:
        return None
    dtype = dtype or torch.float32
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    npdt = np.float64 if dtype == torch.float64 else np.float32
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=npdt)).to(device)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def predict_mean(netG, x, n_repeat=10, mask=None, noise_size=32, rng=None, batch_size=32):
    """Mean of n_repeat generator predictions with fresh N(0,1) noise, each multiplied by `mask` (GE:616-628).
    x: (n, H, W, nicg); mask: (n, H, W) or None.  Returns a float64 CUDA tensor (n, H, W): the reference's running
    sum is float64 and so is the mean it thresholds (GE:617, 628)."""
    torch = _torch()
    lib = _lib.load()
    rng = rng if rng is not None else np.random
    n = len(x)
    eng = netG._ensure_engine(min(batch_size, n))
    dev = eng.device
    xd = _dev(x, dev)
    md = _dev(mask, dev)
    if md is not None and md.numel() != n * eng.height * eng.width:
        raise ValueError("mask must have one value per output pixel")
    acc = torch.zeros((n, eng.height, eng.width), dtype=torch.float64, device=dev)                  # GE:617
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(n_repeat):
        noise = rng.normal(size=(n, noise_size, 1)).astype("float32")               # GE:620
        pred = eng.g_forward(xd, noise)                                               # GE:621
        _lib.check(lib.depgan_eval_accumulate(_p(pred), _p(md), _p(acc), acc.numel(), stream),
                   "depgan_eval_accumulate")                                          # GE:623-624
    _lib.check(lib.depgan_eval_divide(_p(acc), acc.numel(), float(n_repeat), stream), "depgan_eval_divide")  # GE:628
    return acc


def census(x, pred, code_real=None, mask1=None, wmh1=None, mask2=None, wmh2=None, prob2=None, thr=0.5, device=None):
    """The 20 integer counts of depgan_eval_counts (include/depgan.h) as a Python list."""
    torch = _torch()
    lib = _lib.load()
    if device is None:
        device = pred.device if isinstance(pred, torch.Tensor) else torch.device("cuda:%d" % torch.cuda.current_device())
    xd = _dev(x, device)
    if xd.dim() < 2:
        raise ValueError("x must be (..., nicg)")
    nicg = int(xd.shape[-1])
    npix = xd.numel() // nicg
    arrs = [_dev(pred, device, torch.float64)] + [_dev(a, device) for a in (code_real, mask1, wmh1, mask2, wmh2, prob2)]
    for name, a in zip(("pred", "code_real", "mask1", "wmh1", "mask2", "wmh2", "prob2"), arrs):
        if a is not None and a.numel() != npix:
            raise ValueError("%s must have %d elements, got %d" % (name, npix, a.numel()))
    out = (C.c_longlong * NCOUNT)()
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _lib.check(lib.depgan_eval_counts(_p(xd), nicg, *[_p(a) for a in arrs], npix, float(thr), out, stream),
               "depgan_eval_counts")
    return [int(v) for v in out]


def _dice(both, real, fake, smooth=1e-7):
    return (both * 2.0 + smooth) / (smooth + real + fake)                             # GE:746-748


def metrics_from_census(c, voxel_volume):
    """The reference's scalar algebra on the census (GE:640-808)."""
    vol_1tp__ml = c[0] * voxel_volume / 1000                                          # GE:640-641
    vol_2tp__ml = c[1] * voxel_volume / 1000                                          # GE:650-651
    vol_1tp__ml_iam = c[2] * voxel_volume / 1000                                      # GE:659-660
    vol_2tp__ml_iam = c[3] * voxel_volume / 1000                                      # GE:668-669
    vol_out__ml = c[4] * voxel_volume / 1000                                          # GE:683-684
    err_vol = vol_out__ml - vol_2tp__ml                                               # GE:688
    mse_vol = float(np.mean((vol_2tp__ml - vol_out__ml) ** 2))                        # GE:689
    true_pred = true_prog = true_regg = prog = regg = 0                               # GE:692-707
    if (vol_2tp__ml - vol_1tp__ml) >= 0:
        prog = 1
        if vol_out__ml - vol_1tp__ml >= 0:
            true_pred = true_prog = 1
    else:
        regg = 1
        if vol_out__ml - vol_1tp__ml < 0:
            true_pred = true_regg = 1
    dice_1, dice_2, dice_3 = (_dice(*c[5 + 3 * k:8 + 3 * k]) for k in range(3))       # GE:745-758
    dice_4 = _dice(*c[14:17])                                                         # GE:760-769
    dice_5 = _dice(*c[17:20])                                                         # GE:771-786
    dice_6 = _dice(*c[11:14])                                                         # GE:788-797 (== dice_3)
    avg_all_dice = (dice_1 + dice_2 + dice_3) / 3.0
    avg_dice__56 = (dice_5 + dice_6) / 2.0
    vol_dsc = [true_pred, prog, true_prog, regg, true_regg, vol_1tp__ml, vol_2tp__ml, vol_out__ml, mse_vol, err_vol,
               dice_5, dice_6, avg_dice__56, dice_1, dice_2, dice_3, dice_4, avg_all_dice]
    return {"vol_dsc": vol_dsc, "vol_1tp_ml": vol_1tp__ml, "vol_2tp_ml": vol_2tp__ml, "vol_out_ml": vol_out__ml,
            "vol_1tp_ml_im": vol_1tp__ml_iam, "vol_2tp_ml_im": vol_2tp__ml_iam, "err_vol": err_vol,
            "mse_vol": mse_vol, "true_pred": true_pred, "prog": prog, "true_prog": true_prog, "regg": regg,
            "true_regg": true_regg, "dice": [dice_1, dice_2, dice_3, dice_4, dice_5, dice_6],
            "avg_all_dice": avg_all_dice, "avg_dice_56": avg_dice__56, "census": list(c)}


def dem_metrics(x, pred, code_real, mask1, wmh1, mask2, wmh2, prob2, voxel_volume, thr):
    """All per-subject figures of GE:637-790 for one volume of slices."""
    return metrics_from_census(census(x, pred, code_real, mask1, wmh1, mask2, wmh2, prob2, thr), voxel_volume)
