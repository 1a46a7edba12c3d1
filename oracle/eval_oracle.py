"""TEST INFRASTRUCTURE -- CPU restatement (NumPy) of the reference's per-subject evaluation, PARITY UNPINNED
(the reference cannot run here and ships no fixtures; see oracle/__init__.py).

Follows DEP-GAN_testing_4fold.py ("GE") statement by statement: the n_repeat mean prediction GE:616-628 and the
volume / Dice figures GE:637-790.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
import numpy as np


def mean_prediction(predict, x, mask2, n_repeat=10, noise_size=32, rng=None):
    """GE:616-628.  predict([x, noise]) -> (n,H,W,1)."""
    rng = rng if rng is not None else np.random
    output_img_pred_mean = np.zeros(mask2.shape)                                           # GE:617
    for _ in range(n_repeat):
        noise = rng.normal(size=(x.shape[0], noise_size, 1)).astype('float32')             # GE:620
        output_img_pred = predict([x, noise])                                              # GE:621
        output_img_pred = np.squeeze(output_img_pred)                                      # GE:622
        output_img_pred = np.multiply(output_img_pred, mask2)                              # GE:623
        output_img_pred_mean = output_img_pred_mean + output_img_pred                      # GE:624
    return output_img_pred_mean / float(n_repeat)                                          # GE:628


def _dice(fake, real, k, smooth=1e-7):
    """GE:746-748 (the same expression is used for all six figures)."""
    return (np.count_nonzero(fake[real == k] == k) * 2.0 + smooth) / \
        (smooth + np.count_nonzero(real[real == k] == k) + np.count_nonzero(fake[fake == k] == k))


def subject_metrics(x, pred, code_real, mask1, wmh1, mask2, wmh2, prob2, voxel_volume, TRSH_VAL):
    """GE:637-790 for one subject.  x: (n,H,W,nicg) = brain_prob__1tp after the nicg concat (GE:603-612);
    pred: (n,H,W) mean prediction; code_real: brain_code_2tp; voxel_volume = prod(pixdim)."""
    vol_1tp__ml = np.count_nonzero(np.multiply(mask1, wmh1)) * voxel_volume / 1000         # GE:637-641
    vol_2tp__ml = np.count_nonzero(np.multiply(mask2, wmh2)) * voxel_volume / 1000         # GE:647-651
    vol_1tp__ml_iam = np.count_nonzero(x >= TRSH_VAL) * voxel_volume / 1000                # GE:655-660
    vol_2tp__ml_iam = np.count_nonzero(np.copy(prob2) >= TRSH_VAL) * voxel_volume / 1000   # GE:664-669
    fake = x[:, :, :, 0] + pred                                                            # GE:675
    fake[fake < -1] = -1                                                                   # GE:676
    fake[fake > 1] = 1                                                                     # GE:677
    wmh_mask = np.zeros(fake.shape)
    wmh_mask[fake > TRSH_VAL] = 1                                                          # GE:679
    vol_out__ml = np.count_nonzero(np.multiply(mask2, wmh_mask)) * voxel_volume / 1000     # GE:681-684
    err_vol = vol_out__ml - vol_2tp__ml                                                    # GE:688
    mse_vol = np.mean((vol_2tp__ml - vol_out__ml) ** 2)                                    # GE:689
    true_pred = true_prog = true_regg = prog = regg = 0                                    # GE:692-707
    if (vol_2tp__ml - vol_1tp__ml) >= 0:
        prog = 1
        if vol_out__ml - vol_1tp__ml >= 0:
            true_pred = 1
            true_prog = 1
    else:
        regg = 1
        if vol_out__ml - vol_1tp__ml < 0:
            true_pred = 1
            true_regg = 1
    change_fake = np.squeeze(np.zeros(code_real.shape))                                    # GE:714
    change_real = np.squeeze(code_real)                                                    # GE:715
    prob_1tp = np.squeeze(np.copy(x[:, :, :, 0]))                                          # GE:716-717
    f = np.squeeze(np.copy(fake))
    change_fake[np.all([f < TRSH_VAL, prob_1tp >= TRSH_VAL], axis=0)] = 1                  # GE:722-727 shrink
    change_fake[np.all([f >= TRSH_VAL, prob_1tp < TRSH_VAL], axis=0)] = 2                  # GE:729-734 grow
    change_fake[np.all([f >= TRSH_VAL, prob_1tp >= TRSH_VAL], axis=0)] = 3                 # GE:736-741 stay
    dice_1 = _dice(change_fake, change_real, 1)                                            # GE:745-758
    dice_2 = _dice(change_fake, change_real, 2)
    dice_3 = _dice(change_fake, change_real, 3)
    dice_4 = _dice(change_fake > 0, change_real > 0, 1)                                    # GE:760-769
    a_fake = ((change_fake == 1) + (change_fake == 2)) > 0                                 # GE:771-780
    a_real = ((change_real == 1) + (change_real == 2)) > 0
    dice_5 = _dice(a_fake, a_real, 1)                                                      # GE:783-786
    dice_6 = _dice(change_fake == 3, change_real == 3, 1)                                  # GE:788-797
    avg_all_dice = (dice_1 + dice_2 + dice_3) / 3.0
    avg_dice__56 = (dice_5 + dice_6) / 2.0
    vol_dsc = [true_pred, prog, true_prog, regg, true_regg, vol_1tp__ml, vol_2tp__ml, vol_out__ml, mse_vol, err_vol,
               dice_5, dice_6, avg_dice__56, dice_1, dice_2, dice_3, dice_4, avg_all_dice]   # GE:804-808
    return {"vol_dsc": vol_dsc, "vol_1tp_ml_im": vol_1tp__ml_iam, "vol_2tp_ml_im": vol_2tp__ml_iam}
