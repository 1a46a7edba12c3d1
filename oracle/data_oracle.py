"""ORACLE (test infrastructure, not product): NumPy restatement of the reference's data step.

PARITY UNPINNED, like the rest of oracle/: the reference scripts cannot run here (Python 2, nibabel / Keras absent)
and ship no data, so these are the reference's NumPy statements retyped, each with its GT line
(GT = DEP-GAN_PROB_IM_twoCritics_training_4fold.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import numpy as np


def data_prep(image):
    """GT:106-118: (X, Y, Z) volume -> (Z, X, Y, 1) float32 slices."""
    images = []
    for z in range(image.shape[2]):
        images += [image[:, :, z]]
    images = np.array(images, dtype="float32")
    return np.expand_dims(images, axis=3)


def data_prep_save(image_data):
    """GT:121-127: network output (Z, X, Y, 1) -> volume orientation for saving."""
    image_data = np.squeeze(image_data)
    output_img = np.swapaxes(image_data, 0, 2)
    output_img = np.rot90(output_img)
    return output_img[::-1, ...]


def map_image_to_intensity_range(image, min_o, max_o, percentiles=0):
    """GT:130-146."""
    if image.dtype in [np.uint8, np.uint16, np.uint32]:
        assert min_o >= 0
    if image.dtype == np.uint8:
        assert max_o <= 255
    min_i = np.percentile(image, 0 + percentiles)
    max_i = np.percentile(image, 100 - percentiles)
    # NumPy 1.x value-based casting (the reference's NumPy) keeps a float32 array float32 against Python / float64
    # scalars; NumPy 2 would promote, so the scalars are cast explicitly to state the same arithmetic.
    t = image.dtype.type if image.dtype.kind == "f" else np.float64
    image = (np.divide((image - t(min_i)), t(max_i) - t(min_i)) * t(max_o - min_o) + t(min_o)).copy()
    image[image > max_o] = max_o
    image[image < min_o] = min_o
    return image


def prep_subject(p1, f1, icv1, sl1, p2, icv2, sl2, nicg):
    """GT:667-716 for one subject; volumes are (X, Y, Z) arrays (sl1 / sl2 None when the file is missing)."""
    ip1, ip2, ii1, ii2 = data_prep(p1), data_prep(p2), data_prep(icv1), data_prep(icv2)
    brain_prob_1tp = np.multiply(ip1, ii1)                                  # GT:686
    brain_prob_2tp = np.multiply(ip2, ii2)                                  # GT:688
    brain_flair_1tp = np.multiply(data_prep(f1), ii1) if f1 is not None else None   # GT:687
    if sl1 is not None:                                                     # GT:690-696
        s = 1 - data_prep(sl1)
        brain_prob_1tp = np.multiply(brain_prob_1tp, s)
        if brain_flair_1tp is not None:
            brain_flair_1tp = np.multiply(brain_flair_1tp, s)
    if sl2 is not None:                                                     # GT:698-703
        brain_prob_2tp = np.multiply(brain_prob_2tp, 1 - data_prep(sl2))
    if brain_flair_1tp is not None:
        brain_flair_1tp = map_image_to_intensity_range(brain_flair_1tp, 0, 1, percentiles=0)   # GT:707
    brain_prob_1tp[brain_prob_1tp < 0] = 0                                  # GT:716-717
    brain_prob_2tp[brain_prob_2tp < 0] = 0
    if nicg == 2:                                                           # GT:719-723
        brain_prob_1tp = np.concatenate((brain_prob_1tp, brain_flair_1tp), axis=-1)
    return brain_prob_1tp, brain_prob_2tp


def split_and_shuffle(x, y, seed_shuffle=None):
    """GT:738-760: sklearn train_test_split(test_size=0.02, random_state=42), then np.random.shuffle of the training
    indices.  The split restated: ShuffleSplit draws RandomState(42).permutation(n); the first ceil(0.02 n) indices are
    the validation set, the following n - ceil(0.02 n) the training set."""
    n = x.shape[0]
    n_val = int(np.ceil(0.02 * n))
    perm = np.random.RandomState(42).permutation(n)
    val, train = perm[:n_val], perm[n_val:]
    xt, xv, yt, yv = x[train], x[val], y[train], y[val]
    if seed_shuffle is not None:
        idx = np.array(range(xt.shape[0]))
        np.random.RandomState(seed_shuffle).shuffle(idx)
        xt, yt = xt[idx], yt[idx]
    return xt, xv, yt, yv
