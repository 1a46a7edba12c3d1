"""Data step in front of the path: NIfTI volumes -> training slices resident in HBM.

Mirrors what DEP-GAN_PROB_IM_twoCritics_training_4fold.py ("GT") does between its file lists and the first
training batch (GT:613-760): read seven lists of volume paths, per subject load the volumes, extract the 2D slices,
mask by the intracranial volume and (when the file exists) the inverted stroke-lesion mask, map the FLAIR channel
to [0, 1], clamp the probability maps at 0, concatenate the channels, stack all subjects, split off 2 % for
validation (train_test_split, random_state 42) and shuffle the rest.

Here the arithmetic of a subject runs on the GPU (`depgan_data_prep_subject`, include/depgan.h) while a host
thread reads and decodes the next subject's files into pinned memory, so a training set is assembled at file-read
speed and never exists on the host as float arrays.  Results are bit-identical to the NumPy statements
(oracle/data_oracle.py restates them for the tests).
"""
from __future__ import annotations

import ctypes as C
import os
import queue
import threading
from collections import namedtuple

import numpy as np

from . import _lib, nifti

SubjectFiles = namedtuple("SubjectFiles", "prob_1tp flair_1tp icv_1tp sl_1tp prob_2tp icv_2tp sl_2tp")

# list-file stems, GT:613-660
_LISTS = (("prob_1tp", "wmh_prob_1tp"), ("flair_1tp", "flair_1tp"), ("icv_1tp", "icv_1tp"),
          ("sl_1tp", "sl_cleaned_1tp"), ("prob_2tp", "wmh_prob_2tp"), ("icv_2tp", "icv_2tp"),
          ("sl_2tp", "sl_cleaned_2tp"))


def read_list(path):
    """GT:614-618: one path per line, trailing newline stripped (nothing else)."""
    with open(path, "r") as f:
        return [line.strip("\n") for line in f]


def training_file_lists(config_dir, fold):
    """The seven lists of GT:613-660 as one SubjectFiles per line index."""
    cols = {k: read_list(os.path.join(config_dir, "%s_fold%s.txt" % (stem, fold))) for k, stem in _LISTS}
    n = len(cols["prob_1tp"])
    for k, v in cols.items():
        if len(v) < n:
            raise ValueError("list %s has %d entries, wmh_prob_1tp has %d" % (k, len(v), n))
    return [SubjectFiles(*[cols[k][i] for k in SubjectFiles._fields]) for i in range(n)]


def _file_order_f32(vol):
    """(X, Y, Z) volume of any dtype -> flat float32 in file order (x fastest); the cast is data_prep's (GT:113)."""
    a = np.asarray(vol)
    if a.ndim != 3:
        raise ValueError("expected a 3-D volume, got shape %s" % (a.shape,))
    a = np.asfortranarray(a, dtype=np.float32)
    if not a.flags.writeable:           # a float32 file decodes to a read-only view of the file buffer
        a = a.copy(order="F")
    return a.reshape(-1, order="F")


def prep_subject(p1, f1, icv1, sl1, p2, icv2, sl2, nicg=2, device=None, stream=None):
    """One subject on the GPU.  Volumes: (X, Y, Z) NumPy arrays of any dtype or flat float32 CUDA tensors already in
    file order together with `shape=` ... (see prep_subject_flat); sl1 / sl2 / f1 may be None.
    Returns (x (Z, X, Y, nicg), y2 (Z, X, Y, 1)) as float32 CUDA tensors."""
    import torch
    dev = torch.device(device if device is not None else "cuda:0")
    shape = tuple(np.shape(p1))
    vols = []
    for v in (p1, f1, icv1, sl1, p2, icv2, sl2):
        if v is None:
            vols.append(None)
            continue
        if tuple(np.shape(v)) != shape:
            raise ValueError("volume shapes differ: %s vs %s" % (np.shape(v), shape))
        vols.append(torch.from_numpy(_file_order_f32(v)).to(dev, non_blocking=True))
    return prep_subject_flat(vols, shape, nicg, dev, stream)


def prep_subject_flat(vols, shape, nicg, dev, stream=None):
    """vols: [p1, f1, icv1, sl1, p2, icv2, sl2] flat float32 CUDA tensors in file order (None where absent)."""
    import torch
    lib = _lib.load()
    X, Y, Z = (int(s) for s in shape)
    if nicg == 2 and vols[1] is None:
        raise ValueError("nicg = 2 needs the FLAIR volume")
    for v in vols:
        if v is not None and (v.dtype != torch.float32 or v.numel() != X * Y * Z or not v.is_cuda):
            raise ValueError("volumes must be float32 CUDA tensors of X*Y*Z elements")
    x = torch.empty((Z, X, Y, nicg), dtype=torch.float32, device=dev)
    y2 = torch.empty((Z, X, Y, 1), dtype=torch.float32, device=dev)
    scratch = torch.empty(int(lib.depgan_data_prep_scratch_floats(X, Y, Z)), dtype=torch.float32, device=dev)
    st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream

    def p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    _lib.check(lib.depgan_data_prep_subject(p(vols[0]), p(vols[1]), p(vols[2]), p(vols[3]), p(vols[4]), p(vols[5]),
                                            p(vols[6]), X, Y, Z, int(nicg), p(x), p(y2), p(scratch),
                                            C.c_void_p(st)), "depgan_data_prep_subject")
    return x, y2


def _read_subject(files, nicg, pin):
    """Host side of one subject: decode the NIfTI files into flat float32 (pinned) tensors.  A missing stroke-lesion
    file means "no mask" (GT:690, 698: os.path.isfile)."""
    import torch
    out, shape = [], None
    for key in SubjectFiles._fields:
        path = getattr(files, key)
        optional = key.startswith("sl_")
        if (key == "flair_1tp" and nicg == 1) or (optional and not os.path.isfile(path)):
            out.append(None)
            continue
        vol = nifti.load(path).image
        if shape is None:
            shape = vol.shape
        elif vol.shape != shape:
            raise ValueError("%s: shape %s differs from %s" % (path, vol.shape, shape))
        t = torch.from_numpy(_file_order_f32(vol))
        out.append(t.pin_memory() if pin else t)
    return out, shape


def load_training_set(subjects, nicg=2, device=None, prefetch=2, progress=None):
    """GT:663-733: every subject whose wmh_prob_1tp file exists, stacked along the slice axis.
    subjects: list of SubjectFiles (training_file_lists).  Returns (x (N, X, Y, nicg), y2 (N, X, Y, 1)) on `device`.
    A reader thread keeps `prefetch` decoded subjects ahead of the GPU."""
    import torch
    dev = torch.device(device if device is not None else "cuda:0")
    todo = [s for s in subjects if os.path.isfile(s.prob_1tp)]
    q = queue.Queue(maxsize=max(1, int(prefetch)))

    def reader():
        try:
            for s in todo:
                q.put((s, _read_subject(s, nicg, pin=True)))
            q.put(None)
        except BaseException as e:      # surfaced on the consumer side
            q.put(e)

    th = threading.Thread(target=reader, daemon=True)
    th.start()
    xs, ys = [], []
    while True:
        item = q.get()
        if item is None:
            break
        if isinstance(item, BaseException):
            raise item
        s, (host, shape) = item
        vols = [None if t is None else t.to(dev, non_blocking=True) for t in host]
        x, y2 = prep_subject_flat(vols, shape, nicg, dev)
        xs.append(x)
        ys.append(y2)
        if progress is not None:
            progress(s, tuple(x.shape))
    th.join()
    if not xs:
        raise ValueError("no subject with an existing wmh_prob_1tp file")
    return torch.cat(xs, 0), torch.cat(ys, 0)


def split_and_shuffle(x, y2, rng=None):
    """GT:738-760 on device tensors: train_test_split(test_size=0.02, random_state=42) restated (ShuffleSplit: the
    first ceil(0.02 n) entries of RandomState(42).permutation(n) validate, the rest train), then the training
    indices shuffled with `rng` (np.random.RandomState or module np.random, as the reference's np.random.shuffle).
    Returns x_train, x_val, y2_train, y2_val."""
    import torch
    n = int(x.shape[0])
    n_val = int(np.ceil(0.02 * n))
    perm = np.random.RandomState(42).permutation(n)
    val, train = perm[:n_val], perm[n_val:]
    idx = np.array(range(train.shape[0]))
    (rng if rng is not None else np.random).shuffle(idx)
    train = train[idx]
    tv = torch.from_numpy(val.astype(np.int64)).to(x.device)
    tt = torch.from_numpy(train.astype(np.int64)).to(x.device)
    return x.index_select(0, tt), x.index_select(0, tv), y2.index_select(0, tt), y2.index_select(0, tv)


def data_prep_save(image_data):
    """GT:121-127: (Z, X, Y, 1) network output -> the orientation the reference saves to NIfTI."""
    a = np.squeeze(np.asarray(image_data))
    a = np.swapaxes(a, 0, 2)
    a = np.rot90(a)
    return a[::-1, ...]
