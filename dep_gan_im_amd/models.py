"""Keras-style model objects over the HIP engine.

Same constructor names and call surface as the reference script
(DEP-GAN_PROB_IM_twoCritics_training_4fold.py, "GT"):
    Gen_UNet2D(input_shape, noiseZ_shape=(32, 1), first_fm=32, nc_out=1)   GT:349
    Dis_C2D_FCN1(input_shape)                                              GT:316
returning objects with .summary() .predict() .trainable_weights .get_weights()
.set_weights() .save() .load_weights() (GT:514-521, 846-848, 892; GE:383).
Weights are keyed by the reference's Keras layer names ('conv2d_gen_0/kernel',
'bn_gen_0/gamma', 'dense_noise_2_mul_m1/kernel', 'deconv2d_de_gen_9/kernel', ...)
in Keras layouts, so an .h5 importer is a pure rename.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

from .engine import Engine


class WeightRef:
    """What model.trainable_weights enumerates (name + shape, like a tf.Variable)."""

    def __init__(self, name, shape):
        self.name, self.shape = name, tuple(shape)

    def __repr__(self):
        return "<Weight %s %s>" % (self.name, self.shape)


def _glorot_uniform(rng, shape, fan_in, fan_out):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


# keras.initializers.he_normal = VarianceScaling(scale=2, mode='fan_in', distribution='normal'): a normal truncated at
# two standard deviations.  From Keras 2.2.3 on the sampled stddev is divided by .87962566103423978 (the standard deviation
# of the truncated unit normal) so that the RESULT has std sqrt(2/fan_in); earlier 2.x releases omit the correction
# (version-sensitive, unverifiable here).  The reference needs Python 2 + TF 1.x, for which 2.2.4 was the
# common release, so the corrected form is used -- the same constant as oracle/depgan_oracle.py::_he_normal.
HE_NORMAL_TRUNC_STD = 0.87962566103423978


def _he_normal(rng, shape, fan_in):
    std = math.sqrt(2.0 / fan_in) / HE_NORMAL_TRUNC_STD
    v = rng.standard_normal(size=shape)
    bad = np.abs(v) > 2
    while bad.any():
        v[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(v) > 2
    return (v * std).astype(np.float32)


def _keras_init(name, shape, rng):
    """Keras default initialisers for the layers the reference builds."""
    if name.endswith("/bias") or name.endswith("/beta") or name.endswith("/moving_mean"):
        return np.zeros(shape, np.float32)
    if name.endswith("/gamma") or name.endswith("/moving_variance"):
        return np.ones(shape, np.float32)
    if name.startswith("dense_") or name.startswith("dis_9"):   # he_normal (GT:256, 263, 339, 342)
        fan_in = int(np.prod(shape[:-1]))
        return _he_normal(rng, shape, fan_in)
    if name.startswith("deconv2d_"):                             # (kh,kw,Cout,Cin)
        kh, kw, co, ci = shape
        return _glorot_uniform(rng, shape, kh * kw * co, kh * kw * ci)
    kh, kw, ci, co = shape                                       # Conv2D glorot_uniform
    return _glorot_uniform(rng, shape, kh * kw * ci, kh * kw * co)


class _Model:
    net = None        # engine net id: 'G', 'D_y2', 'D_dem'
    name = "model"

    def __init__(self, input_shape, seed=None):
        self.input_shape = tuple(input_shape)
        self._engine = None
        self._host = None          # OrderedDict while unbound
        self._seed = seed
        self._private_engine = None

    # -- engine binding --
    def _spec_engine(self, batch):
        raise NotImplementedError

    def _bind(self, engine, net):
        host = self._weights_dict()
        self._engine, self.net = engine, net
        self._host = None
        engine.set_weights(net, host)

    def _ensure_engine(self, batch=32):
        if self._engine is None:
            eng = self._spec_engine(batch)
            self._bind(eng, self.net)
            self._private_engine = eng
        return self._engine

    def _table(self):
        if self._engine is not None:
            return self._engine.param_table(self.net)
        return self._static_table()

    def _weights_dict(self):
        if self._engine is not None:
            return self._engine.get_weights(self.net)
        if self._host is None:
            rng = np.random.default_rng(self._seed)
            self._host = OrderedDict((n, _keras_init(n, s, rng)) for n, s, _ in self._static_table())
        return self._host

    # -- keras surface --
    @property
    def trainable_weights(self):
        return [WeightRef(n, s) for n, s, tr in self._named_table() if tr]

    def _named_table(self):
        if self._engine is not None:
            return [(n, s, tr) for n, s, _, tr in self._engine.param_table(self.net)]
        return self._static_table()

    def count_params(self):
        return int(sum(np.prod(s) for _, s, _ in self._named_table()))

    def get_weights(self):
        return list(self._weights_dict().values())

    def get_weights_dict(self):
        return OrderedDict(self._weights_dict())

    def set_weights(self, weights):
        if isinstance(weights, dict):
            new = weights
        else:
            names = [n for n, _, _ in self._named_table()]
            if len(weights) != len(names):
                raise ValueError("set_weights: expected %d arrays, got %d" % (len(names), len(weights)))
            new = OrderedDict(zip(names, weights))
        for (n, s, _) in self._named_table():
            if n in new and tuple(np.shape(new[n])) != tuple(s):
                raise ValueError("weight %s: expected shape %s, got %s" % (n, s, np.shape(new[n])))
        if self._engine is not None:
            self._engine.set_weights(self.net, new)
        else:
            cur = self._weights_dict()
            for n, v in new.items():
                if n not in cur:
                    raise ValueError("unknown weight name %s" % n)
                cur[n] = np.asarray(v, np.float32)

    def save_weights(self, path):
        np.savez(path, **self._weights_dict())

    save = save_weights   # netG.save(...) GT:892 (architecture is code here; the file holds the weights)

    def load_weights(self, path):
        """`.npz` written by save_weights, or a Keras 2.x HDF5 file (`model.save` / `save_weights`: GT:892, GE:383)
        through h5py when it is installed and through dep_gan_im_amd.h5lite otherwise -- the names and layouts here are
        the Keras ones, so that import is a lookup."""
        if str(path).lower().endswith((".h5", ".hdf5")):
            try:
                import h5py as h5
            except ImportError:
                from . import h5lite as h5      # pure-Python reader of the HDF5 subset Keras weight files use
            with h5.File(path, "r") as f:
                self.set_weights(weights_from_keras_h5(f, [n for n, _, _ in self._named_table()]))
            return
        with np.load(path) as f:
            self.set_weights({k: f[k] for k in f.files})

    def summary(self, print_fn=print):
        print_fn('Model: "%s"' % self.name)
        print_fn("%-44s %-22s %10s" % ("Weight (layer/name)", "Shape", "Param #"))
        print_fn("=" * 78)
        tot = tr_tot = 0
        for n, s, tr in self._named_table():
            k = int(np.prod(s))
            tot += k
            tr_tot += k if tr else 0
            print_fn("%-44s %-22s %10d" % (n, str(tuple(s)), k))
        print_fn("=" * 78)
        print_fn("Total params: %d\nTrainable params: %d\nNon-trainable params: %d" % (tot, tr_tot, tot - tr_tot))


def weights_from_keras_h5(f, names):
    """Maps a Keras 2.x HDF5 weight file onto this package's weight names ("<layer>/<weight>").

    f: an open h5py.File (anything indexable the same way).  `model.save` files keep the layers under the group
    "model_weights", `save_weights` files at the root; each layer group holds its tensors at
    "<layer>/<weight>:0" (a nested group named after the layer again).  Layers the file does not have, or extra ones,
    are an error: a silent partial load would pass every shape check and train from a half-initialised model."""
    g = f["model_weights"] if "model_weights" in f else f
    out, missing = OrderedDict(), []
    for n in names:
        layer, w = n.split("/", 1)
        ds = None
        if layer in g:
            lg = g[layer]
            for key in ("%s/%s:0" % (layer, w), "%s:0" % w, "%s/%s" % (layer, w), w):
                node, ok = lg, True
                for part in key.split("/"):
                    if hasattr(node, "keys") and part in node:
                        node = node[part]
                    else:
                        ok = False
                        break
                if ok and not hasattr(node, "keys"):
                    ds = node
                    break
        if ds is None:
            missing.append(n)
        else:
            out[n] = np.asarray(ds[()] if hasattr(ds, "shape") and not isinstance(ds, np.ndarray) else ds, np.float32)
    if missing:
        raise KeyError("Keras HDF5 file lacks %d of %d weights, e.g. %s" % (len(missing), len(names), missing[:3]))
    return out


def _gen_static_table(nicg, fm, nc_out):
    T = []

    def bn(n, c):
        T.extend([(n + "/gamma", (c,), True), (n + "/beta", (c,), True), (n + "/moving_mean", (c,), False),
                  (n + "/moving_variance", (c,), False)])

    def dense(n, fi, fo):
        T.extend([("dense_" + n + "/kernel", (fi, fo), True), ("dense_" + n + "/bias", (fo,), True)])
        bn("dense_bn_" + n, fo)

    dense("noise_1_add_f0", 1, fm)
    dense("noise_1_add_f1", fm, fm)
    for sfx, m in (("add_m3", 3), ("mul_m3", 3), ("add_m2", 2), ("mul_m2", 2), ("add_m1", 1), ("mul_m1", 1),
                   ("add", 4), ("mul", 4), ("add_p3", 3), ("mul_p3", 3), ("add_p2", 2), ("mul_p2", 2),
                   ("add_p1", 1), ("mul_p1", 1)):
        dense("noise_2_" + sfx, 32 * fm, fm * m)
    convs = [("gen_0", nicg, fm), ("gen_noise_m1", fm, fm), ("gen_1", fm, fm), ("gen_2", fm, 2 * fm),
             ("gen_noise_m2", 2 * fm, 2 * fm), ("gen_3", 2 * fm, 2 * fm), ("gen_4", 2 * fm, 3 * fm),
             ("gen_noise_m3", 3 * fm, 3 * fm), ("gen_5", 3 * fm, 3 * fm), ("gen_8", 3 * fm, 4 * fm),
             ("gen_noise_p4", 4 * fm, 4 * fm), ("gen_9", 4 * fm, 4 * fm), ("D:de_gen_9", 4 * fm, 4 * fm),
             ("gen_10", 7 * fm, 3 * fm), ("gen_noise_p3", 3 * fm, 3 * fm), ("gen_11", 3 * fm, 3 * fm),
             ("D:de_gen_11", 3 * fm, 3 * fm), ("gen_14", 5 * fm, 2 * fm), ("gen_noise_p2", 2 * fm, 2 * fm),
             ("gen_15", 2 * fm, 2 * fm), ("D:de_gen_15", 2 * fm, 2 * fm), ("gen_16", 3 * fm, fm),
             ("gen_noise_p1", fm, fm), ("gen_17", fm, fm)]
    for n, ci, co in convs:
        if n.startswith("D:"):
            n = n[2:]
            T.extend([("deconv2d_" + n + "/kernel", (2, 2, co, ci), True), ("deconv2d_" + n + "/bias", (co,), True)])
        else:
            T.extend([("conv2d_" + n + "/kernel", (3, 3, ci, co), True), ("conv2d_" + n + "/bias", (co,), True)])
        bn("bn_" + n, co)
    T.extend([("gen_segmentation/kernel", (1, 1, fm, nc_out), True), ("gen_segmentation/bias", (nc_out,), True)])
    return T


class History:
    """keras.callbacks.History: .history['loss'] / ['val_loss'] per epoch (UT:609-618)."""

    def __init__(self, history):
        self.history = history


class GeneratorModel(_Model):
    net = "G"
    name = "Gen_UNet2D"

    def __init__(self, input_shape, noiseZ_shape=(32, 1), first_fm=32, nc_out=1, seed=None):
        super().__init__(input_shape, seed)
        if tuple(noiseZ_shape) != (32, 1) or first_fm != 32:
            raise ValueError("the HIP path is built for noiseZ_shape=(32,1), first_fm=32 (GT:520)")
        if nc_out not in (1, 4):
            raise ValueError("nc_out must be 1 (DEP-GAN generator, GT:520) or 4 (DEP-UResNet, UT:583)")
        self.noiseZ_shape, self.first_fm, self.nc_out = tuple(noiseZ_shape), first_fm, nc_out
        if nc_out != 1:
            self.name = "DEP_UResNet"
        # Gen_UNet2D compiles the softmax variant itself: Adam(lr=1e-4), categorical cross-entropy (UT:427)
        self._lr = 1e-4
        self._drop_rng = np.random.RandomState(seed)

    def _static_table(self):
        return _gen_static_table(self.input_shape[2], self.first_fm, self.nc_out)

    def _spec_engine(self, batch):
        H, W, nicg = self.input_shape
        if self.nc_out == 1:
            return Engine(batch, H, W, nicg)
        return Engine(batch, H, W, nicg, lrG=self._lr, beta1=0.9, beta2=0.999, nc_out=self.nc_out)

    def predict(self, inputs, batch_size=32):
        """netG.predict([x, z])  (GT:848, 859; GE:621; UE:  my_network.predict)."""
        if not isinstance(inputs, (list, tuple)) or len(inputs) != 2:
            raise ValueError("Gen_UNet2D.predict expects [images, noise]")
        x, z = inputs
        eng = self._ensure_engine(min(batch_size, max(1, len(x))))
        return eng.g_forward(x, z).cpu().numpy()

    # ---- supervised surface of the softmax variant (DEP-UResNet, UT:427, 583-618) ----
    def _need_softmax(self, what):
        if self.nc_out == 1:
            raise RuntimeError("%s: the tanh generator is trained through the WGAN-GP closures "
                               "(trainers.build_trainers), not compiled with a loss" % what)

    def compile(self, optimizer="adam", loss="categorical_crossentropy", lr=None, **_):
        """model.compile(optimizer=Adam(lr=1e-4), loss='categorical_crossentropy')  (UT:427)."""
        self._need_softmax("compile")
        if loss != "categorical_crossentropy":
            raise ValueError("only loss='categorical_crossentropy' is built (UT:427)")
        if lr is None:
            lr = getattr(optimizer, "lr", None)
        if lr is not None:
            if self._engine is not None and float(lr) != self._lr:
                raise RuntimeError("compile(lr=...) must come before the first fit / train_on_batch / predict")
            self._lr = float(lr)
        return self

    def _next_drop_seed(self):
        return int(self._drop_rng.randint(1, 2 ** 31 - 1))

    def train_on_batch(self, inputs, labels, drop_seed=None):
        """One learning-phase-1 Adam step; returns the batch loss."""
        self._need_softmax("train_on_batch")
        x, z = inputs
        eng = self._ensure_engine(len(x))
        return eng.uresnet(x, z, labels, "step", self._next_drop_seed() if drop_seed is None else drop_seed)

    def test_on_batch(self, inputs, labels):
        self._need_softmax("test_on_batch")
        x, z = inputs
        return self._ensure_engine(len(x)).uresnet(x, z, labels, "eval")

    def evaluate(self, inputs, labels, batch_size=32, verbose=0):
        """Sample-weighted mean of the phase-0 loss over batches (keras Model.evaluate)."""
        self._need_softmax("evaluate")
        x, z = inputs
        eng = self._ensure_engine(min(batch_size, len(x)))
        bs = min(batch_size, eng.batch)
        tot = 0.0
        for i in range(0, len(x), bs):
            m = min(bs, len(x) - i)
            tot += m * eng.uresnet(x[i:i + m], z[i:i + m], labels[i:i + m], "eval")
        return tot / len(x)

    def fit(self, inputs, labels, epochs=1, batch_size=32, shuffle=True, validation_data=None, verbose=1,
            print_fn=print):
        """my_network.fit([flair, noise], onehot, epochs=1, batch_size=nb_samples, shuffle=..., validation_data=...)
        (UT:602-606).  Batches in index order (after an optional np.random shuffle, as keras does), a short last
        batch, per-epoch loss = sample-weighted mean of the batch losses; returns an object with .history."""
        self._need_softmax("fit")
        x, z = inputs
        n = len(x)
        if len(z) != n or len(labels) != n:
            raise ValueError("fit: images, noise and labels must have the same length")
        eng = self._ensure_engine(min(batch_size, n))
        bs = min(batch_size, eng.batch)
        hist = {"loss": []}
        if validation_data is not None:
            hist["val_loss"] = []
        for ep in range(epochs):
            order = np.arange(n)
            if shuffle:
                np.random.shuffle(order)
            tot = 0.0
            for i in range(0, n, bs):
                idx = order[i:i + bs]
                tot += len(idx) * eng.uresnet(x[idx], z[idx], labels[idx], "step", self._next_drop_seed())
            hist["loss"].append(tot / n)
            msg = "Epoch %d/%d - loss: %.4f" % (ep + 1, epochs, hist["loss"][-1])
            if validation_data is not None:
                (vx, vz), vy = validation_data
                hist["val_loss"].append(self.evaluate([vx, vz], vy, batch_size=bs))
                msg += " - val_loss: %.4f" % hist["val_loss"][-1]
            if verbose:
                print_fn(msg)
        return History(hist)

    def get_config(self):
        return {"name": self.name, "input_shape": self.input_shape, "noiseZ_shape": self.noiseZ_shape,
                "first_fm": self.first_fm, "nc_out": self.nc_out}

    def to_json(self):
        import json
        return json.dumps(self.get_config())


_DIS = [("dis_0a", 5, 1, 16), ("dis_0b", 5, 16, 16), ("dis_1a", 5, 16, 32), ("dis_1b", 5, 32, 32),
        ("dis_2", 3, 32, 64), ("dis_3", 3, 64, 64), ("dis_4", 3, 64, 128), ("dis_5", 3, 128, 128),
        ("dis_6", 3, 128, 256), ("dis_7", 3, 256, 256), ("dis_8", 3, 256, 256)]


class CriticModel(_Model):
    net = "D_y2"
    name = "Dis_C2D_FCN1"

    def _static_table(self):
        H, W, _ = self.input_shape
        T = []
        for n, k, ci, co in _DIS:
            T.extend([("conv2d_" + n + "/kernel", (k, k, ci, co), True), ("conv2d_" + n + "/bias", (co,), True)])
        T.extend([("dis_9/kernel", (1, 1, 256, 1), True), ("dis_9/bias", (1,), True),
                  ("dense_1/kernel", ((H // 16) * (W // 16), 1), True), ("dense_1/bias", (1,), True)])
        return T

    def _spec_engine(self, batch):
        H, W, _ = self.input_shape
        return Engine(batch, H, W, 1)

    def predict(self, x, batch_size=32):
        """netD.predict(images)  (GT:846-848)."""
        eng = self._ensure_engine(min(batch_size, max(1, len(x))))
        return eng.d_forward(self.net, x).cpu().numpy()


def Gen_UNet2D(input_shape, noiseZ_shape=(32, 1), first_fm=32, nc_out=1, seed=None):
    return GeneratorModel(input_shape, noiseZ_shape, first_fm, nc_out, seed)


def Dis_C2D_FCN1(input_shape, seed=None):
    if tuple(input_shape)[2] != 1:
        raise ValueError("Dis_C2D_FCN1 takes single-channel images (GT:513)")
    return CriticModel(input_shape, seed)
