"""Writes a Keras-2.x-layout HDF5 weight file with the REAL HDF5 library (h5py) from an .npz of named tensors.

Run under an interpreter that has h5py -- in this image /opt/conda/bin/python3.9 (h5py 3.3.0, HDF5 1.10); the main
interpreter has none, which is why dep_gan_im_amd/h5lite.py exists:

    /opt/conda/bin/python3.9 tests/golden/make_keras_h5.py weights.npz out.h5 [model|weights]

Layout = what keras.engine.saving.save_weights_to_hdf5_group writes (Keras 2.1 - 2.3, the reference's range; GT:892
`netG.save(...)`, GE:383 `load_weights`): attributes `layer_names`, `backend`, `keras_version` on the weights group; one
group per layer with attribute `weight_names` = [b"<layer>/<weight>:0", ...] and one contiguous float32 dataset per
weight under that (nested) name; `model.save` files keep the weights group under "model_weights", `save_weights` files at
the root.  Layers without weights (Activation, Add, MaxPooling2D ...) appear as empty groups, as in a real file.
keras / tensorflow are not installed anywhere in this image: the layout is restated here, the container is genuine HDF5.
tests/golden/keras_layout_small.h5 was made by this script from keras_layout_small.npz (mode "model").
"""
import sys

import h5py
import numpy as np


def main():
    src, dst = sys.argv[1], sys.argv[2]
    mode = sys.argv[3] if len(sys.argv) > 3 else "model"
    with np.load(src) as z:
        tensors = {k: z[k] for k in z.files}
    layers = []
    for k in tensors:                                   # "<layer>/<weight>" in creation order
        layer = k.split("/", 1)[0]
        if layer not in layers:
            layers.append(layer)
    with h5py.File(dst, "w") as f:
        if mode == "model":
            f.attrs["keras_version"] = b"2.2.4"
            f.attrs["backend"] = b"tensorflow"
            f.attrs["model_config"] = b'{"class_name": "Model", "config": {"name": "model_1"}}'
            g = f.create_group("model_weights")
        else:
            g = f
        names = []
        for i, layer in enumerate(layers):
            names.append(layer)
            if i % 3 == 2:
                names.append("activation_%d" % i)       # a weight-less layer between them, as in the real graph
        # h5py 2.x (the reference's era) stored a list of bytes as FIXED-length strings, h5py 3.x stores it as
        # variable-length ones: the layer list goes in the old way, the per-layer weight lists in the new one, so both
        # forms are in the fixture
        g.attrs["layer_names"] = np.array([n.encode("utf8") for n in names], dtype="S")
        g.attrs["backend"] = b"tensorflow"
        g.attrs["keras_version"] = b"2.2.4"
        for n in names:
            lg = g.create_group(n)
            ws = [k for k in tensors if k.split("/", 1)[0] == n]
            lg.attrs["weight_names"] = [(k + ":0").encode("utf8") for k in ws]
            for k in ws:
                val = np.asarray(tensors[k])
                d = lg.create_dataset(k + ":0", val.shape, dtype=val.dtype)
                if val.shape:
                    d[:] = val
                else:
                    d[()] = val
    print("wrote", dst, "with h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version)


if __name__ == "__main__":
    main()
