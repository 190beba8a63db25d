"""Writes tests/golden/hdf5/h5py_*.h5 with the REAL h5py (3.3.0 on HDF5 1.10.6: /opt/conda/bin/python3.9 of the build image), issuing the
calls keras/engine/saving.py issues (save_weights_to_hdf5_group / save_attributes_to_hdf5_group / _save_model), so the reader meets the
bytes h5py really leaves for a Keras weight file and for a full-model file.  Run:

    /opt/conda/bin/python3.9 tests/golden/make_h5py_fixtures.py

Contents follow the rule of make_hdf5_fixtures.c (val(seed, i)), so tests/test_hdf5_pinned.py recomputes what to expect.
"""
import json
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hdf5")


def val(seed, n):
    i = np.arange(n, dtype=np.uint64)
    h = ((i + np.uint64(977 * seed)) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    return ((h >> np.uint64(8)).astype(np.float64) / 16777216.0 - 0.5).astype(np.float32)


LAYERS = [("input_1", []), ("conv2d_1", [("conv2d_1/kernel:0", (3, 3, 1, 4)), ("conv2d_1/bias:0", (4,))]),
          ("batch_normalization_1", [("batch_normalization_1/gamma:0", (4,)), ("batch_normalization_1/beta:0", (4,)),
                                     ("batch_normalization_1/moving_mean:0", (4,)), ("batch_normalization_1/moving_variance:0", (4,))]),
          ("a_layer_with_a_rather_long_name_1", [("a_layer_with_a_rather_long_name_1/kernel:0", (5, 7))])]


def save_attributes(group, name, data):
    """Keras splits an attribute whose encoded size passes 64 KiB into name0, name1, ...; small ones go in whole"""
    group.attrs[name] = data


def save_weights_to_group(f, as_bytes=True):
    enc = (lambda s: s.encode("utf8")) if as_bytes else (lambda s: s)
    save_attributes(f, "layer_names", [enc(n) for n, _ in LAYERS])
    f.attrs["backend"] = enc("tensorflow")
    f.attrs["keras_version"] = enc("2.3.1" if as_bytes else "2.4.0")
    for l, (ln, ws) in enumerate(LAYERS):
        g = f.create_group(ln)
        save_attributes(g, "weight_names", [enc(n) for n, _ in ws])
        for k, (wn, shape) in enumerate(ws):
            v = val(10 * l + k, int(np.prod(shape))).reshape(shape)
            d = g.create_dataset(wn, v.shape, dtype=v.dtype)
            if not v.shape:
                d[()] = v
            else:
                d[:] = v


CONFIG = {"class_name": "Model", "config": {"name": "model_1", "layers": [{"name": "input_1", "class_name": "InputLayer"},
          {"name": "conv2d_1", "class_name": "Conv2D", "config": {"filters": 4, "kernel_size": [3, 3], "padding": "same"}}]}}

with h5py.File(os.path.join(OUT, "h5py_keras_weights.h5"), "w") as f:          # model.save_weights (T1:1079), Keras 2.3: byte strings
    save_weights_to_group(f)

with h5py.File(os.path.join(OUT, "h5py_keras_fullmodel.h5"), "w") as f:        # model.save / ModelCheckpoint (T1:1046-1047), tf.keras 2.4 + h5py 3: str
    f.attrs["keras_version"] = "2.4.0"
    f.attrs["backend"] = "tensorflow"
    f.attrs["model_config"] = json.dumps(CONFIG)
    save_weights_to_group(f.create_group("model_weights"), as_bytes=False)
    f.attrs["training_config"] = json.dumps({"loss": "bce_dice_loss", "optimizer_config": {"class_name": "Adam", "config": {"lr": 0.0005}}})
    og = f.create_group("optimizer_weights")
    names = ["Adam/iterations:0", "Adam/conv2d_1/kernel/m:0"]
    og.attrs["weight_names"] = names
    d = og.create_dataset(names[0], (), dtype="int64"); d[()] = 1234567890123
    v = val(77, 36).reshape(3, 3, 1, 4)
    d = og.create_dataset(names[1], v.shape, dtype=v.dtype); d[:] = v

# Keras chunks attributes above 64 KiB (HDF5's object-header limit): 2000 layer names of 40 bytes -> layer_names0 .. layer_names1
with h5py.File(os.path.join(OUT, "h5py_keras_chunked_attrs.h5"), "w") as f:
    names = [("layer_%04d" % i).ljust(40, "x").encode() for i in range(2000)]
    parts = np.array_split(names, 2)
    for i, p in enumerate(parts):
        f.attrs["layer_names%d" % i] = p
    f.attrs["backend"] = b"tensorflow"; f.attrs["keras_version"] = b"2.3.1"
    for n in names[:3] + names[-2:]:                                       # only a few layer groups exist: the reader test looks at the names
        f.create_group(n.decode()).attrs["weight_names"] = []
print("h5py", h5py.__version__, "HDF5", h5py.version.hdf5_version, "->", sorted(p for p in os.listdir(OUT) if p.startswith("h5py_")))
