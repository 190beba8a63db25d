"""Run by tests/test_hdf5_pinned.py under the image's Anaconda interpreter (/opt/conda/bin/python3.9: h5py 3.3.0 on HDF5 1.10.6) -- the system
python has no h5py.  Opens FILE with the real h5py the way keras/engine/saving.py loads weights (load_weights_from_hdf5_group: `layer_names`
-> group -> `weight_names` -> np.asarray(g[name])) and writes what it read to OUT.npz for the caller to compare.

    /opt/conda/bin/python3.9 tests/h5py_check.py FILE OUT.npz
"""
import sys

import h5py
import numpy as np


def s(x):
    return x.decode("utf8") if isinstance(x, bytes) else str(x)


def main(path, out):
    res = {}
    with h5py.File(path, "r") as f:
        for k in ("keras_version", "backend", "model_config"):
            if k in f.attrs:
                res["rootattr::" + k] = np.array(s(f.attrs[k]))
        g = f["model_weights"] if "layer_names" not in f.attrs and "model_weights" in f else f
        names = [s(n) for n in g.attrs["layer_names"]]
        res["layer_names"] = np.array(names)
        res["group::backend"] = np.array(s(g.attrs["backend"])); res["group::keras_version"] = np.array(s(g.attrs["keras_version"]))
        for ln in names:
            lg = g[ln]
            wn = [s(n) for n in lg.attrs["weight_names"]]
            res["weight_names::" + ln] = np.array(wn, dtype="U") if wn else np.zeros((0,), "U1")
            for n in wn:
                d = lg[n]
                assert d.chunks is None and d.compression is None
                res["w::" + ln + "::" + n] = np.asarray(d)
                res["dtype::" + ln + "::" + n] = np.array(d.dtype.str)
        if "optimizer_weights" in f:                                   # saving.py _deserialize_model: weight_names -> [group[n] for n in names] -> optimizer.set_weights
            og = f["optimizer_weights"]
            on = [s(n) for n in og.attrs["weight_names"]]
            res["optimizer_weight_names"] = np.array(on)
            for i, n in enumerate(on):
                res["ow::%d" % i] = np.asarray(og[n])
            res["training_config"] = np.array(s(f.attrs["training_config"]))
        n_obj = [0]
        f.visititems(lambda name, obj: n_obj.__setitem__(0, n_obj[0] + 1))
        res["n_objects"] = np.array(n_obj[0])
    np.savez(out, **res)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
