"""Weight table of the U-Net in Keras order/layouts, initialisers, and (de)serialisation.

Mirrors what the reference gets from Keras for the model built at
task1_preprocessing_plus_unet_with_comments.py:853-916:  model.get_weights() order,
``save_weights`` / ``load_weights`` (T1:1073, 1079) and ``to_json`` (T1:1091-1093).
The on-disk container is the Keras HDF5 weight layout (written / read by hdf5_min.py: h5py is not
in this image); ``.npz`` archives keyed by the Keras weight names are the alternative for paths
ending in ``.npz``.
"""
from __future__ import annotations

import json
import math
from collections import OrderedDict

import numpy as np

ENC = (32, 64, 128, 256)
DEC = (256, 128, 64, 32)


# U-Net++ (task1_unet_plus_plus.py:858-950): (node, width, ConvT source, skip tensors concatenated after the up-sampled one)
PP_NODES = (("x1_2", 32, "c2", ("c1",)), ("x2_2", 64, "c3", ("c2",)), ("x1_3", 32, "x2_2", ("c1", "x1_2")),
            ("x3_2", 128, "c4", ("c3",)), ("x2_3", 64, "x3_2", ("c2", "x2_2")), ("x1_4", 32, "x2_3", ("c1", "x1_2", "x1_3")))
PP_WIDTH = {"c1": 32, "c2": 64, "c3": 128, "c4": 256, "x1_2": 32, "x2_2": 64, "x1_3": 32, "x3_2": 128, "x2_3": 64, "x1_4": 32}
PP_ORDER = ("e1", "e2", "x1_2", "e3", "x2_2", "x1_3", "e4", "x3_2", "x2_3", "x1_4")          # Keras creation order UPP:876-924


def _pp_layer_table(in_ch):
    nodes = {n[0]: n for n in PP_NODES}
    t = []
    for it in PP_ORDER:
        if it[0] == "e":
            k = int(it[1]); c = ENC[k - 1]; cin = in_ch if k == 1 else ENC[k - 2]
            t += [(f"c{k}a", "conv3", cin, c), (f"c{k}b", "conv3", c, c), (f"bn{k}", "bn", c, c)]
        else:
            _, c, src, skips = nodes[it]
            ctot = c + sum(PP_WIDTH[s] for s in skips)
            t += [(f"u{it[1:]}", "convT", PP_WIDTH[src], c), (f"{it}a", "conv3", ctot, c), (f"{it}abn", "bn", c, c),
                  (f"{it}b", "conv3", c, c), (f"{it}bbn", "bn", c, c)]
    t.append(("out", "conv1", 32, 1))
    return t


CLS_C = (16, 32, 64)            # slice classifier, task2_covid19_classifcation.py:747-776
CLS_HIDDEN = 32
CLS_HW = (224, 224)             # the reference's input size (new_dim = 224, T2:489); `hw` overrides it (fc1 fan-in = (H/8)(W/8)64)


def _cls_layer_table(in_ch, hw=None):
    h, w = hw or CLS_HW
    t, cp = [], in_ch
    for k, c in enumerate(CLS_C, 1):
        t += [(f"c{k}a", "conv3", cp, c), (f"bn{k}a", "bn", c, c), (f"c{k}b", "conv3", c, c), (f"bn{k}b", "bn", c, c)]
        cp = c
    t += [("fc1", "dense", (h // 8) * (w // 8) * CLS_C[-1], CLS_HIDDEN), ("fc2", "dense", CLS_HIDDEN, 1)]
    return t


def layer_table(in_ch: int = 1, arch: str = "unet", hw=None):
    """[(name, kind, cin, cout)], kind in conv3|convT|bn|conv1|dense -- Keras creation order."""
    if arch == "unetpp":
        return _pp_layer_table(in_ch)
    if arch == "classifier":
        return _cls_layer_table(in_ch, hw)
    t, cp = [], in_ch
    for k, c in enumerate(ENC, 1):
        t += [(f"c{k}a", "conv3", cp, c), (f"c{k}b", "conv3", c, c), (f"bn{k}", "bn", c, c)]
        cp = c
    t += [("c5a", "conv3", 256, 512), ("c5b", "conv3", 512, 512)]
    cp = 512
    for k, c in zip((6, 7, 8, 9), DEC):
        t += [(f"u{k}", "convT", cp, c), (f"bn{k}", "bn", 2 * c, 2 * c), (f"c{k}a", "conv3", 2 * c, c), (f"c{k}b", "conv3", c, c)]
        cp = c
    t.append(("out", "conv1", 32, 1))
    return t


def weight_shapes(in_ch: int = 1, arch: str = "unet", hw=None):
    d = OrderedDict()
    for name, kind, cin, cout in layer_table(in_ch, arch, hw):
        if kind == "conv3":
            d[f"{name}/kernel"] = (3, 3, cin, cout); d[f"{name}/bias"] = (cout,)
        elif kind == "conv1":
            d[f"{name}/kernel"] = (1, 1, cin, cout); d[f"{name}/bias"] = (cout,)
        elif kind == "convT":
            d[f"{name}/kernel"] = (2, 2, cout, cin); d[f"{name}/bias"] = (cout,)
        elif kind == "dense":
            d[f"{name}/kernel"] = (cin, cout); d[f"{name}/bias"] = (cout,)
        else:
            for p in ("gamma", "beta", "mean", "var"):
                d[f"{name}/{p}"] = (cout,)
    return d


def keras_names(in_ch: int = 1, arch: str = "unet", hw=None):
    """our name -> Keras auto-name (conv2d_N/kernel:0 ...), counting per layer type in creation order."""
    out, nc, nt, nb, nd = OrderedDict(), 0, 0, 0, 0
    for name, kind, _, _ in layer_table(in_ch, arch, hw):
        if kind in ("conv3", "conv1"):
            nc += 1; base = f"conv2d_{nc}"
            out[f"{name}/kernel"] = f"{base}/kernel:0"; out[f"{name}/bias"] = f"{base}/bias:0"
        elif kind == "convT":
            nt += 1; base = f"conv2d_transpose_{nt}"
            out[f"{name}/kernel"] = f"{base}/kernel:0"; out[f"{name}/bias"] = f"{base}/bias:0"
        elif kind == "dense":
            nd += 1; base = f"dense_{nd}"
            out[f"{name}/kernel"] = f"{base}/kernel:0"; out[f"{name}/bias"] = f"{base}/bias:0"
        else:
            nb += 1; base = f"batch_normalization_{nb}"
            for p, kp in (("gamma", "gamma"), ("beta", "beta"), ("mean", "moving_mean"), ("var", "moving_variance")):
                out[f"{name}/{p}"] = f"{base}/{kp}:0"
    return out


def count_params(in_ch: int = 1, arch: str = "unet", hw=None):
    sh = weight_shapes(in_ch, arch, hw)
    total = sum(int(np.prod(s)) for s in sh.values())
    non_train = sum(int(np.prod(s)) for k, s in sh.items() if k.endswith("/mean") or k.endswith("/var"))
    return total, total - non_train


def _truncated_normal(rng, shape):
    k = rng.standard_normal(shape)
    bad = np.abs(k) > 2.0
    while bad.any():
        k[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(k) > 2.0
    return k


def init_weights(seed: int = 0, in_ch: int = 1, arch: str = "unet", hw=None):
    """he_normal for the 3x3 convs (T1:859...), Keras-default glorot_uniform for ConvT / the U-Net head (he_normal for the
    U-Net++ head, UPP:946), zero biases, BN gamma 1 / beta 0 / moving mean 0 / moving var 1."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, kind, cin, cout in layer_table(in_ch, arch, hw):
        if kind == "conv3":
            std = math.sqrt(2.0 / (9 * cin)) / 0.87962566103423978
            w[f"{name}/kernel"] = (_truncated_normal(rng, (3, 3, cin, cout)) * std).astype(np.float32)
            w[f"{name}/bias"] = np.zeros(cout, np.float32)
        elif kind == "conv1":
            if arch == "unetpp":
                std = math.sqrt(2.0 / cin) / 0.87962566103423978
                w[f"{name}/kernel"] = (_truncated_normal(rng, (1, 1, cin, cout)) * std).astype(np.float32)
            else:
                lim = math.sqrt(6.0 / (cin + cout))
                w[f"{name}/kernel"] = rng.uniform(-lim, lim, (1, 1, cin, cout)).astype(np.float32)
            w[f"{name}/bias"] = np.zeros(cout, np.float32)
        elif kind == "dense":                        # Keras default glorot_uniform (T2:773, 776)
            lim = math.sqrt(6.0 / (cin + cout))
            w[f"{name}/kernel"] = rng.uniform(-lim, lim, (cin, cout)).astype(np.float32)
            w[f"{name}/bias"] = np.zeros(cout, np.float32)
        elif kind == "convT":
            lim = math.sqrt(6.0 / (4 * cout + 4 * cin))
            w[f"{name}/kernel"] = rng.uniform(-lim, lim, (2, 2, cout, cin)).astype(np.float32)
            w[f"{name}/bias"] = np.zeros(cout, np.float32)
        else:
            w[f"{name}/gamma"] = np.ones(cout, np.float32); w[f"{name}/beta"] = np.zeros(cout, np.float32)
            w[f"{name}/mean"] = np.zeros(cout, np.float32); w[f"{name}/var"] = np.ones(cout, np.float32)
    return w


_KERAS_WEIGHT_ORDER = {"kernel": 0, "bias": 1, "gamma": 0, "beta": 1, "mean": 2, "var": 3}


def _layer_weight_lists(weights, in_ch, arch, hw):
    """[(keras layer name, [(keras weight name, array)])] in model.layers order, weight-less layers included (keras_graph)."""
    from . import keras_graph as KG
    kn = keras_names(in_ch, arch, hw)
    per = {}
    for k, v in weights.items():
        per.setdefault(k.split("/")[0], []).append((kn[k], np.asarray(v, np.float32), _KERAS_WEIGHT_ORDER[k.split("/")[1]]))
    out = []
    for l in KG.keras_layers(in_ch, arch, hw or ((224, 224) if arch != "classifier" else CLS_HW)):
        ws = sorted(per.get(l["engine"], []), key=lambda t: t[2]) if l["engine"] else []
        out.append((l["name"], [(n, a) for n, a, _ in ws]))
    return out


ADAM_DEFAULTS = {"beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-7, "decay": 0.0, "amsgrad": False}          # Adam(lr=0.0005), T1:1053


def trainable_names(in_ch: int = 1, arch: str = "unet", hw=None):
    """engine tensor names in Keras' model.trainable_weights order (layer order; kernel, bias / gamma, beta)"""
    order = []
    for _, ws in _layer_weight_lists({k: np.zeros(0, np.float32) for k in weight_shapes(in_ch, arch, hw)}, in_ch, arch, hw):
        order += [n for n, _ in ws if not (n.endswith("moving_mean:0") or n.endswith("moving_variance:0"))]
    back = {v: k for k, v in keras_names(in_ch, arch, hw).items()}
    return [back[n] for n in order]


def _optimizer_lists(opt, in_ch, arch, hw):
    """Keras 2.3 Adam: optimizer.weights = [iterations] + ms + vs + vhats (vhat_i = zeros(1) without amsgrad), created under the name scopes
    Adam/ (iterations) and training/Adam/ (keras/optimizers.py Adam.get_updates)"""
    names = trainable_names(in_ch, arch, hw)
    out = [("Adam/iterations:0", np.asarray(int(opt["step"]), np.int64))]
    out += [(f"training/Adam/m_{i}:0", np.asarray(opt["m"][k], np.float32)) for i, k in enumerate(names)]
    out += [(f"training/Adam/v_{i}:0", np.asarray(opt["v"][k], np.float32)) for i, k in enumerate(names)]
    out += [(f"training/Adam/vhat_{i}:0", np.zeros(1, np.float32)) for i in range(len(names))]
    cfg = {"optimizer_config": {"class_name": "Adam", "config": dict(ADAM_DEFAULTS, learning_rate=float(opt["lr"]))},
           "loss": opt.get("loss", "bce_dice_loss"), "metrics": list(opt.get("metrics", ["dice_coeff"])), "weighted_metrics": None, "sample_weight_mode": None,
           "loss_weights": None}
    return out, json.dumps(cfg)


def load_optimizer(path: str, in_ch: int = 1, arch: str = "unet", hw=None):
    """The optimizer state of a full-model file (model.save / ModelCheckpoint, T1:1046-1047) -> {"step", "lr", "m": {...}, "v": {...}, "loss", "metrics"} keyed by
    engine tensor names, or None when the file carries none.  Keras restores by position (optimizer.set_weights); so does this."""
    from . import hdf5_min as H5
    ows, tc = H5.load_keras_optimizer(path)
    if not ows:
        return None
    names = trainable_names(in_ch, arch, hw)
    sh = weight_shapes(in_ch, arch, hw)
    P = len(names)
    if len(ows) not in (1 + 2 * P, 1 + 3 * P):
        raise ValueError(f"{path}: {len(ows)} optimizer weights for {P} trainable tensors (expected iterations + m + v [+ vhat])")
    vals = [a for _, a in ows]
    m = OrderedDict(); v = OrderedDict()
    for i, k in enumerate(names):
        for dst, a in ((m, vals[1 + i]), (v, vals[1 + P + i])):
            if tuple(a.shape) != tuple(sh[k]):
                raise ValueError(f"{path}: optimizer slot of {k} has shape {a.shape}, the graph wants {sh[k]}")
            dst[k] = np.asarray(a, np.float32)
    cfg = json.loads(tc) if tc else {}
    oc = cfg.get("optimizer_config", {}).get("config", {})
    return {"step": int(np.asarray(vals[0]).reshape(-1)[0]), "lr": float(oc.get("learning_rate", oc.get("lr", 0.0005))), "m": m, "v": v,
            "loss": cfg.get("loss", "bce_dice_loss"), "metrics": cfg.get("metrics", ["dice_coeff"])}


def save_weights(path: str, weights, in_ch: int = 1, arch: str = "unet", hw=None, full_model: bool = False, optimizer=None):
    """model.save_weights(path) (T1:1079) / the file ModelCheckpoint writes (T1:1046-1047, `full_model=True`): a Keras HDF5 weight file --
    root (or `model_weights/`) attributes `layer_names`, `backend`, `keras_version`; one group per layer with `weight_names` and the
    datasets `<layer>/<layer>/kernel:0` ... (hdf5_min.py).  optimizer ({"step", "lr", "m", "v"}: engine.get_optimizer_state) adds the
    `optimizer_weights/` group and `training_config` of a compiled model's full-model file.  A path ending in `.npz` writes a NumPy archive
    keyed by the same Keras weight names instead."""
    if str(path).endswith(".npz"):
        kn = keras_names(in_ch, arch, hw)
        with open(path, "wb") as f:
            np.savez(f, **{kn[k]: np.asarray(v) for k, v in weights.items()})
        return
    from . import hdf5_min as H5
    cfg = None
    if full_model:
        from . import keras_graph as KG
        h, w = hw or ((224, 224) if arch != "classifier" else CLS_HW)
        cfg = KG.to_json(h, w, in_ch, arch)
    ow = tc = None
    if full_model and optimizer is not None:
        ow, tc = _optimizer_lists(optimizer, in_ch, arch, hw)
    H5.save_keras_weights(path, _layer_weight_lists(weights, in_ch, arch, hw), full_model=full_model, model_config=cfg, optimizer_weights=ow, training_config=tc)


def load_weights(path: str, in_ch: int = 1, arch: str = "unet", hw=None):
    """model.load_weights(path) (T1:1073): a Keras HDF5 file in either layout (weights at the root, or a full-model file with
    `model_weights/`), detected by its signature; layers are matched by their Keras names, or -- like Keras' topological loading -- in
    `layer_names` order when the names carry another session's counters (`conv2d_20` ...).  `.npz` archives (keyed by the Keras
    weight names) are read too.  Shapes are checked against the graph."""
    kn = keras_names(in_ch, arch, hw)
    sh = weight_shapes(in_ch, arch, hw)
    with open(path, "rb") as f:
        head = f.read(8)
    if head[:4] == b"PK\x03\x04":
        z = np.load(path)
        flat = {k: z[k] for k in z.files}
    else:
        from . import hdf5_min as H5
        if head != H5.SIGNATURE:
            raise ValueError(f"{path}: neither an HDF5 file (Keras weights) nor an .npz archive")
        layers, _ = H5.load_keras_weights(path)
        want_layers = []
        for k in sh:
            ln = kn[k].split("/")[0]
            if ln not in want_layers:
                want_layers.append(ln)
        flat = {}
        if all(ln in layers for ln in want_layers):
            for ln in want_layers:
                flat.update(layers[ln])
        else:
            have = [(ln, ws) for ln, ws in layers.items() if ws]
            if len(have) != len(want_layers):
                raise ValueError(f"{path}: {len(have)} layers with weights, the {arch} graph has {len(want_layers)}")
            from . import keras_graph as KG
            order = [l["name"] for l in KG.keras_layers(in_ch, arch, hw or ((224, 224) if arch != "classifier" else CLS_HW)) if l["name"] in want_layers]
            for mine, (theirs, ws) in zip(order, have):
                names = [v for k, v in kn.items() if v.split("/")[0] == mine]
                if len(names) != len(ws):
                    raise ValueError(f"{path}: layer {theirs} has {len(ws)} weights, {mine} needs {len(names)}")
                # Keras weight order inside a layer: kernel, bias / gamma, beta, moving_mean, moving_variance
                rank = {"kernel:0": 0, "bias:0": 1, "gamma:0": 0, "beta:0": 1, "moving_mean:0": 2, "moving_variance:0": 3}
                for nm, a in zip(sorted(names, key=lambda n: rank[n.split("/")[1]]), ws.values()):
                    flat[nm] = a
    out = OrderedDict()
    for k, shape in sh.items():
        if kn[k] not in flat:
            raise ValueError(f"{path}: weight {kn[k]} is missing")
        a = np.asarray(flat[kn[k]])
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{path}: {kn[k]} has shape {a.shape}, expected {shape}")
        out[k] = a.astype(np.float32)
    return out


def to_json(h: int, w: int, in_ch: int = 1, arch: str = "unet") -> str:
    """model.to_json() (T1:1091-1093): the Keras 2.3 architecture JSON of the graph (keras_graph.py)."""
    from . import keras_graph as KG
    return KG.to_json(h, w, in_ch, arch)
