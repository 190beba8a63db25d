"""The three models as Keras sees them: every layer (weight-less ones included) with its Keras auto-name, class, constructor config
and inbound connections, in `model.layers` order -- what `model.to_json()` (T1:1091-1093, T2:867-869) serialises and what orders the
`layer_names` attribute of a Keras weight file (keras/engine/saving.py).

Auto-names count per class in CREATION order (`conv2d_1` ... in the order the reference's script calls the constructors: T1:853-913,
UPP:876-947, T2:747-776).  `model.layers` of a functional Model is sorted by decreasing graph depth, ties broken by the order in which
the backwards depth-first traversal from the output first meets a layer (keras/engine/network.py `_map_graph_network`); a Sequential
keeps insertion order.  Restated from Keras 2.3.1 -- third-party, un-vendored: "parity unpinned" (the engine's own loader matches by
layer name first, so the order only matters for files whose names were shifted by earlier models in the same Keras session)."""
from __future__ import annotations

import json
from collections import OrderedDict

from .weights import CLS_C, CLS_HIDDEN, DEC, ENC, PP_NODES, PP_WIDTH

KERAS_VERSION, BACKEND = "2.3.1", "tensorflow"
_CLASS_PREFIX = {"InputLayer": "input", "Conv2D": "conv2d", "Conv2DTranspose": "conv2d_transpose", "BatchNormalization": "batch_normalization",
                 "MaxPooling2D": "max_pooling2d", "Dropout": "dropout", "Concatenate": "concatenate", "Flatten": "flatten", "Dense": "dense"}
_ZEROS = {"class_name": "Zeros", "config": {}}
_ONES = {"class_name": "Ones", "config": {}}
_HE_NORMAL = {"class_name": "VarianceScaling", "config": {"scale": 2.0, "mode": "fan_in", "distribution": "normal", "seed": None}}
_GLOROT = {"class_name": "VarianceScaling", "config": {"scale": 1.0, "mode": "fan_avg", "distribution": "uniform", "seed": None}}


class _Builder:
    def __init__(self):
        self.layers = []                      # creation order: dict(name, class_name, config, inbound [names], engine)
        self.counts = {}

    def add(self, cls, inbound, engine=None, **config):
        self.counts[cls] = self.counts.get(cls, 0) + 1
        name = f"{_CLASS_PREFIX[cls]}_{self.counts[cls]}"
        cfg = OrderedDict(name=name)
        if cls != "InputLayer":
            cfg["trainable"] = True
        cfg.update(config)
        self.layers.append({"name": name, "class_name": cls, "config": cfg, "inbound": list(inbound), "engine": engine})
        return name

    def conv(self, x, filters, k, act, init, engine, padding="same"):
        return self.add("Conv2D", [x], engine, dtype="float32", filters=filters, kernel_size=[k, k], strides=[1, 1], padding=padding, data_format="channels_last",
                        dilation_rate=[1, 1], activation=act, use_bias=True, kernel_initializer=init, bias_initializer=_ZEROS, kernel_regularizer=None,
                        bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, bias_constraint=None)

    def convT(self, x, filters, engine):
        return self.add("Conv2DTranspose", [x], engine, dtype="float32", filters=filters, kernel_size=[2, 2], strides=[2, 2], padding="same", data_format="channels_last",
                        dilation_rate=[1, 1], activation="linear", use_bias=True, kernel_initializer=_GLOROT, bias_initializer=_ZEROS, kernel_regularizer=None,
                        bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, bias_constraint=None, output_padding=None)

    def bn(self, x, engine):
        return self.add("BatchNormalization", [x], engine, dtype="float32", axis=-1, momentum=0.99, epsilon=0.001, center=True, scale=True, beta_initializer=_ZEROS,
                        gamma_initializer=_ONES, moving_mean_initializer=_ZEROS, moving_variance_initializer=_ONES, beta_regularizer=None,
                        gamma_regularizer=None, beta_constraint=None, gamma_constraint=None)

    def pool(self, x):
        return self.add("MaxPooling2D", [x], dtype="float32", pool_size=[2, 2], padding="valid", strides=[2, 2], data_format="channels_last")

    def drop(self, x, rate):
        return self.add("Dropout", [x], dtype="float32", rate=rate, noise_shape=None, seed=None)

    def cat(self, xs, axis=3):
        return self.add("Concatenate", xs, dtype="float32", axis=axis)


def _unet(b, hw, in_ch):
    x = b.add("InputLayer", [], batch_input_shape=[None, hw[0], hw[1], in_ch], dtype="float32", sparse=False)
    skips = {}
    for k, c in enumerate(ENC, 1):                                                       # T1:859-881
        x = b.conv(x, c, 3, "relu", _HE_NORMAL, f"c{k}a"); x = b.conv(x, c, 3, "relu", _HE_NORMAL, f"c{k}b")
        x = skips[k] = b.bn(x, f"bn{k}")
        x = b.drop(b.pool(x), 0.25)
    x = b.conv(x, 512, 3, "relu", _HE_NORMAL, "c5a"); x = b.conv(x, 512, 3, "relu", _HE_NORMAL, "c5b")      # T1:883-884
    for k, c, sk in zip((6, 7, 8, 9), DEC, (4, 3, 2, 1)):                                # T1:886-911
        x = b.cat([b.convT(x, c, f"u{k}"), skips[sk]], 3 if k == 9 else -1)             # (only T1:908 passes axis=3; T1:887, 894, 901 take the default -1)
        x = b.bn(x, f"bn{k}")
        x = b.conv(x, c, 3, "relu", _HE_NORMAL, f"c{k}a"); x = b.conv(x, c, 3, "relu", _HE_NORMAL, f"c{k}b")
    return b.conv(x, 1, 1, "sigmoid", _GLOROT, "out", padding="valid")                   # T1:913 (no padding argument: Keras' default)


def _unetpp(b, hw, in_ch):
    x = b.add("InputLayer", [], batch_input_shape=[None, hw[0], hw[1], in_ch], dtype="float32", sparse=False)
    T = {}
    nodes = {n[0]: n for n in PP_NODES}
    todo = {1: [], 2: ["x1_2"], 3: ["x2_2", "x1_3"], 4: ["x3_2", "x2_3", "x1_4"]}       # creation order UPP:876-924
    for k, c in enumerate(ENC, 1):
        x = b.conv(x, c, 3, "elu", _HE_NORMAL, f"c{k}a"); x = b.drop(x, 0.2); x = b.conv(x, c, 3, "elu", _HE_NORMAL, f"c{k}b")
        T[f"c{k}"] = b.bn(x, f"bn{k}")
        x = b.pool(T[f"c{k}"])
        for nm in todo[k]:
            _, c2, src, sk = nodes[nm]
            h = b.cat([b.convT(T[src], c2, f"u{nm[1:]}")] + [T[s] for s in sk])
            for ab in "ab":                                                              # conv_block UPP:860-868
                h = b.conv(h, c2, 3, "elu", _HE_NORMAL, f"{nm}{ab}"); h = b.drop(h, 0.4); h = b.bn(h, f"{nm}{ab}bn")
            T[nm] = h
    return b.conv(T["x1_4"], 1, 1, "sigmoid", _HE_NORMAL, "out")                         # UPP:946-947


def _order_functional(layers, out_name):
    """model.layers order of a functional Model (see the module docstring)."""
    by = {l["name"]: l for l in layers}
    index, consumers = {}, {l["name"]: [] for l in layers}
    for l in layers:
        for i in l["inbound"]:
            consumers[i].append(l["name"])
    stack_seen = set()

    def visit(n):                                         # pre-order index, inputs in call order
        if n not in index:
            index[n] = len(index)
        if n in stack_seen:
            return
        stack_seen.add(n)
        for i in by[n]["inbound"]:
            visit(i)
    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 10000))
    visit(out_name)
    depth = {}

    def d(n):
        if n not in depth:
            depth[n] = 0 if n == out_name else 1 + max(d(c) for c in consumers[n] if c in index)      # (layers that do not reach the output -- UPP:881 `p4` -- are not part of the Model)
        return depth[n]
    return sorted((l for l in layers if l["name"] in index), key=lambda l: (-d(l["name"]), index[l["name"]]))


def keras_layers(in_ch: int = 1, arch: str = "unet", hw=(224, 224)):
    """[{name, class_name, config, inbound, engine}] in model.layers order."""
    b = _Builder()
    if arch == "classifier":                                                             # Sequential, T2:747-776 (no InputLayer entry in `layers`)
        x, first = None, True
        for k, c in enumerate(CLS_C, 1):
            for ab in "ab":
                x = b.conv(x, c, 3, "relu", _HE_NORMAL, f"c{k}{ab}")
                if first:
                    b.layers[-1]["config"]["batch_input_shape"] = [None, hw[0], hw[1], in_ch]; first = False
                x = b.bn(x, f"bn{k}{ab}")
            x = b.pool(x)
        x = b.add("Flatten", [x], dtype="float32", data_format="channels_last")
        dense = dict(dtype="float32", use_bias=True, kernel_initializer=_GLOROT, bias_initializer=_ZEROS, kernel_regularizer=None, bias_regularizer=None,
                     activity_regularizer=None, kernel_constraint=None, bias_constraint=None)
        x = b.add("Dense", [x], "fc1", units=CLS_HIDDEN, activation="relu", **dense)
        x = b.drop(x, 0.4)
        b.add("Dense", [x], "fc2", units=1, activation="sigmoid", **dense)
        for l in b.layers:
            l["inbound"] = [i for i in l["inbound"] if i]
        return b.layers
    out = (_unet if arch == "unet" else _unetpp)(b, hw, in_ch)
    return _order_functional(b.layers, out)


def model_config(h: int, w: int, in_ch: int = 1, arch: str = "unet") -> dict:
    layers = keras_layers(in_ch, arch, (h, w))
    if arch == "classifier":
        return {"class_name": "Sequential", "config": {"name": "sequential_1", "layers": [{"class_name": l["class_name"], "config": l["config"]} for l in layers]},
                "keras_version": KERAS_VERSION, "backend": BACKEND}
    ent = [{"name": l["name"], "class_name": l["class_name"], "config": l["config"],
            "inbound_nodes": ([[[i, 0, 0, {}] for i in l["inbound"]]] if l["inbound"] else [])} for l in layers]
    return {"class_name": "Model", "config": {"name": "model_1", "layers": ent, "input_layers": [["input_1", 0, 0]],
                                              "output_layers": [[layers[-1]["name"], 0, 0]]}, "keras_version": KERAS_VERSION, "backend": BACKEND}


def to_json(h: int, w: int, in_ch: int = 1, arch: str = "unet") -> str:
    """model.to_json() (T1:1091-1093): the Keras 2.3 architecture description of the graph the engine runs."""
    return json.dumps(model_config(h, w, in_ch, arch))
