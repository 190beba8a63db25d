"""The three graphs, pinned to the reference MECHANICALLY: tests/golden/graph_goldens.json is the layer list the reference's own graph-building statements
(T1:853-915, T3:850-912, CV3:919-981, CV4:957-1019, UPP:860-949, T2:747-778) produced when they were executed against recording stand-ins for the Keras
constructors (tests/golden/make_graph_goldens.py).  Checked against it here:
  * the oracle's layer tables (kind, channels, creation order) and -- numerically -- the CONNECTIVITY of the oracle's forward functions: a generic
    interpreter walks the recorded layer list (class, arguments, inbound tensors in call order) with torch functional ops and must reproduce
    oracle.forward / pp_forward / cls_forward on random weights (a swapped concat order, a wrong activation, filter count or skip source shows up at once);
  * the product's tables (weights.layer_table / keras_names) and the Keras graph description it serialises (keras_graph: class, arguments, auto-names,
    inbound order of every layer, weight-less ones included).
What stays outside the pin: the semantics of the Keras ops themselves (the interpreter uses the same documented semantics)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from covidseg_amd import keras_graph as KG
from covidseg_amd import weights as W
from oracle import unet_oracle as O

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graph_goldens.json")))
ARCHS = {"unet": "T1", "unetpp": "UPP", "classifier": "T2"}
KIND = {"Conv2DTranspose": "convT", "BatchNormalization": "bn", "Dense": "dense"}


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def weighted(tag):
    """[(keras name, kind, cin, cout)] of the recorded layers that carry weights, in creation order; cin from the recorded tensor shapes"""
    by = {l["name"]: l for l in G[tag]["layers"]}
    out, prev = [], None
    for l in G[tag]["layers"]:
        c = l["class_name"]
        if c in ("Conv2D", "Conv2DTranspose", "BatchNormalization", "Dense"):
            if l["inbound"]:
                cin = by[l["inbound"][0]]["output_shape"][-1]
            else:                                                    # first layer of the Sequential: its input_shape argument
                cin = l["kwargs"]["input_shape"][-1]
            if G[tag]["model"]["kind"] == "Sequential" and prev is not None and not l["inbound"]:
                cin = prev["output_shape"][-1]
            kind = KIND.get(c) or {(3, 3): "conv3", (1, 1): "conv1"}[_pair(l["args"][1])]
            cout = l["output_shape"][-1]
            out.append((l["name"], kind, cin, cout))
        prev = l
    return out


def test_the_four_unet_scripts_build_the_same_graph():
    for tag in ("T3", "CV3", "CV4"):
        assert G[tag]["layers"] == G["T1"]["layers"] and G[tag]["model"] == G["T1"]["model"], tag
    assert [G[t]["source"].split(":")[1] for t in ("T1", "T3", "CV3", "CV4", "UPP", "T2")] == ["853-915", "850-912", "919-981", "957-1019", "860-949", "747-778"]


@pytest.mark.parametrize("arch", ["unet", "unetpp", "classifier"])
def test_layer_tables_follow_the_recorded_creation_order(arch):
    rec = weighted(ARCHS[arch])
    otab = {"unet": O.layer_table, "unetpp": O.pp_layer_table, "classifier": O.cls_layer_table}[arch](1)
    ptab = W.layer_table(1, arch, (224, 224))
    assert [(k, ci, co) for _, k, ci, co in rec] == [(k, ci, co) for _, k, ci, co in otab]
    assert [tuple(r) for r in otab] == [tuple(r) for r in ptab]
    # Keras auto-names of the product's weight map == the recorded names, layer by layer
    names = W.keras_names(1, arch, (224, 224))
    eng_to_keras = {}
    for k, v in names.items():
        eng_to_keras.setdefault(k.split("/")[0], v.split("/")[0])
    assert [eng_to_keras[n] for n, *_ in ptab] == [n for n, *_ in rec]
    total = {"unet": 7765281, "unetpp": 2209697, "classifier": 1678385}[arch]          # (the classifier's count is the one Keras printed, NB2:1704)
    assert W.count_params(1, arch, (224, 224))[0] == total


def _norm_init(l):
    ki = l["kwargs"].get("kernel_initializer")
    return {"he_normal": "he_normal", None: "glorot_uniform"}[ki]


@pytest.mark.parametrize("arch", ["unet", "unetpp", "classifier"])
def test_keras_graph_description_matches_the_recording(arch):
    tag = ARCHS[arch]
    rec = {l["name"]: l for l in G[tag]["layers"]}
    mine = {l["name"]: l for l in KG.keras_layers(1, arch, (224, 224))}
    reach = set(mine)
    if arch == "unetpp":
        # UPP:881 builds `p4 = MaxPooling2D(...)(c4)` but nothing consumes it (the c5 path is commented out, UPP:926-944): Keras leaves it out of the Model
        dead = [n for n, l in rec.items() if n not in mine]
        assert dead == ["max_pooling2d_4"] and not any("max_pooling2d_4" in l["inbound"] for l in rec.values())
    else:
        assert set(rec) == reach
    for name, m in mine.items():
        r = rec[name]
        assert m["class_name"] == r["class_name"], name
        assert m["inbound"] == r["inbound"], name                      # incl. the order of a concatenate's inputs
        cfg, a, kw = m["config"], r["args"], r["kwargs"]
        c = m["class_name"]
        if c == "Conv2D":
            assert (cfg["filters"], tuple(cfg["kernel_size"]), cfg["activation"], cfg["padding"]) == (a[0], _pair(a[1]), kw["activation"], kw.get("padding", "valid")), name
            assert cfg["kernel_initializer"] == {"he_normal": KG._HE_NORMAL, "glorot_uniform": KG._GLOROT}[_norm_init(r)], name
            assert tuple(cfg["strides"]) == (1, 1) and "strides" not in kw
        elif c == "Conv2DTranspose":
            assert (cfg["filters"], tuple(cfg["kernel_size"]), tuple(cfg["strides"]), cfg["padding"], cfg["activation"]) == (a[0], _pair(a[1]), _pair(kw["strides"]), kw["padding"], "linear"), name
            assert cfg["kernel_initializer"] == KG._GLOROT and "kernel_initializer" not in kw
        elif c == "MaxPooling2D":
            assert tuple(cfg["pool_size"]) == _pair(a[0] if a else kw["pool_size"]) and cfg["padding"] == "valid", name
        elif c == "Dropout":
            assert cfg["rate"] == a[0], name
        elif c == "Concatenate":
            assert cfg["axis"] == kw["axis"], name
        elif c == "Dense":
            assert (cfg["units"], cfg["activation"]) == (a[0], kw["activation"]), name
        elif c == "BatchNormalization":
            assert not a and not kw                                    # constructed with Keras' defaults everywhere (momentum .99, eps 1e-3)
        elif c == "InputLayer":
            assert cfg["batch_input_shape"][1:] == kw["shape"]
    if arch == "classifier":
        assert [l["name"] for l in KG.keras_layers(1, arch, (224, 224))] == G[tag]["model"]["layers"]
        assert mine["conv2d_1"]["config"]["batch_input_shape"][1:] == rec["conv2d_1"]["kwargs"]["input_shape"]
    else:
        cfg = KG.model_config(224, 224, 1, arch)["config"]
        assert [i[0] for i in cfg["input_layers"]] == G[tag]["model"]["inputs"] and [o[0] for o in cfg["output_layers"]] == G[tag]["model"]["outputs"]


# ---------------------------------------------------------------------------------------------------------------------------------
# a generic interpreter of the recorded layer list (inference mode): documented Keras semantics per class, connectivity from the recording
# ---------------------------------------------------------------------------------------------------------------------------------
ACT = {"relu": torch.relu, "elu": F.elu, "sigmoid": torch.sigmoid, None: lambda t: t}


def interpret(tag, weights_in_creation_order, x):
    """x [N,H,W,C] float64 -> output of the recorded model; weights: list (one entry per weighted layer, creation order) of dicts"""
    g = G[tag]
    wq = list(weights_in_creation_order)
    T = {}
    seq_prev = None
    for l in g["layers"]:
        c, a, kw = l["class_name"], l["args"], l["kwargs"]
        if c == "InputLayer":
            T[l["name"]] = x
            continue
        ins = [T[i] for i in l["inbound"]] if l["inbound"] else [x if seq_prev is None else seq_prev]
        if g["model"]["kind"] == "Sequential":
            ins = [x if seq_prev is None else seq_prev]
        h = ins[0]
        if c == "Conv2D":
            w = wq.pop(0)
            k = _pair(a[1])
            assert w["kernel"].shape == (k[0], k[1], h.shape[-1], a[0]), (l["name"], w["kernel"].shape)
            pad = (k[0] // 2, k[1] // 2) if kw.get("padding", "valid") == "same" else (0, 0)
            y = F.conv2d(h.permute(0, 3, 1, 2), w["kernel"].permute(3, 2, 0, 1), w["bias"], padding=pad).permute(0, 2, 3, 1)
            y = ACT[kw.get("activation")](y)
        elif c == "Conv2DTranspose":
            w = wq.pop(0)                                              # Keras kernel [kh, kw, out, in]
            assert w["kernel"].shape == (2, 2, a[0], h.shape[-1]) and _pair(kw["strides"]) == (2, 2)
            y = F.conv_transpose2d(h.permute(0, 3, 1, 2), w["kernel"].permute(3, 2, 0, 1), w["bias"], stride=2).permute(0, 2, 3, 1)
        elif c == "BatchNormalization":
            w = wq.pop(0)
            y = (h - w["mean"]) / torch.sqrt(w["var"] + 1e-3) * w["gamma"] + w["beta"]
        elif c == "MaxPooling2D":
            y = F.max_pool2d(h.permute(0, 3, 1, 2), _pair(a[0] if a else kw["pool_size"])).permute(0, 2, 3, 1)
        elif c == "Dropout":
            y = h                                                      # inference
        elif c == "Concatenate":
            y = torch.cat(ins, dim=3)
        elif c == "Flatten":
            y = h.reshape(h.shape[0], -1)
        elif c == "Dense":
            w = wq.pop(0)
            y = ACT[kw.get("activation")](h @ w["kernel"] + w["bias"])
        else:
            raise AssertionError(c)
        if y.dim() == 4:                                              # recorded at new_dim = 224; run here at a smaller size
            r = l["output_shape"]
            assert list(y.shape[1:]) == [r[0] * x.shape[1] // g["new_dim"], r[1] * x.shape[2] // g["new_dim"], r[2]], l["name"]
        T[l["name"]] = y
        seq_prev = y
    assert not wq
    return T[g["model"]["outputs"][0]] if g["model"]["kind"] == "Model" else seq_prev


def _random_weights(table, rng):
    w = {}
    for name, kind, cin, cout in table:
        if kind == "bn":
            w[name + "/gamma"] = rng.uniform(0.5, 1.5, cout); w[name + "/beta"] = rng.normal(0, 0.3, cout)
            w[name + "/mean"] = rng.normal(0, 0.3, cout); w[name + "/var"] = rng.uniform(0.5, 1.5, cout)
        else:
            shape = {"conv3": (3, 3, cin, cout), "conv1": (1, 1, cin, cout), "convT": (2, 2, cout, cin), "dense": (cin, cout)}[kind]
            fan = np.prod(shape[:-1]) if kind != "convT" else 4 * cin
            w[name + "/kernel"] = rng.normal(0, (2.0 / fan) ** 0.5, shape); w[name + "/bias"] = rng.normal(0, 0.2, cout)
    return w


@pytest.mark.parametrize("arch", ["unet", "unetpp", "classifier"])
def test_oracle_forward_has_the_recorded_connectivity(arch):
    rng = np.random.default_rng(11)
    size = 32 if arch != "classifier" else 16
    table = {"unet": O.layer_table, "unetpp": O.pp_layer_table, "classifier": lambda c: O.cls_layer_table(c, (size, size))}[arch](1)
    w = _random_weights(table, rng)
    x = rng.random((2, size, size, 1))
    fwd = {"unet": O.forward, "unetpp": O.pp_forward, "classifier": O.cls_forward}[arch]
    want = fwd(w, x, training=False, dtype=torch.float64)[0].detach()
    per_layer = [{k.split("/")[1]: torch.from_numpy(np.asarray(v)) for k, v in w.items() if k.split("/")[0] == name} for name, *_ in table]
    got = interpret(ARCHS[arch], per_layer, torch.from_numpy(x))
    assert got.reshape(-1).shape == want.reshape(-1).shape
    assert float((got.reshape(-1) - want.reshape(-1)).abs().max()) < 1e-12
    if arch == "unet":
        # sensitivity: the same interpreter with one concatenate's inputs swapped is far off -- the comparison does see the order
        g = G["T1"]
        cat = next(l for l in g["layers"] if l["class_name"] == "Concatenate")
        cat["inbound"].reverse()
        try:
            bad = interpret("T1", per_layer, torch.from_numpy(x))
        finally:
            cat["inbound"].reverse()
        assert float((bad.reshape(-1) - want.reshape(-1)).abs().max()) > 1e-3
