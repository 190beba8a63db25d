"""Generate tests/golden/graph_goldens.json by EXECUTING the reference's own graph-building statements.

Runs only in the build container (needs /root/reference).  For every script that builds a model on the path
  T1  task1_preprocessing_plus_unet_with_comments.py:853-915     (U-Net, holdout runner)
  T3  task3_lung_segmentation_unet.py:850-912                    (U-Net, lung runner)
  CV3 task1_crossval_3folds_unet.py:919-981, CV4 task1_crossval_4folds_unet.py:957-1019   (U-Net, k-fold runners)
  UPP task1_unet_plus_plus.py:860-949                            (U-Net++: dropout_rate / activation / conv_block + the graph)
  T2  task2_covid19_classifcation.py:747-778                     (Sequential classifier)
the statements are picked out of the runner's FunctionDef by line number (the span is found from the `Input(` / `Sequential()` statement to the
`Model(` statement / the last `model.add(`), compiled from the AST and executed against RECORDING stand-ins for the Keras names they use
(Input, Conv2D, BatchNormalization, MaxPooling2D, Dropout, Conv2DTranspose, concatenate, Flatten, Dense, Model, Sequential).  A stand-in layer
notes its class, constructor arguments, Keras auto-name (per-class counter in creation order) and, when called, its inbound tensors in call order and
its output shape.  Only the RECORDED layer list is committed (data: class, arguments, connections) -- no reference source text is stored.

    python tests/golden/make_graph_goldens.py
"""
import ast
import json
import os
import sys

REF = "/root/reference/Scripts"
SCRIPTS = {"T1": "task1_preprocessing_plus_unet_with_comments.py", "T3": "task3_lung_segmentation_unet.py", "CV3": "task1_crossval_3folds_unet.py",
           "CV4": "task1_crossval_4folds_unet.py", "UPP": "task1_unet_plus_plus.py", "T2": "task2_covid19_classifcation.py"}
NEW_DIM = 224                                  # `new_dim = 224` in every script (T1:479, T3:468, UPP:514, T2:489)
PREFIX = {"Conv2D": "conv2d", "Conv2DTranspose": "conv2d_transpose", "BatchNormalization": "batch_normalization", "MaxPooling2D": "max_pooling2d",
          "Dropout": "dropout", "Concatenate": "concatenate", "Flatten": "flatten", "Dense": "dense", "InputLayer": "input"}


class Recorder:
    def __init__(self):
        self.layers, self.counts, self.model = [], {}, None

    def new_layer(self, cls, args, kwargs):
        self.counts[cls] = self.counts.get(cls, 0) + 1
        ent = {"name": f"{PREFIX[cls]}_{self.counts[cls]}", "class_name": cls, "args": _plain(args), "kwargs": _plain(kwargs), "inbound": [], "output_shape": None}
        self.layers.append(ent)
        return ent


def _plain(v):
    if isinstance(v, (list, tuple)):
        return [_plain(a) for a in v]
    if isinstance(v, dict):
        return {k: _plain(a) for k, a in v.items()}
    assert v is None or isinstance(v, (int, float, str, bool)), v
    return v


class Tensor:
    def __init__(self, layer, shape):
        self.layer, self.shape = layer, tuple(shape)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def make_namespace(rec):
    def layer_class(cls, shape_fn):
        class L:
            def __init__(self, *args, **kwargs):
                kw = dict(kwargs)
                self.input_shape = kw.pop("input_shape", None)          # Sequential's first layer (T2:748)
                self.ent = rec.new_layer(cls, args, kw)
                if self.input_shape is not None:
                    self.ent["kwargs"]["input_shape"] = _plain(self.input_shape)
                self.args, self.kw = args, kw

            def __call__(self, x):
                xs = x if isinstance(x, (list, tuple)) else [x]
                self.ent["inbound"] = [t.layer["name"] for t in xs]
                out = Tensor(self.ent, shape_fn(self, [t.shape for t in xs]))
                self.ent["output_shape"] = list(out.shape)
                return out
        L.__name__ = cls
        return L

    def conv_shape(self, s):
        (h, w, _), = s
        assert _pair(self.kw.get("strides", 1)) == (1, 1)
        kh, kw_ = _pair(self.args[1])
        if self.kw.get("padding", "valid") == "same":
            return (h, w, self.args[0])
        return (h - kh + 1, w - kw_ + 1, self.args[0])                  # Keras default padding='valid' (the 1x1 head, T1:913)

    def convT_shape(self, s):
        (h, w, _), = s
        st = _pair(self.kw.get("strides", 1))
        assert self.kw.get("padding", "valid") == "same"
        return (h * st[0], w * st[1], self.args[0])

    def pool_shape(self, s):
        (h, w, c), = s
        ps = _pair(self.args[0] if self.args else self.kw.get("pool_size", 2))
        assert self.kw.get("padding", "valid") == "valid" and self.kw.get("strides") is None
        return (h // ps[0], w // ps[1], c)

    def same(self, s):
        return s[0]

    def flat(self, s):
        n = 1
        for d in s[0]:
            n *= d
        return (n,)

    def dense(self, s):
        return (self.args[0],)

    Concat = layer_class("Concatenate", lambda self, s: s[0][:-1] + (sum(t[-1] for t in s),))

    def concatenate(tensors, axis=-1, **kw):
        assert not kw and axis in (-1, 3), (axis, kw)                   # channels_last: axis 3 == -1 for [N,H,W,C]
        return Concat(axis=axis)(list(tensors))

    def Input(shape, **kw):
        assert not kw
        ent = rec.new_layer("InputLayer", (), {"shape": list(shape)})
        ent["output_shape"] = list(shape)
        return Tensor(ent, shape)

    class Model:
        def __init__(self, inputs, outputs, **kw):
            assert not kw
            rec.model = {"kind": "Model", "inputs": [t.layer["name"] for t in inputs], "outputs": [t.layer["name"] for t in outputs]}

        def summary(self, *a, **k):
            return None

    class Sequential:
        def __init__(self):
            self.t = None
            rec.model = {"kind": "Sequential", "layers": []}

        def add(self, layer):
            if self.t is None:
                assert layer.input_shape is not None
                self.t = Tensor({"name": None}, layer.input_shape)
                out = layer(self.t)
                layer.ent["inbound"] = []
            else:
                out = layer(self.t)
            self.t = out
            rec.model["layers"].append(layer.ent["name"])

        def summary(self, *a, **k):
            return None

    return {"Input": Input, "Conv2D": layer_class("Conv2D", conv_shape), "Conv2DTranspose": layer_class("Conv2DTranspose", convT_shape),
            "BatchNormalization": layer_class("BatchNormalization", same), "MaxPooling2D": layer_class("MaxPooling2D", pool_shape),
            "Dropout": layer_class("Dropout", same), "Flatten": layer_class("Flatten", flat), "Dense": layer_class("Dense", dense),
            "concatenate": concatenate, "Model": Model, "Sequential": Sequential, "new_dim": NEW_DIM, "print": lambda *a, **k: None}


def _calls(node, name):
    return any(isinstance(n, ast.Call) and ((isinstance(n.func, ast.Name) and n.func.id == name) or (isinstance(n.func, ast.Attribute) and n.func.attr == name))
               for n in ast.walk(node))


def graph_statements(tag):
    """the top-level statements of the runner that build the model, found by what they call; returns (statements, first line, last line)"""
    tree = ast.parse(open(os.path.join(REF, SCRIPTS[tag])).read())
    runner = next(n for n in tree.body if isinstance(n, ast.FunctionDef))
    body = runner.body
    if tag == "T2":
        first = next(i for i, s in enumerate(body) if isinstance(s, ast.Assign) and isinstance(s.value, ast.Call) and isinstance(s.value.func, ast.Name)
                     and s.value.func.id == "Sequential")          # (`iaa.Sequential` of the unused augmentation, T2:559, is an Attribute call)
        last = max(i for i, s in enumerate(body) if i > first and isinstance(s, ast.Expr) and _calls(s, "add") and
                   isinstance(s.value.func, ast.Attribute) and isinstance(s.value.func.value, ast.Name) and s.value.func.value.id == "model")
    else:
        first = next(i for i, s in enumerate(body) if isinstance(s, ast.Assign) and _calls(s, "Input"))
        last = next(i for i, s in enumerate(body) if i > first and isinstance(s, ast.Assign) and _calls(s, "Model"))
        if tag == "UPP":                       # conv_block and the two names it closes over (UPP:860-868) sit right in front of the graph
            cb = next(i for i, s in enumerate(body) if isinstance(s, ast.FunctionDef) and s.name == "conv_block")
            k = cb
            while k > 0 and isinstance(body[k - 1], ast.Assign) and all(isinstance(t, ast.Name) and t.id in ("dropout_rate", "activation") for t in body[k - 1].targets):
                k -= 1
            assert first == cb + 1, "conv_block is expected directly in front of `inputs = Input(...)`"
            first = k
    stmts = body[first:last + 1]
    return stmts, stmts[0].lineno, max(getattr(s, "end_lineno", s.lineno) for s in stmts)


def record(tag):
    stmts, l0, l1 = graph_statements(tag)
    rec = Recorder()
    ns = make_namespace(rec)
    exec(compile(ast.Module(body=stmts, type_ignores=[]), f"<ref:{tag}:{l0}-{l1}>", "exec"), ns)
    return {"source": f"{SCRIPTS[tag]}:{l0}-{l1}", "new_dim": NEW_DIM, "model": rec.model, "layers": rec.layers}


def main():
    out = {tag: record(tag) for tag in SCRIPTS}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_goldens.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for tag, g in out.items():
        print(tag, g["source"], len(g["layers"]), "layers", g["model"]["kind"])
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference not present; goldens are generated in the build container only")
    main()
