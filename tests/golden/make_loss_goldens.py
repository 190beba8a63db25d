"""Generate tests/golden/loss_goldens.npz by EXECUTING the reference's own loss closures.

Runs only in the build container (needs /root/reference).  The reference's nested pure
functions are AST-extracted from
  /root/reference/Scripts/task1_preprocessing_plus_unet_with_comments.py
    dice_coeff T1:784-790, dice_loss T1:792-794, weighted_bce_loss T1:819-825,
    tversky_loss T1:801-816 region, weighted_dice_loss T1:827-833 region
and exec'd against a NumPy-backed stand-in for ``keras.backend`` (``K``).  Only the
resulting input/output VECTORS are committed -- no reference source text is stored.

    python tests/golden/make_loss_goldens.py
"""
import ast
import os
import sys

import numpy as np

REF = "/root/reference/Scripts/task1_preprocessing_plus_unet_with_comments.py"
WANT = ["dice_coeff", "dice_loss", "weighted_bce_loss"]


class K:  # minimal float64 NumPy shim of the keras.backend calls those closures make
    flatten = staticmethod(lambda a: np.asarray(a).reshape(-1))
    sum = staticmethod(lambda a, axis=None: np.sum(a, axis=axis))
    clip = staticmethod(np.clip)
    log = staticmethod(np.log)
    exp = staticmethod(np.exp)
    abs = staticmethod(np.abs)
    maximum = staticmethod(np.maximum)
    mean = staticmethod(lambda a, axis=None: np.mean(a, axis=axis))


def extract(names):
    src = open(REF).read()
    tree = ast.parse(src)
    ns = {"K": K, "np": np}
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names:
            found[node.name] = node
    for n in names:
        mod = ast.Module(body=[found[n]], type_ignores=[])
        exec(compile(mod, f"<ref:{n}>", "exec"), ns)
    return ns


def cases(rng):
    out = []
    for shape in [(1, 8, 8, 1), (2, 16, 16, 1), (3, 32, 32, 1), (5, 64, 64, 1)]:
        t = np.round(rng.random(shape) ** 3 * 255) / 255.0           # soft labels k/255
        t[rng.random(shape) < 0.5] = 0.0
        p = rng.random(shape)
        out.append((t, p))
    t = np.zeros((2, 8, 8, 1)); out.append((t, np.full_like(t, 0.25)))   # empty mask
    t = np.ones((2, 8, 8, 1)); out.append((t, np.full_like(t, 0.9)))      # full mask
    t = (rng.random((2, 8, 8, 1)) > 0.5).astype(float); out.append((t, t.copy()))       # p == t (clip path)
    p = rng.random((2, 8, 8, 1)); p.flat[:4] = [0.0, 1.0, 1e-9, 1 - 1e-9]               # clip edges
    out.append((t, p))
    return out


def main():
    ns = extract(WANT)
    rng = np.random.default_rng(20260928)
    arrs = {}
    for i, (t, p) in enumerate(cases(rng)):
        arrs[f"t{i}"] = t
        arrs[f"p{i}"] = p
        arrs[f"dice_coeff{i}"] = np.float64(ns["dice_coeff"](t, p))
        arrs[f"dice_loss{i}"] = np.float64(ns["dice_loss"](t, p))
        # weighted_bce_loss(t,p,ones) == mean Keras BCE (T1:819-825), pins the BCE half
        arrs[f"bce_mean{i}"] = np.float64(ns["weighted_bce_loss"](t, p, np.ones_like(t)))
    arrs["n_cases"] = np.int64(i + 1)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loss_goldens.npz")
    np.savez_compressed(out, **arrs)
    print("wrote", out, os.path.getsize(out), "bytes;", i + 1, "cases")


if __name__ == "__main__":
    if not os.path.exists(REF):
        sys.exit("reference not present; goldens are generated in the build container only")
    main()
