"""Generate tests/golden/f1_goldens.npz by EXECUTING the reference's own metric closures of the classification task.

Runs only in the build container (needs /root/reference).  The nested pure functions
  recall T2:688-692, precision T2:694-698, f1 T2:700-703   (T2 = Scripts/task2_covid19_classifcation.py)
are AST-extracted and exec'd against a NumPy-backed stand-in for ``keras.backend`` (K.round = round-half-to-even as TensorFlow's,
K.epsilon() = 1e-7).  Only the resulting input/output VECTORS are committed -- no reference source text is stored.

    python tests/golden/make_f1_goldens.py
"""
import ast
import os

import numpy as np

REF = "/root/reference/Scripts/task2_covid19_classifcation.py"
WANT = ["recall", "precision", "f1"]


class K:
    sum = staticmethod(lambda a, axis=None: np.sum(a, axis=axis))
    round = staticmethod(np.round)                       # numpy and TF both round half to even
    clip = staticmethod(np.clip)
    epsilon = staticmethod(lambda: 1e-7)


def extract(names):
    tree = ast.parse(open(REF).read())
    ns = {"K": K, "np": np}
    found = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name in names and node.name not in found:
            found[node.name] = node
    for n in names:
        exec(compile(ast.Module(body=[found[n]], type_ignores=[]), f"<ref:{n}>", "exec"), ns)
    return ns


def cases(rng):
    out = []
    for n in (1, 8, 32, 33, 250):
        t = (rng.random(n) < rng.uniform(0.2, 0.8)).astype(np.float64)
        out.append((t, rng.random(n)))
    t = np.zeros(16); out.append((t, rng.random(16)))                    # no positives at all
    t = np.ones(16); out.append((t, np.full(16, 0.2)))                    # nothing predicted
    t = (rng.random(16) > 0.5).astype(np.float64); out.append((t, t.copy()))
    p = rng.random(8); p[:4] = [0.5, 0.5000001, 0.4999999, 1.0]; out.append((np.array([1, 1, 1, 1, 0, 0, 1, 0], np.float64), p))   # the rounding edge
    return out


def main():
    ns = extract(WANT)
    rng = np.random.default_rng(20260929)
    arrs = {}
    for i, (t, p) in enumerate(cases(rng)):
        arrs[f"t{i}"] = t; arrs[f"p{i}"] = p
        for k in WANT:
            arrs[f"{k}{i}"] = np.float64(ns[k](t, p))
    arrs["n_cases"] = np.int64(i + 1)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "f1_goldens.npz")
    np.savez_compressed(out, **arrs)
    print("wrote", out, os.path.getsize(out), "bytes;", i + 1, "cases")


if __name__ == "__main__":
    main()
