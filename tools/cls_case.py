"""Gradient errors of the classifier's 96 x 128 x 24 live-oracle case (tests/test_gpu_classifier.py) under the current environment switches."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import unet_oracle as O
import test_gpu_classifier as T
h, w_, n = 96, 128, 24
rng = np.random.default_rng(h + n)
wts = T.rand_weights(h, (h, w_))
x = rng.random((n, h, w_, 1)).astype(np.float32); y = (rng.random(n) > 0.5).astype(np.float32)
cw = (0.8, 1.4)
r = O.cls_loss_and_grads(wts, x, y, class_weights=cw, dtype=torch.float64, want_acts=True)
eng = T.make(h, w_, conv_algo=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
eng.set_weights(wts); eng.set_class_weights(*cw)
ld = eng.forward_backward(x, y).cpu().numpy()
convs = [f"c{k}{ab}" for k in (1, 2, 3) for ab in "ab"]
flips = T.count_flips(eng, n, r["acts"], convs)
r2 = O.cls_loss_and_grads(wts, x, y, class_weights=cw, dtype=torch.float64, relu_masks={c: (eng.tap(n, c) > 0).astype(np.float64) for c in convs}, pool_sel={f'p{k}': O.pool_selection(eng.tap(n, f'bn{k}b')) for k in (1, 2, 3)}, want_acts=True)
g = eng.get_grads()
print("flips", flips, "loss err", abs(ld[0] - r["loss"]))
for k in g:
    print(f"{k:14s} vs oracle {T.relerr(g[k], r['grads'][k]):.2e}  vs oracle on engine masks {T.relerr(g[k], r2['grads'][k]):.2e}")
for nm in ("c3b", "c3a", "c2b", "c2a", "c1b"):
    try:
        got = eng.tap(n, nm, grad=True); want = r2["act_grads"][nm] * (eng.tap(n, nm) > 0)
        a = np.abs(want[want != 0]); print(f"dgrad {nm}: relerr {T.relerr(got, want):.2e}  |dy| max {a.max():.2e} median {np.median(a):.2e} p1 {np.percentile(a, 1):.2e}")
    except Exception as e:
        print("tap", nm, "n/a", str(e)[:60])
for nm in convs + ["bn1a", "bn2a", "bn2b", "bn3a", "bn3b", "p3"]:
    a, b = eng.tap(n, nm), r["acts"][nm]
    print(f"fwd {nm}: relerr {T.relerr(a, b):.2e}  max abs err {np.abs(a - b).max():.2e}  |x| max {np.abs(b).max():.2e} median {np.median(np.abs(b)):.2e}")
