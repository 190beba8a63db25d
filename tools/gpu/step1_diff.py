import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import unet_oracle as O
from covidseg_amd.engine import HipUNet
rng = np.random.default_rng(21)
wts = O.init_weights(seed=8)
x = rng.random((3, 64, 64, 1)).astype(np.float32); y = (rng.random((3, 64, 64, 1)) > 0.75).astype(np.float32)
tr = O.OracleTrainer({k: v.astype(np.float64) for k, v in wts.items()}, torch.float64)
l0 = tr.train_step(x, y); wo = {k: np.asarray(v, np.float64) for k, v in tr.get_weights().items()} if hasattr(tr, "get_weights") else None
res = {}
for name, opts in (("fused", None), ("unfused", {"bn_fuse_stats": 0})):
    eng = HipUNet(64, 64, 1, dropout_rate=0.0, options=opts); eng.set_weights(wts)
    eng.forward_backward(x, y); g = eng.get_grads(); eng.adam_step(); w = eng.get_weights()
    res[name] = (g, w)
lr = 5e-4
rows = []
for k in res["fused"][1]:
    if k.endswith(("/mean", "/var")): continue
    dw = np.abs(res["fused"][1][k] - res["unfused"][1][k]).max() / lr
    gf, gu = res["fused"][0][k], res["unfused"][0][k]
    rows.append((dw, k, np.abs(gf).max(), np.abs(gf - gu).max()))
for dw, k, gm, gd in sorted(rows, reverse=True)[:14]:
    print(f"{k:14s} max|dw|/lr {dw:.2e}   max|g| {gm:.2e}  max|g_fused - g_unfused| {gd:.2e}")
# both against the float64 oracle's gradients of the same step (no sign patterns fed: flips count as error)
og = O.loss_and_grads({k: v.astype(np.float64) for k, v in wts.items()}, x, y, dtype=torch.float64)["grads"]
for k in ("c1b/kernel", "c2b/kernel", "c3b/kernel", "c2a/kernel", "c8a/kernel", "c5a/kernel"):
    o = np.asarray(og[k], np.float64)
    for nm in ("fused", "unfused"):
        g = res[nm][0][k].astype(np.float64)
        print(f"{k:12s} {nm:8s} rel L2 vs fp64 {np.linalg.norm(g - o) / np.linalg.norm(o):.2e}   max abs {np.abs(g - o).max():.2e}")
