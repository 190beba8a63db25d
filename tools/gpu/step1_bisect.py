import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from oracle import unet_oracle as O
from covidseg_amd.engine import HipUNet
rng = np.random.default_rng(21)
wts = O.init_weights(seed=8)
x = rng.random((3, 64, 64, 1)).astype(np.float32); y = (rng.random((3, 64, 64, 1)) > 0.75).astype(np.float32)
engs = {}
for name, opts in (("fused", None), ("unfused", {"bn_fuse_stats": 0})):
    e = HipUNet(64, 64, 1, dropout_rate=0.0, options=opts, private_context=True); e.set_weights(wts); e.forward_backward(x, y); engs[name] = e
def rel(a, b): return np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30)
print("forward taps (fused vs unfused)")
for nm in ("c1b", "p1", "c2b", "p2", "c3b", "p3", "c4b", "p4", "c5b", "u6", "c6a", "c6b", "u7", "c7a", "c7b", "u8", "c8a", "c8b", "u9", "c9a"):
    try:
        a, b = engs["fused"].tap(3, nm), engs["unfused"].tap(3, nm)
        print(f"  {nm:5s} {rel(a, b):.2e}")
    except Exception as ex:
        print("  ", nm, "tap failed", str(ex)[:60])
print("gradient taps, backward order")
for nm in ("c9a", "bn9", "u9", "c8b", "c8a", "u8", "c7b", "c7a", "u7", "c6b", "c6a", "u6", "c5b", "c5a", "p4", "c4b", "c4a", "p3", "c3b", "c3a", "p2", "c2b", "c2a", "p1", "c1b", "c1a"):
    try:
        a, b = engs["fused"].tap(3, nm, grad=True), engs["unfused"].tap(3, nm, grad=True)
        print(f"  d{nm:5s} {rel(a, b):.2e}")
    except Exception as ex:
        print("  ", nm, "grad tap failed", str(ex)[:60])
sf, su = engs["fused"].state.cpu().numpy(), engs["unfused"].state.cpu().numpy()
print("moving statistics max rel diff", np.abs(sf - su).max() / np.abs(su).max())
a, b = engs["fused"].tap(3, "c9a", grad=True).astype(np.float64), engs["unfused"].tap(3, "c9a", grad=True).astype(np.float64)
print("dc9a: scale fused/unfused", (a * b).sum() / (b * b).sum(), " residual after scaling", np.linalg.norm(a - b * (a * b).sum() / (b * b).sum()) / np.linalg.norm(b))
d = np.abs(a - b); i = np.unravel_index(d.argmax(), d.shape); print("worst element", i, a[i], b[i], " fraction of elements differing > 1e-3 rel:", (d > 1e-3 * np.abs(b).max()).mean())
per_img = [np.linalg.norm(a[k] - b[k]) / np.linalg.norm(b[k]) for k in range(3)]; print("per image", per_img)
rows = np.sqrt(((a - b) ** 2).sum(axis=(0, 2, 3))) ; print("rows with the largest difference", np.argsort(rows)[-6:], rows.max() / max(np.median(rows), 1e-30))
cols = np.sqrt(((a - b) ** 2).sum(axis=(0, 1, 3))) ; print("cols with the largest difference", np.argsort(cols)[-6:], cols.max() / max(np.median(cols), 1e-30))
for e in engs.values():
    ld = e._loss_tensor(e._plan(3)).cpu().numpy(); print("loss pair", ld)
