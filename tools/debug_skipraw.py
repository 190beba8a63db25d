"""debug: gradients / gradient taps of the default engine (skip_raw) against options={"skip_raw": 0} on the options-test case"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import unet_oracle as O
from covidseg_amd.engine import HipUNet
def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
rng = np.random.default_rng(2)
wts = O.init_weights(seed=6)
for k in wts:
    if k.endswith("/gamma"): wts[k] = rng.uniform(0.5, 1.5, wts[k].shape).astype(np.float32)
    elif k.endswith("/beta") or k.endswith("/bias"): wts[k] = (rng.standard_normal(wts[k].shape) * 0.1).astype(np.float32)
x = rng.random((2, 64, 96, 1)).astype(np.float32); y = (rng.random((2, 64, 96, 1)) > 0.7).astype(np.float32)
a = HipUNet(64, 96, 1, dropout_rate=0.0); a.set_weights(wts); a.forward_backward(x, y); ga = a.get_grads()
b = HipUNet(64, 96, 1, dropout_rate=0.0, options={"skip_raw": 0}); b.set_weights(wts); b.forward_backward(x, y); gb = b.get_grads()
for k in ga: print(f"{k:14s} {relerr(ga[k], gb[k]):.2e}")
for nm in ("c9b", "c9a", "u9", "cat9", "c8b", "c8a", "cat8", "c5b", "p4", "c4b", "c4a", "p3", "c1b", "c1a", "bn1", "bn4"):
    try: print("grad tap", nm, f"{relerr(a.tap(2, nm, grad=True), b.tap(2, nm, grad=True)):.2e}")
    except Exception as e: print("grad tap", nm, "n/a", str(e)[:60])
tot = 0
for nm in ("c1a","c1b","c2a","c2b","c3a","c3b","c4a","c4b","c5a","c5b","c6a","c6b","c7a","c7b","c8a","c8b","c9a","c9b"):
    ta, tb = a.tap_device(2, nm), b.tap_device(2, nm)
    d = int(((ta > 0) != (tb > 0)).sum().item()); tot += d
    if d: print("sign flips", nm, d, "of", ta.numel())
print("total flips", tot)
