"""debug: per-tensor error of the bf16-storage U-Net against the storage-emulating oracle"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import unet_oracle as O
from test_gpu_bf16_model import _case, make, relerr
h, w_, n = 64, 48, 2
wts, x, y = _case(h, w_, n, seed=h)
r = O.loss_and_grads(wts, x, y, dtype=torch.float64, want_acts=True, store=O.store_bf16)
eng = make(h, w_, dropout_rate=0.0, dtype="bf16"); eng.set_weights(wts)
ld = eng.forward_backward(x, y).cpu().numpy()
print("loss", ld, r["loss"], r["dice"])
p = eng._p_train.cpu().numpy().reshape(r["p"].shape)
print("p maxabs", np.abs(p - r["p"]).max(), "rel", relerr(p, r["p"]))
for name in r["acts"]:
    if name == "out": continue
    print("act", name, "%.2e" % relerr(eng.tap(n, name), r["acts"][name]))
for name in r["acts"]:
    if name == "out" or r["act_grads"][name] is None: continue
    masked = name[0] == "c"
    want = r["act_grads"][name] * ((r["acts"][name] > 0) if masked else 1.0)
    got = eng.tap(n, name, grad=True)
    d = got - want
    print("grad", name, "%.2e" % relerr(got, want), "max|d| %.2e max|want| %.2e" % (np.abs(d).max(), np.abs(want).max()), "nz got/want", (got != 0).mean().round(3), (want != 0).mean().round(3))
g = eng.get_grads()
for k in g: print("pg", k, "%.2e" % relerr(g[k], r["grads"][k]))
