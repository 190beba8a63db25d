"""Numerical triage: gradient error of HIP (MFMA / direct) and of the fp32 CPU oracle, each vs the fp64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import unet_oracle as O
from covidseg_amd.engine import HipUNet

def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))

h, w_, n = 64, 48, 2
rng = np.random.default_rng(h)
wts = O.init_weights(seed=h)
for k in wts:
    if k.endswith("/bias") or k.endswith("/beta"): wts[k] = (rng.standard_normal(wts[k].shape) * 0.1).astype(np.float32)
    if k.endswith("/gamma"): wts[k] = rng.uniform(0.5, 1.5, wts[k].shape).astype(np.float32)
x = rng.random((n, h, w_, 1)).astype(np.float32)
y = (np.round(rng.random((n, h, w_, 1)) ** 4 * 255) / 255).astype(np.float32)
r64 = O.loss_and_grads(wts, x, y, dtype=torch.float64, want_acts=True)
r32 = O.loss_and_grads(wts, x, y, dtype=torch.float32, want_acts=True)
res = {}
for algo in (0, 1):
    eng = HipUNet(h, w_, 1, conv_algo=algo, dropout_rate=0.0); eng.set_weights(wts)
    eng.forward_backward(x, y); res[algo] = (eng.get_grads(), {k: eng.tap(n, k) for k in r64["acts"] if k != "out"})
print(f"{'tensor':14s} {'mfma':>10s} {'direct':>10s} {'cpu-fp32':>10s}   (relative L2 error vs fp64 oracle)")
for k in r64["grads"]:
    print(f"{k:14s} {relerr(res[0][0][k], r64['grads'][k]):10.2e} {relerr(res[1][0][k], r64['grads'][k]):10.2e} {relerr(r32['grads'][k], r64['grads'][k]):10.2e}")
print("relu-mask flips vs fp64 (count of elements whose >0 state differs):")
for k in ("c1a","c1b","c2a","c2b","c3a","c3b","c4a","c4b","c5a","c5b","c6a","c6b","c7a","c7b","c8a","c8b","c9a","c9b"):
    m64 = r64["acts"][k] > 0
    print(f"  {k}: mfma {int(((res[0][1][k] > 0) != m64).sum())} direct {int(((res[1][1][k] > 0) != m64).sum())} cpu-fp32 {int(((r32['acts'][k] > 0) != m64).sum())}")
