"""A/B of the encoder-tail backward forms on the GPU in ONE process: two engines on contexts with UNET_OPT_ENC_BN_FUSED = 0 / 1 (one fused apply pass with
the sums from the pooled tensors + closed-form skip term, vs pool_bwd_bnstats + bn_bwd_apply), every parameter gradient compared.
    python tools/check_enc_fused.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import covidseg_amd  # noqa: F401
from covidseg_amd.engine import HipUNet
from oracle import unet_oracle as O
rng = np.random.default_rng(5)
h, w, n = 64, 96, 3
wts = O.init_weights(seed=3)
for k in wts:
    if k.endswith("/bias") or k.endswith("/beta"):
        wts[k] = (rng.standard_normal(wts[k].shape) * 0.3).astype(np.float32)
    if k.endswith("/gamma"):
        wts[k] = (rng.uniform(0.05, 0.4, wts[k].shape) * rng.choice([-1, 1], wts[k].shape)).astype(np.float32)      # small |gamma|: eps / (var + eps) is large
x = rng.random((n, h, w, 1)).astype(np.float32); y = (rng.random((n, h, w, 1)) > 0.7).astype(np.float32)
grads = []
for fused in (0, 1):
    eng = HipUNet(h, w, 1, dropout_rate=0.25, options={"enc_bn_fused": fused})
    eng.set_weights(wts)
    eng.forward_backward(x, y)
    grads.append(eng.get_grads())
worst = 0.0
for k in grads[0]:
    d = float(np.linalg.norm(grads[0][k] - grads[1][k]) / (np.linalg.norm(grads[0][k]) + 1e-30))
    worst = max(worst, d)
    if "bn" in k or d > 1e-5:
        print(f"{k:16s} rel diff {d:.2e}")
print("worst", worst)
