"""the first lines of tools/gpu/asan_check.sh's run on whatever library COVIDSEG_AMD_LIB names (default: the product)"""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from covidseg_amd.data import synthetic_ct
from covidseg_amd.engine import HipUNet
from covidseg_amd import weights as W, _lib
print("library:", _lib.LIB_PATH)
x, y = synthetic_ct(3, 64, seed=1)
print("label mean", float(y.mean()))
for arch, opts in (("unet", None), ("unet", {"head_fused": 0, "skip_raw": 0}), ("unetpp", None)):
    eng = HipUNet(64, 64, 1, device=0, arch=arch, options=opts, dropout_rate=0.25 if arch == "unet" else 0.2, private_context=True)
    eng.set_weights(W.init_weights(0, 1, arch, (64, 64)))
    for n in (3, 2, 3, 2):
        print(arch, opts, n, eng.train_batch(x[:n], y[:n]).cpu().numpy())
    eng.close()
