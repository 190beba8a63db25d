"""Generate tests/golden/model_goldens.npz: a small end-to-end known-answer fixture computed by the
CPU oracle in float64 (seeded weights + seeded synthetic batch, 32x32, batch 3, training-mode BN,
dropout off).  The -m gpu tests compare the HIP engine with these COMMITTED numbers (and, separately,
with the oracle run live).  Inputs are regenerated from seeds (numpy Generator streams are stable).

    python tests/golden/make_model_goldens.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O          # noqa: E402
from covidseg_amd.data import synthetic_ct   # noqa: E402


def main():
    w = O.init_weights(seed=123)
    x, y = synthetic_ct(3, 32, seed=7)
    r = O.loss_and_grads(w, x, y, dtype=torch.float64)
    thr = np.array([0.3, 0.5, 0.547], np.float32)
    arrs = dict(loss=np.float64(r["loss"]), dice=np.float64(r["dice"]), p=r["p"].astype(np.float32), thresholds=thr,
                x_sum=np.float64(x.astype(np.float64).sum()), y_sum=np.float64(y.astype(np.float64).sum()),
                w_checksum=np.float64(sum(float(np.abs(v.astype(np.float64)).sum()) for v in w.values())))
    for k, g in r["grads"].items():
        arrs["gnorm/" + k] = np.float64(np.linalg.norm(g))
        arrs["gsum/" + k] = np.float64(g.sum())
    for k in ("c1a/kernel", "out/kernel", "bn1/gamma", "u9/bias", "c9b/bias"):
        arrs["grad/" + k] = r["grads"][k].astype(np.float32)
    # inference-mode forward with the (initial) moving statistics, and thresholded metric sums
    with torch.no_grad():
        pi = O.forward(w, x, training=False, dtype=torch.float64)[0].numpy()
    arrs["p_infer"] = pi.astype(np.float32)
    arrs["thr_sums"] = O.threshold_sums(y, pi.astype(np.float32), thr)
    # three optimizer steps (fp64 oracle)
    tr = O.OracleTrainer({k: v.astype(np.float64) for k, v in w.items()}, torch.float64)
    arrs["traj"] = np.array([tr.train_step(x, y) for _ in range(3)])
    arrs["w_after/out/kernel"] = tr.w["out/kernel"].astype(np.float32)
    arrs["w_after/bn1/mean"] = tr.w["bn1/mean"].astype(np.float32)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_goldens.npz")
    np.savez_compressed(out, **arrs)
    print("wrote", out, os.path.getsize(out), "bytes; loss", r["loss"], "dice", r["dice"], "traj", arrs["traj"].tolist())


if __name__ == "__main__":
    main()
