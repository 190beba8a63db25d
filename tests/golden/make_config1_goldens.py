"""BASELINE.json configs[0]: holdout_runner_unet_infection_segmentation() on CPU, 8 synthetic 512x512 grayscale slices,
1 epoch -- run through the CPU ORACLE backend (fp32, the stand-in for the reference's Keras/CPU path) and commit the
scalars it prints: history, test loss/dice, and the Dice / IoU / precision / recall threshold tables.  The -m gpu test
runs the same runner on the HIP engine and must agree within 1e-3 (the BASELINE parity bar).

    python tests/golden/make_config1_goldens.py        (about 2 minutes of CPU)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from covidseg_amd.data import synthetic_ct                                   # noqa: E402
from covidseg_amd.runners import holdout_runner_unet_infection_segmentation   # noqa: E402
from oracle_backend import OracleBackend                                     # noqa: E402


def main():
    import tempfile
    x, y = synthetic_ct(8, 512, seed=0)
    with tempfile.TemporaryDirectory() as d:
        out = holdout_runner_unet_infection_segmentation(data=(x, y), epochs=1, dropout=False, workdir=d, verbose=0,
                                                         backend=OracleBackend(512, 512), seed=0)
    arrs = {"x_sum": np.float64(x.astype(np.float64).sum()), "y_sum": np.float64(y.astype(np.float64).sum()),
            "score": np.array(out["score"]), "dices": np.array(out["dices"]), "ious": np.array(out["ious"]),
            "new_dices": np.array(out["new_dices"]), "new_ious": np.array(out["new_ious"]),
            "precisions": np.array(out["precisions"]), "recalls": np.array(out["recalls"])}
    for k, v in out["history"].items():
        arrs["hist_" + k] = np.array(v)
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config1_goldens.npz")
    np.savez_compressed(p, **arrs)
    print("wrote", p, {k: (v.tolist() if v.size < 4 else v.shape) for k, v in arrs.items()})


if __name__ == "__main__":
    main()
