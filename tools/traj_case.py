#!/usr/bin/env python
"""Loss / Dice trajectory of N optimizer steps of the fp32 U-Net on a fixed synthetic batch (one JSON line): the subprocess half of
tests/test_gpu_model.py::test_forty_step_trajectory_h2_vs_fp32_mfma (the kernel-family switches are read once per process).
    python tools/traj_case.py [steps] [size] [batch]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    import numpy as np
    from covidseg_amd import weights as W
    from covidseg_amd.data import synthetic_ct
    from covidseg_amd.engine import HipUNet
    x, y = synthetic_ct(batch, size, seed=11)
    eng = HipUNet(size, size, 1, dropout_rate=0.0)
    eng.set_weights(W.init_weights(5, 1, "unet", (size, size)))
    out = [eng.train_batch(x, y).cpu().numpy().tolist() for _ in range(steps)]
    p, ld = eng.predict_batch(x, y)
    print(json.dumps({"traj": out, "final": ld.cpu().numpy().tolist(), "p_mean": float(p.float().mean())}))


if __name__ == "__main__":
    main()
