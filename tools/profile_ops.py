#!/usr/bin/env python
"""Per-op hipEvent profile of one training step (ops serialised): ms, TFLOP/s, GB/s per op.
    python tools/profile_ops.py [--batch 16] [--size 512] [--algo 0]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16); ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--algo", type=int, default=0); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--warm", type=int, default=25); ap.add_argument("--dtype", default="fp32"); ap.add_argument("--arch", default="unet"); ap.add_argument("--options", default="", help='JSON of context options, e.g. {"deterministic": 1}')
    a = ap.parse_args()
    import numpy as np
    import torch
    from covidseg_amd import weights as W
    from covidseg_amd.data import synthetic_ct
    from covidseg_amd.engine import HipUNet
    if a.arch == "classifier":
        from covidseg_amd.data import synthetic_classification
        xs, ys = synthetic_classification(min(a.batch, 8), a.size, seed=0); ys = ys.astype(np.float32)
    else:
        xs, ys = synthetic_ct(min(a.batch, 4), a.size, seed=0)
    r = (a.batch + len(xs) - 1) // len(xs)
    x = torch.from_numpy(np.concatenate([xs] * r)[:a.batch]).cuda(); y = torch.from_numpy(np.concatenate([ys] * r)[:a.batch]).cuda()
    eng = HipUNet(a.size, a.size, 1, conv_algo=a.algo, dtype=a.dtype, arch=a.arch, dropout_rate=0.25, options=__import__("json").loads(a.options) if a.options else None)
    eng.set_weights(W.init_weights(0, 1, a.arch, (a.size, a.size)))
    for _ in range(a.warm):          # the chip needs ~1 s of load to reach its sustained clocks
        eng.train_batch(x, y)
    torch.cuda.synchronize()
    eng.set_profiling(True, a.batch)
    for _ in range(a.reps):
        eng.train_batch(x, y)
    torch.cuda.synchronize()
    eng.set_profiling(False)
    tot = 0.0
    print(f"{'op':28s} {'ms':>8s} {'TFLOP/s':>9s} {'GB/s':>9s} {'GFLOP':>9s} {'MB':>9s}")
    for prog in (0, 1):
        for name, fl, by, ms, calls in eng.op_profile(a.batch, prog):
            ms = ms / max(calls, 1); tot += ms
            print(f"{name:28s} {ms:8.3f} {fl / ms / 1e9 if ms > 0 else 0:9.1f} {by / ms / 1e6 if ms > 0 else 0:9.0f} {fl / 1e9:9.1f} {by / 1e6:9.1f}")
    print(f"sum of op times {tot:.2f} ms")


if __name__ == "__main__":
    main()
