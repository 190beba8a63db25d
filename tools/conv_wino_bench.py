#!/usr/bin/env python
"""Direct MFMA vs Winograd-F(2,3) forward conv, layer by layer (U-Net shapes at 512x512, batch 16): ms and effective TFLOP/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from covidseg_amd import _lib

LAYERS = [("c1b", 512, 32, 32), ("c2a", 256, 32, 64), ("c2b", 256, 64, 64), ("c3a", 128, 64, 128), ("c3b", 128, 128, 128), ("c4a", 64, 128, 256),
          ("c4b", 64, 256, 256), ("c5a", 32, 256, 512), ("c5b", 32, 512, 512), ("c6a", 64, 512, 256), ("c7a", 128, 256, 128), ("c8a", 256, 128, 64),
          ("c9a", 512, 64, 32), ("d9a", 512, 32, 64), ("d8a", 256, 64, 128)]


def main():
    import argparse
    ap = argparse.ArgumentParser(); ap.add_argument("--layers", default=""); ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    global LAYERS
    if a.layers:
        LAYERS = [l for l in LAYERS if l[0] in a.layers.split(",")]
    lib = _lib.load(); ctx = _lib.Context.get(0)
    n = 16
    s = torch.cuda.current_stream().cuda_stream
    print(f"{'layer':6s} {'S':>4s} {'cin':>4s} {'cout':>4s} {'direct ms':>10s} {'TF':>7s} {'wino ms':>9s} {'TF':>7s} {'speedup':>8s} {'maxdiff':>9s}")
    for name, S, ci, co in LAYERS:
        x = torch.randn(n, S, S, ci, device="cuda"); k = torch.randn(3, 3, ci, co, device="cuda") * 0.1; b = torch.randn(co, device="cuda")
        ws = torch.empty(int(lib.unet_conv3x3_w_ws_floats(ci, co)), device="cuda")
        ys = []
        res = []
        for algo in (2, 3):
            y = torch.empty(n, S, S, co, device="cuda")
            for _ in range(min(20, a.reps)):
                ctx.check(lib.unet_conv3x3_fwd(ctx.handle, x.data_ptr(), k.data_ptr(), b.data_ptr(), y.data_ptr(), n, S, S, ci, co, 1, 0.0, 0, algo, ws.data_ptr(), s))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                lib.unet_conv3x3_fwd(ctx.handle, x.data_ptr(), k.data_ptr(), b.data_ptr(), y.data_ptr(), n, S, S, ci, co, 1, 0.0, 0, algo, ws.data_ptr(), s)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            res.append(ms); ys.append(y)
        fl = 2.0 * 9 * ci * co * n * S * S
        print(f"{name:6s} {S:4d} {ci:4d} {co:4d} {res[0]:10.3f} {fl / res[0] / 1e9:7.1f} {res[1]:9.3f} {fl / res[1] / 1e9:7.1f} {res[0] / res[1]:8.2f} {float((ys[0] - ys[1]).abs().max()):9.2e}")


if __name__ == "__main__":
    main()
