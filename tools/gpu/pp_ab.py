#!/usr/bin/env python
"""Same-process A/B of the two conv3x3 schedules (context option CONV_PP 0 / 1) on ONE box: (a) the op alone, launches back to back, hipEvents around 20 of them;
(b) per-op profile of the training step (tools/profile_ops.py) with either option set.   python tools/gpu/pp_ab.py [--ops-only]"""
import os, sys, subprocess, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from covidseg_amd import _lib


def time_op(ctx, lib, fn, reps=20, rounds=5):
    best = []
    for _ in range(rounds):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ctx.check(fn(), "op")
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps)
    return float(np.median(best))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--ops-only", action="store_true"); ap.add_argument("--batch", type=int, default=16)
    a = ap.parse_args()
    lib = _lib.load()
    dev = torch.cuda.current_device()
    ctxs = {v: _lib.Context.get(dev, {"conv_pp": v}) for v in (0, 1)}
    n, h, w, c = a.batch, 512, 512, 32
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randn((n, h, w, c), device="cuda", generator=g); k = torch.randn((3, 3, c, c), device="cuda", generator=g) * 0.1; b = torch.randn((c,), device="cuda", generator=g)
    dy = torch.randn((n, h, w, c), device="cuda", generator=g) * 1e-4
    bits = torch.randint(-2**62, 2**62, (n * h * w * c // 64,), dtype=torch.int64, device="cuda")
    y = torch.empty_like(x); wws = torch.empty(int(lib.unet_conv3x3_w_ws_floats(c, c)), device="cuda")
    # warm the chip
    for _ in range(30):
        ctxs[0].check(lib.unet_conv3x3_fwd(ctxs[0].handle, x.data_ptr(), k.data_ptr(), b.data_ptr(), y.data_ptr(), n, h, w, c, c, 1, 0.0, 0, 0, wws.data_ptr(), s), "warm")
    torch.cuda.synchronize()
    for rnd in range(3):
        for v in (0, 1):
            cx = ctxs[v]
            t_f = time_op(cx, lib, lambda: lib.unet_conv3x3_fwd(cx.handle, x.data_ptr(), k.data_ptr(), b.data_ptr(), y.data_ptr(), n, h, w, c, c, 1, 0.0, 0, 0, wws.data_ptr(), s))
            t_d = time_op(cx, lib, lambda: lib.unet_conv3x3_bwd_data(cx.handle, dy.data_ptr(), k.data_ptr(), None, 0, 0.0, 0, y.data_ptr(), wws.data_ptr(), n, h, w, c, c, 0, s))
            t_m = time_op(cx, lib, lambda: lib.unet_conv3x3_bwd_data(cx.handle, dy.data_ptr(), k.data_ptr(), bits.data_ptr(), 9, 0.0, 0, y.data_ptr(), wws.data_ptr(), n, h, w, c, c, 0, s))
            print(f"round {rnd} conv_pp={v}: fwd(+weights image) {t_f:.4f} ms   dgrad {t_d:.4f} ms   dgrad+bits {t_m:.4f} ms", flush=True)
    if a.ops_only:
        return
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for rnd in range(2):
        for v in (0, 1):
            out = subprocess.run([sys.executable, os.path.join(root, "tools", "profile_ops.py"), "--options", '{"conv_pp": %d}' % v], capture_output=True, text=True).stdout
            keep = [ln for ln in out.splitlines() if any(t in ln for t in ("conv3x3_fwd:c1b", "conv3x3_dgrad:c1b", "conv3x3_fwd_head:c9b", "conv3x3_dgrad:c9b", "sum of op"))]
            print(f"--- profile_ops round {rnd} conv_pp={v}"); print("\n".join(keep), flush=True)


if __name__ == "__main__":
    main()
