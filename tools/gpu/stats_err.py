"""How accurate are the BatchNorm statistics a conv epilogue leaves (fp32 per-workgroup partials) against float64 sums of the stored tensor?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np, torch
from gpu_util import Ops
ops = Ops()
rng = np.random.default_rng(0)
for (n, h, w, ci, co) in ((3, 64, 64, 32, 32), (16, 512, 512, 32, 32), (3, 32, 32, 64, 64), (16, 256, 256, 64, 64)):
    x = np.maximum(rng.standard_normal((n, h, w, ci)), 0).astype(np.float32); k = (rng.standard_normal((3, 3, ci, co)) * (2.0 / (9 * ci)) ** 0.5).astype(np.float32)
    b = (rng.standard_normal(co) * 0.3).astype(np.float32)
    xd, kd, bd = ops.d(x), ops.d(k), ops.d(b); pixels = n * h * w
    y = ops.z(n, h, w, co); fused = ops.z(2 * co, dtype=torch.float64); plain = ops.z(2 * co, dtype=torch.float64)
    ops.ck(ops.lib.unet_request_bn_stats(ops.h, co), "arm")
    ops.ck(ops.lib.unet_conv3x3_fwd(ops.h, xd.data_ptr(), kd.data_ptr(), bd.data_ptr(), y.data_ptr(), n, h, w, ci, co, 1, 0.0, 0, 0, ops.wws(ci, co), ops.s), "conv")
    ops.ck(ops.lib.unet_bn_stats(ops.h, y.data_ptr(), co, fused.data_ptr(), pixels, co, ops.s), "fold")
    ops.ck(ops.lib.unet_bn_stats(ops.h, y.data_ptr(), co, plain.data_ptr(), pixels, co, ops.s), "pass")
    y64 = y.cpu().numpy().astype(np.float64).reshape(-1, co)
    want = np.concatenate([y64.sum(0), (y64 * y64).sum(0)])
    def var(s): m = s[:co] / pixels; return s[co:] / pixels - m * m
    for nm, t in (("epilogue", fused.cpu().numpy()), ("pass", plain.cpu().numpy())):
        print(f"{(n,h,w,ci,co)} {nm:9s} max rel err: sum {np.abs(t[:co]/want[:co]-1).max():.1e}  sumsq {np.abs(t[co:]/want[co:]-1).max():.1e}  var {np.abs(var(t)/var(want)-1).max():.1e}")
