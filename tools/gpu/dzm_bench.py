"""Times the two gradients of the last conv3x3 at 512 x 512 x 16 from the fp32 tensor (unet_conv3x3_bwd_data / _bwd_weights) and from the head's {dz, mask} stream
(unet_conv3x3_bwd_data_dzm / _bwd_weights_dzm), back to back on one box.   python tools/gpu/dzm_bench.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from covidseg_amd import _lib

lib = _lib.load(); ctx = _lib.Context.get(0); H = ctx.handle
n, h, w, c = 16, 512, 512, 32
px = n * h * w
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.relu(torch.randn(n, h, w, c, device="cuda", generator=g))
dy = torch.randn(n, h, w, c, device="cuda", generator=g) * (torch.rand(n, h, w, c, device="cuda", generator=g) > 0.5) * 1e-6
k3 = torch.randn(3, 3, c, c, device="cuda", generator=g) * 0.08
kh = torch.randn(c, device="cuda", generator=g)
dzm = torch.empty(px, 2, dtype=torch.int32, device="cuda")
dzm[:, 0] = (torch.randn(px, device="cuda", generator=g) * 1e-6).view(torch.int32)
dzm[:, 1] = torch.randint(-2**31, 2**31 - 1, (px,), device="cuda", generator=g, dtype=torch.int64).to(torch.int32)
bits_in = torch.randint(-2**62, 2**62, (px * c // 64,), device="cuda", generator=g, dtype=torch.int64)
dx = torch.empty(n, h, w, c, device="cuda"); dw = torch.empty(3, 3, c, c, device="cuda"); db = torch.empty(c, device="cuda")
wws = torch.empty(int(lib.unet_conv3x3_w_ws_floats(c, c)) + 64, device="cuda")
wsb = int(lib.unet_conv3x3_bwd_weights_ws_bytes(n, h, w, c, c)); ws = torch.empty(wsb // 4 + 16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
calls = {
    "dgrad tensor": lambda: lib.unet_conv3x3_bwd_data(H, dy.data_ptr(), k3.data_ptr(), bits_in.data_ptr(), 9, 0.0, 0, dx.data_ptr(), wws.data_ptr(), n, h, w, c, c, 0, s),
    "dgrad stream": lambda: lib.unet_conv3x3_bwd_data_dzm(H, dzm.data_ptr(), k3.data_ptr(), kh.data_ptr(), bits_in.data_ptr(), dx.data_ptr(), wws.data_ptr(), n, h, w, c, s),
    "wgrad tensor": lambda: lib.unet_conv3x3_bwd_weights(H, x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, n, h, w, c, c, 0, s),
    "wgrad stream": lambda: lib.unet_conv3x3_bwd_weights_dzm(H, x.data_ptr(), dzm.data_ptr(), kh.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), wsb, n, h, w, c, s),
}
for rnd in range(3):
    for name, f in calls.items():
        for _ in range(5):
            ctx.check(f(), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f"round {rnd} {name}: {e0.elapsed_time(e1) / 30:.4f} ms (incl. the weight image / slab reduction launches)")
