"""Warm vs cold-cache timing of the MFMA conv (flush = stream 2 GiB through the chip between launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from covidseg_amd import _lib
lib = _lib.load(); ctx = _lib.Context.get(0)
shapes = [("c1b", 16, 512, 512, 32, 32), ("c3b", 16, 128, 128, 128, 128), ("c5b", 16, 32, 32, 512, 512), ("c6a", 16, 64, 64, 512, 256)]
s = torch.cuda.current_stream().cuda_stream
junk = torch.empty(512 * 1024 * 1024, device="cuda")
for relu_like in (0, 1):
    out = []
    for name, n, h, w, ci, co in shapes:
        x = torch.randn(n, h, w, ci, device="cuda")
        if relu_like: x = torch.relu(x)
        k = torch.randn(3, 3, ci, co, device="cuda") * 0.05; b = torch.zeros(co, device="cuda"); y = torch.empty(n, h, w, co, device="cuda")
        res = []
        for flush in (0, 1):
            ts = []
            for _ in range(4):
                if flush: junk.add_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); lib.unet_conv3x3_fwd(ctx.handle, x.data_ptr(), k.data_ptr(), b.data_ptr(), y.data_ptr(), n, h, w, ci, co, 1, 0.0, 0, 0, None, s); e1.record()
                torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            ms = sorted(ts)[1]
            res.append(f"{2*9*ci*co*n*h*w/ms/1e9:6.1f}")
        out.append(f"{name}: warm {res[0]} cold {res[1]} TF")
    print(("relu-like x  " if relu_like else "randn x      ") + "  ".join(out))
