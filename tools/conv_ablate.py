"""Timing ablations of the MFMA conv kernel (UNET_CONV_ABL is read once per process -> run once per value)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from covidseg_amd import _lib
lib = _lib.load(); ctx = _lib.Context.get(0)
shapes = [("c1b", 16, 512, 512, 32, 32), ("c2b", 16, 256, 256, 64, 64), ("c3b", 16, 128, 128, 128, 128), ("c4b", 16, 64, 64, 256, 256),
          ("c5b", 16, 32, 32, 512, 512), ("c6a", 16, 64, 64, 512, 256), ("c9a", 16, 512, 512, 64, 32)]
s = torch.cuda.current_stream().cuda_stream
out = []
for name, n, h, w, ci, co in shapes:
    x = torch.randn(n, h, w, ci, device="cuda"); k = torch.randn(3, 3, ci, co, device="cuda") * 0.05; b = torch.zeros(co, device="cuda"); y = torch.empty(n, h, w, co, device="cuda")
    for _ in range(2):
        ctx.check(lib.unet_conv3x3_fwd(ctx.handle, x.data_ptr(), k.data_ptr(), b.data_ptr(), y.data_ptr(), n, h, w, ci, co, 1, 0.0, 0, 0, None, s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.unet_conv3x3_fwd(ctx.handle, x.data_ptr(), k.data_ptr(), b.data_ptr(), y.data_ptr(), n, h, w, ci, co, 1, 0.0, 0, 0, None, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    out.append(f"{name}:{2*9*ci*co*n*h*w/ms/1e9:6.1f}TF")
print(f"ABL={os.environ.get('UNET_CONV_ABL','0')} PF={os.environ.get('UNET_CONV_PF','1')}  " + "  ".join(out))
