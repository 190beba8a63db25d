"""Helpers for the -m gpu tests: call the C ABI (include/unet_hip.h) with torch-allocated device buffers."""
import numpy as np
import torch

from covidseg_amd import _lib


class Ops:
    def __init__(self):
        self.lib = _lib.load()
        self.ctx = _lib.Context.get(torch.cuda.current_device())
        self.h = self.ctx.handle
        self._keep = []          # keep device inputs alive: the C ABI only sees raw pointers

    @property
    def s(self):
        return torch.cuda.current_stream().cuda_stream

    def d(self, a, dtype=np.float32):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype)).cuda()
        self._keep.append(t)
        if len(self._keep) > 256:
            torch.cuda.synchronize(); del self._keep[:128]
        return t

    def wws(self, cin, cout):
        """device scratch for transformed conv weights (unet_conv3x3_w_ws_floats)"""
        t = torch.empty(max(int(self.lib.unet_conv3x3_w_ws_floats(cin, cout)), 4), dtype=torch.float32, device="cuda")
        self._keep.append(t)
        return t.data_ptr()

    def z(self, *shape, dtype=torch.float32):
        return torch.zeros(shape, dtype=dtype, device="cuda")

    def ck(self, rc, what=""):
        self.ctx.check(rc, what)
        torch.cuda.synchronize()


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
