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


# ---- per-element error bound of the fp16-split (h2) kernels -------------------------------------------------------------------------------
# An h2 product is xh wh + xh wm + xm wh with h = RN_f16(v 2^e), m = RN_f16(v 2^e - h): the dropped xm wm term and the two representation residuals are each
# <= 2^-22 |x w|, so an output y = sum_k x_k w_k carries |err| <= ~3 * 2^-22 * sum_k |x_k w_k| (+ the fp32 accumulation every fp32 kernel has).  The block
# scaling adds an ABSOLUTE floor: an operand element keeps 2^-36 of the largest magnitude its workgroup tile (activations / gradients) or its layer
# (weights) has seen.  The op tests assert, PER OUTPUT ELEMENT (not an L2 norm over the tensor),
#       |err| <= 4 * 2^-22 * A1 + floor * 2^-36 * A2,     A1 = sum_k |x_k| |w_k|,  A2 = max|x| sum_k |w_k| + max|w| sum_k |x_k|
# with floor = 0 wherever the operands of a tile are within 2^14 of each other (every ordinary tensor), floor = 1 in the out-of-domain cases.
EPS_SPLIT = 4.0 * 2.0 ** -22
EPS_FLOOR = 2.0 ** -36


def elem_ratio(got, want, a1, a2=None, floor=0.0):
    """max over elements of |got - want| / (EPS_SPLIT * a1 + floor * EPS_FLOOR * a2): <= 1 means the bound holds everywhere.  Elements whose bound is 0
    (no contributing products at all) must be exact."""
    got = np.asarray(got, np.float64); want = np.asarray(want, np.float64)
    bound = EPS_SPLIT * np.asarray(a1, np.float64)
    if a2 is not None and floor:
        bound = bound + floor * EPS_FLOOR * np.asarray(a2, np.float64)
    err = np.abs(got - want)
    tiny = 1e-300
    r = err / (bound + tiny)
    r[(bound == 0) & (err == 0)] = 0.0
    return float(r.max())


def conv_abs_sums(x, k, dy, with_floor=True):
    """A1 / A2 arrays (float64 numpy) of conv3x3 forward (y), data gradient (dx) and weight gradient (dw) for the bound above, from |x|, |k|, |dy|."""
    import torch
    import torch.nn.functional as F
    xa = torch.tensor(np.abs(x), dtype=torch.float64).permute(0, 3, 1, 2); ka = torch.tensor(np.abs(k), dtype=torch.float64).permute(3, 2, 0, 1)      # NCHW / OIHW
    da = torch.tensor(np.abs(dy), dtype=torch.float64).permute(0, 3, 1, 2)
    ci, co = k.shape[2], k.shape[3]
    xmax, kmax, dmax = float(np.abs(x).max()), float(np.abs(k).max()), float(np.abs(dy).max())
    one_k = torch.ones_like(ka); one_x = torch.ones_like(xa); one_d = torch.ones_like(da)
    out = {}
    out["y_a1"] = F.conv2d(xa, ka, padding=1).permute(0, 2, 3, 1).numpy()
    if with_floor:
        out["y_a2"] = (xmax * F.conv2d(one_x, ka, padding=1) + kmax * F.conv2d(xa, one_k, padding=1)).permute(0, 2, 3, 1).numpy()
    kt = ka.flip(2, 3).transpose(0, 1)                                   # data gradient = conv with the flipped / transposed kernel
    out["dx_a1"] = F.conv2d(da, kt, padding=1).permute(0, 2, 3, 1).numpy()
    if with_floor:
        out["dx_a2"] = (dmax * F.conv2d(one_d, kt, padding=1) + kmax * F.conv2d(da, torch.ones_like(kt), padding=1)).permute(0, 2, 3, 1).numpy()
    # weight gradient dw[a,b,c,o] = sum_p x[p + (a,b)] dy[p]: correlation of |x| with |dy| over pixels
    xp = F.pad(xa, (1, 1, 1, 1))
    n, _, h, w = xa.shape
    a1 = np.zeros((3, 3, ci, co)); sx = np.zeros((3, 3, ci))
    for a in range(3):
        for b in range(3):
            win = xp[:, :, a:a + h, b:b + w]
            a1[a, b] = torch.einsum("nchw,nohw->co", win, da).numpy()
            sx[a, b] = win.sum((0, 2, 3)).numpy()
    sd = da.sum((0, 2, 3)).numpy()
    out["dw_a1"] = a1
    out["dw_a2"] = xmax * sd[None, None, None, :] + dmax * sx[..., None]
    return out
