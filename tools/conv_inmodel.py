"""Why is a conv slower inside the training step than standalone?  Time unet_conv3x3_fwd on the model's own
buffers (weights in the flat param buffer, activations in the workspace) vs freshly allocated copies."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from covidseg_amd import _lib, weights as W
from covidseg_amd.data import synthetic_ct
from covidseg_amd.engine import HipUNet
lib = _lib.load(); ctx = _lib.Context.get(0)
B, S = 16, 512
xs, ys = synthetic_ct(4, S, 0)
x = torch.from_numpy(np.concatenate([xs] * 4)).cuda(); y = torch.from_numpy(np.concatenate([ys] * 4)).cuda()
eng = HipUNet(S, S, 1); eng.set_weights(W.init_weights(0))
for _ in range(2): eng.train_batch(x, y)
torch.cuda.synchronize()
plan = eng._plan(B); s = torch.cuda.current_stream().cuda_stream
def tap(name):
    ptr, ld, nn, hh, ww, cc = _lib.vp(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    lib.unet_model_tap(plan["m"], name.encode(), 0, C.byref(ptr), C.byref(ld), C.byref(nn), C.byref(hh), C.byref(ww), C.byref(cc))
    return ptr.value, nn.value, hh.value, ww.value, cc.value
def timeit(xp, kp, bp, yp, n, h, w, ci, co):
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.unet_conv3x3_fwd(ctx.handle, xp, kp, bp, yp, n, h, w, ci, co, 1, 0.0, 0, 0, None, s); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[2]
for lname, inp in (("c5b", "c5a"), ("c6a", "bn6"), ("c4b", "c4a"), ("c3b", "c3a")):
    xp, n, h, w, ci = tap(inp); yp, _, _, _, co = tap(lname)
    st, off, cnt, shape = eng._tinfo[lname + "/kernel"]; kp = eng.params.data_ptr() + 4 * off
    st, offb, cntb, _ = eng._tinfo[lname + "/bias"]; bp = eng.params.data_ptr() + 4 * offb
    fl = 2 * 9 * ci * co * n * h * w
    t_model = timeit(xp, kp, bp, yp, n, h, w, ci, co)
    # fresh copies of the SAME data
    xin = eng.tap(B, inp); xc = torch.from_numpy(xin).cuda(); kc = eng.params[off:off + cnt].clone(); bc = eng.params[offb:offb + cntb].clone()
    yc = torch.empty(n, h, w, co, device="cuda")
    t_copy = timeit(xc.data_ptr(), kc.data_ptr(), bc.data_ptr(), yc.data_ptr(), n, h, w, ci, co)
    t_kcopy = timeit(xp, kc.data_ptr(), bc.data_ptr(), yp, n, h, w, ci, co)
    xr = torch.randn(n, h, w, ci, device="cuda"); kr = torch.randn(3, 3, ci, co, device="cuda") * 0.05
    t_rand = timeit(xr.data_ptr(), kr.data_ptr(), bc.data_ptr(), yc.data_ptr(), n, h, w, ci, co)
    t_randx = timeit(xr.data_ptr(), kc.data_ptr(), bc.data_ptr(), yc.data_ptr(), n, h, w, ci, co)
    print(f"{lname}: model-buffers {fl/t_model/1e9:6.1f} TF | same data fresh buffers {fl/t_copy/1e9:6.1f} | model x + fresh w {fl/t_kcopy/1e9:6.1f} | randn x, model w values {fl/t_randx/1e9:6.1f} | randn x, randn w {fl/t_rand/1e9:6.1f}"
          f"   (w offset mod 128B = {(kp % 128)}, x zero fraction {float((xin == 0).mean()):.2f})")
