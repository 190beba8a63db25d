"""Does a producer -> consumer pair of 512 x 512 launches run faster when the batch goes through the pair in groups that fit the 256-MB memory-side cache?
   python tools/gpu/subbatch_probe.py [group]     (op-level ABI; conv A: cin -> 32, conv B: 32 -> 32 on A's output; whole batch of 16 vs groups of `group` images)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from covidseg_amd import _lib
lib = _lib.load(); ctx = _lib.Context.get(0)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N, S = 16, 512
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
def run_pair(ci, reps=20):
    x = torch.randn(N, S, S, ci, device="cuda", generator=g).relu_()
    ka = torch.randn(3, 3, ci, 32, device="cuda", generator=g) * (2.0 / (9 * ci)) ** 0.5; kb = torch.randn(3, 3, 32, 32, device="cuda", generator=g) * (2.0 / 288) ** 0.5
    b = torch.zeros(32, device="cuda"); ya = torch.empty(N, S, S, 32, device="cuda"); yb = torch.empty(N, S, S, 32, device="cuda")
    wa = torch.empty(max(int(lib.unet_conv3x3_w_ws_floats(ci, 32)), 4), device="cuda"); wb = torch.empty(max(int(lib.unet_conv3x3_w_ws_floats(32, 32)), 4), device="cuda")
    junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    def conv(xp, kp, yp, n, c, ws):
        ctx.check(lib.unet_conv3x3_fwd(ctx.handle, xp, kp.data_ptr(), b.data_ptr(), yp, n, S, S, c, 32, 1, 0.0, 0, 0, ws.data_ptr(), s), "conv")
    def whole():
        conv(x.data_ptr(), ka, ya.data_ptr(), N, ci, wa); conv(ya.data_ptr(), kb, yb.data_ptr(), N, 32, wb)
    def grouped():
        for i in range(0, N, G):
            conv(x.data_ptr() + i * S * S * ci * 4, ka, ya.data_ptr() + i * S * S * 32 * 4, G, ci, wa)
            conv(ya.data_ptr() + i * S * S * 32 * 4, kb, yb.data_ptr() + i * S * S * 32 * 4, G, 32, wb)
    out = {}
    for name, f in (("whole", whole), ("grouped", grouped), ("whole2", whole), ("grouped2", grouped)):
        for _ in range(5): f()
        ts = []
        for _ in range(reps):
            junk.zero_()                      # (the caches start each repetition holding something else)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        out[name] = sorted(ts)[len(ts) // 2]
    ref = yb.clone(); whole(); torch.cuda.synchronize()
    assert torch.equal(ref, yb)
    return out
for ci in (32, 64):
    r = run_pair(ci)
    print(f"conv {ci}->32 then 32->32 at {N} x {S} x {S}: whole batch {r['whole']:.3f} / {r['whole2']:.3f} ms, groups of {G} images {r['grouped']:.3f} / {r['grouped2']:.3f} ms (incl. the weight-image launches of every call)")
