#!/usr/bin/env python
"""Phase timeline of conv_pp_kernel from s_memtime stamps (measurement build: tools/build_variant.sh trace kernels_conv_pp.hip -DPP_TRACE=1 -> build/exp/libunet_trace.so).
    python tools/pp_timeline.py build/exp/libunet_trace.so [N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def main():
    libp = os.path.abspath(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    mode = sys.argv[3] if len(sys.argv) > 3 else "fwd"
    from covidseg_amd import _lib
    _lib.LIB_PATH = libp
    lib = _lib.load(); ctx = _lib.Context.get(0, {"conv_pp": 1})
    dbg = ctypes.CDLL(libp).unet_debug_pp_trace
    dbg.argtypes = [ctypes.c_void_p, ctypes.c_int]; dbg.restype = ctypes.c_int
    h = w = 512; c = 32
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(n, h, w, c, device="cuda", generator=g).relu_(); k = torch.randn(3, 3, c, c, device="cuda", generator=g) * 0.1; b = torch.randn(c, device="cuda", generator=g)
    y = torch.empty_like(x); ws = torch.empty(int(lib.unet_conv3x3_w_ws_floats(c, c)), device="cuda"); s = torch.cuda.current_stream().cuda_stream
    bits = torch.randint(-2**62, 2**62, (n * h * w * c // 64,), dtype=torch.int64, device="cuda")
    def run():
        if mode == "dgrad": return lib.unet_conv3x3_bwd_data(ctx.handle, x.data_ptr(), k.data_ptr(), bits.data_ptr(), 9, 0.0, 0, y.data_ptr(), ws.data_ptr(), n, h, w, c, c, 0, s)
        return lib.unet_conv3x3_fwd(ctx.handle, x.data_ptr(), k.data_ptr(), b.data_ptr(), y.data_ptr(), n, h, w, c, c, 1, 0.0, 0, 0, ws.data_ptr(), s)
    for _ in range(200): ctx.check(run(), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"n={n} 512x512 32->32 {mode}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (incl. weight image prep; stamps cost ~10 %)")
    buf = np.zeros(256 * 2 * 40 * 12, np.uint64)
    assert dbg(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(256, 2, 40, 12).astype(np.int64)
    iters = (16 * 64 * n // 8 + 63) // 64
    names = ["P1 wait for the patch + max", "P1 barrier wait", "P2 read max, split -> planes", "P2 issue next loads", "P2 epilogue of the previous tile", "P2 barrier wait",
             "P3 first 18 MFMAs", "P3 barrier wait", "P4 the other MFMAs", "P4 barrier wait"]
    for hf in (0, 1):
        tt = t[:, hf, 2:iters - 1, :]                          # steady-state iterations
        print(f"half {hf}: iteration {np.median(tt[:, :, 10] - tt[:, :, 0]):.0f} cycles (median)")
        for i, nm in enumerate(names):
            d = (tt[:, :, i + 1] - tt[:, :, i]).reshape(-1)
            print(f"   {nm:36s} median {np.median(d):7.0f}  p10 {np.percentile(d, 10):7.0f}  p90 {np.percentile(d, 90):7.0f}")
    print(f"per block lifetimes median {np.median(t[:, 0, iters, 10] - t[:, 0, 0, 0]):.0f} cycles (stamps of different XCDs are not on one time base: no chip-wide span)")


if __name__ == "__main__":
    main()
