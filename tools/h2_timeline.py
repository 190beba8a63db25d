#!/usr/bin/env python
"""Phase timeline of conv_h2_kernel from s_memtime stamps (measurement build: `make -C <package>/csrc timeline` -> build/exp/libunet_exp5.so; the library
path is given explicitly, the product library has no such symbol).
    python tools/h2_timeline.py build/exp/libunet_exp5.so N H W CIN COUT [dgrad | convT | convT_dgrad]     (convT: H x W = the ConvT's input size)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    libp = sys.argv[1]; n, h, w, ci, co = map(int, sys.argv[2:7]); mode = sys.argv[7] if len(sys.argv) > 7 else "fwd"; dgrad = mode == "dgrad"
    from covidseg_amd import _lib
    _lib.LIB_PATH = os.path.abspath(libp)
    lib = _lib.load(); ctx = _lib.Context.get(0)
    dbg = ctypes.CDLL(os.path.abspath(libp)).unet_debug_h2_trace
    dbg.argtypes = [ctypes.c_void_p, ctypes.c_int]; dbg.restype = ctypes.c_int
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(n, h, w, ci, device="cuda", generator=g).relu_(); k = torch.randn(3, 3, ci, co, device="cuda", generator=g) * (2.0 / (9 * ci)) ** 0.5
    b = torch.randn(co, device="cuda", generator=g); y = torch.empty(n, h, w, co, device="cuda")
    dy = torch.randn(n, h, w, co, device="cuda", generator=g); dx = torch.empty(n, h, w, ci, device="cuda")
    ws = torch.empty(max(int(lib.unet_conv3x3_w_ws_floats(ci, co)), 4), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    if mode.startswith("convT"):
        kT = torch.randn(2, 2, co, ci, device="cuda", generator=g) * (1.0 / ci) ** 0.5
        u = torch.empty(n, 2 * h, 2 * w, 2 * co, device="cuda"); du = torch.randn(n, 2 * h, 2 * w, 2 * co, device="cuda", generator=g)
    def run():
        if mode == "convT": return lib.unet_convT2x2_fwd(ctx.handle, x.data_ptr(), kT.data_ptr(), b.data_ptr(), u.data_ptr(), 2 * co, n, h, w, ci, co, 0, s)
        if mode == "convT_dgrad": return lib.unet_convT2x2_bwd_data(ctx.handle, du.data_ptr(), 2 * co, kT.data_ptr(), None, dx.data_ptr(), n, h, w, ci, co, 0, s)
        if dgrad: return lib.unet_conv3x3_bwd_data(ctx.handle, dy.data_ptr(), k.data_ptr(), x.data_ptr(), 1, 0.0, 0, dx.data_ptr(), ws.data_ptr(), n, h, w, ci, co, 0, s)
        return lib.unet_conv3x3_fwd(ctx.handle, x.data_ptr(), k.data_ptr(), b.data_ptr(), y.data_ptr(), n, h, w, ci, co, 1, 0.0, 0, 0, ws.data_ptr(), s)
    for _ in range(300): ctx.check(run(), "conv")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"shape n={n} {h}x{w} {ci}->{co} {mode}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (incl. weight image prep)")
    nwg = min(16384, ((w + 31) // 32) * ((h + 7) // 8) * n * max(1, co // 64 if not dgrad else ci // 64))
    if mode == "convT": nwg = min(16384, ((w + 31) // 32) * ((h + 7) // 8) * n * max(1, 4 * co // 128))
    if mode == "convT_dgrad": nwg = min(16384, ((w + 31) // 32) * ((h + 7) // 8) * n * max(1, ci // 128))
    buf = np.zeros(16384 * 16, np.uint64)
    assert dbg(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(16384, 16)[:nwg].astype(np.int64)
    ok = t[:, 8] > t[:, 0]
    t = t[ok]
    print(f"{len(t)} workgroups traced")
    names = ["first loads land + amax + barrier", "stage chunk 0 (split, LDS stores, W slab wait) + barrier", "MFMA chunk 0 issue", "W issue + amax + barrier (wait for the other waves)",
             "stage chunk 1 + barrier", "chunks 2.. (loop remainder)", "last chunk MFMA issue", "epilogue (drain MFMAs, bias, stores issued)"]
    for i, nm in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print(f"  {nm:62s} median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f} cycles")
    for nm, a, b in (("  of the epilogue: barrier in front of it (slowest wave's last MFMAs)", 7, 11), ("  of the epilogue: coefficient reads, LDS transposes, stores issued", 11, 12), ("  of the epilogue: statistics barrier + atomics", 12, 8)):
        d = t[:, b] - t[:, a]
        print(f"  {nm:62s} median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f} cycles")
    tot = t[:, 8] - t[:, 0]
    print(f"  {'workgroup lifetime':62s} median {np.median(tot):8.0f}  p10 {np.percentile(tot, 10):8.0f}  p90 {np.percentile(tot, 90):8.0f} cycles")
    xcc = t[:, 10] & 0xF
    hw = t[:, 9]
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 20 + cu
    # (no per-XCC kernel span: s_memtime is a per-XCC counter -- stamps of different XCDs are not on one time base, and round 5's line compared unrelated numbers;
    #  the per-CU figures below only combine stamps of one CU)
    conc = []
    for kk in np.unique(key)[:400]:
        m = key == kk
        a, bb = t[m, 0], t[m, 8]
        span = bb.max() - a.min()
        conc.append((bb - a).sum() / max(span, 1))
    print(f"  CUs seen {len(np.unique(key))}; mean resident workgroups per CU over its busy span {np.mean(conc):.2f}; workgroups per CU {len(t) / len(np.unique(key)):.1f}")


if __name__ == "__main__":
    main()
