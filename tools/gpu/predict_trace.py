"""batch-1 predict at 512 x 512 (T1:1137): N synchronised calls on unchanged weights -- run under `rocprofv3 --kernel-trace --stats` for the per-kernel table of the
inference program, or alone for the latency (median / min of the synchronised calls, and the back-to-back rate without a sync per call)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from covidseg_amd.data import synthetic_ct
from covidseg_amd.engine import HipUNet
from covidseg_amd import weights as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
x, _ = synthetic_ct(bs, 512, seed=0)
x = torch.from_numpy(x).cuda()
eng = HipUNet(512, 512, 1, device=0)
eng.set_weights(W.init_weights(0, 1, "unet", (512, 512)))
for _ in range(5):
    eng.predict_batch(x)
torch.cuda.synchronize()
lat = []
for _ in range(n):
    t = time.perf_counter(); eng.predict_batch(x); torch.cuda.synchronize(); lat.append((time.perf_counter() - t) * 1e3)
t = time.perf_counter()
for _ in range(n):
    eng.predict_batch(x)
torch.cuda.synchronize()
b2b = (time.perf_counter() - t) * 1e3 / n
print(f"predict batch {bs}: median {sorted(lat)[n // 2]:.3f} ms, min {min(lat):.3f} ms, back-to-back {b2b:.3f} ms/call")
