import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from oracle import unet_oracle as O
from covidseg_amd.engine import HipUNet
rng = np.random.default_rng(21)
wts = O.init_weights(seed=8)
x = rng.random((3, 64, 64, 1)).astype(np.float32); y = (rng.random((3, 64, 64, 1)) > 0.75).astype(np.float32)
tr = O.OracleTrainer({k: v.astype(np.float64) for k, v in wts.items()}, torch.float64)
ref=[tr.train_step(x,y) for _ in range(10)]
for name,opts in (("default",None),("head_fused 0",{"head_fused":0}),("pool_sums_fused 0",{"pool_sums_fused":0}),("bn_fuse_stats 0",{"bn_fuse_stats":0}),("all three 0",{"head_fused":0,"pool_sums_fused":0,"bn_fuse_stats":0}),("head_bwd_fused 0",{"head_bwd_fused":0}),("skip_raw 0",{"skip_raw":0}),("strict algo",None)):
    eng = HipUNet(64,64,1,dropout_rate=0.0,options=opts, conv_algo=2 if "strict" in name else 0); eng.set_weights(wts)
    d=[]
    for s in range(10):
        a=eng.train_batch(x,y).cpu().numpy(); d.append(max(abs(a[0]-ref[s][0]),abs(a[1]-ref[s][1])))
    print(name, " ".join(f"{v:.1e}" for v in d))
