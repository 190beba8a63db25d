"""Debug aid: two ranks on one GPU (gloo) vs the single-process full batch, per-tensor gradient differences, for a chosen arch / option set.
    python tools/debug_dp.py classifier '{"bn_fold": 0}'"""
import json, os, socket, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, wfile, x, y, out, arch, opts):
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from covidseg_amd.engine import HipUNet
    eng = HipUNet(x.shape[1], x.shape[2], 1, device=0, process_group=dist.group.WORLD, dropout_rate=0.0, arch=arch, options=opts or None)
    eng.set_weights(dict(np.load(wfile)))
    n = x.shape[0] // world
    loss = eng.forward_backward(x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]).cpu().numpy()
    g = eng.get_grads()
    taps = {}
    for nm in os.environ.get("DBG_TAPS", "").split(","):
        if nm:
            gr = nm.startswith("d:")
            taps["tap/" + nm] = eng.tap(n, nm[2:] if gr else nm, grad=gr)
    np.savez(out + f".r{rank}.npz", loss=loss, **taps)
    if rank == 0:
        np.savez(out, **g)
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    from covidseg_amd import weights as W
    from covidseg_amd.data import synthetic_classification, synthetic_ct
    from covidseg_amd.engine import HipUNet
    arch = sys.argv[1] if len(sys.argv) > 1 else "classifier"
    opts = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
    if arch == "classifier":
        x, y = synthetic_classification(8, 32, seed=5); y = y.astype(np.float32)
    else:
        x, y = synthetic_ct(4, 32, seed=5)
    wts = W.init_weights(4, 1, arch, (32, 32))
    tmp = tempfile.mkdtemp()
    wfile = os.path.join(tmp, "w.npz"); np.savez(wfile, **wts); out = os.path.join(tmp, "dp.npz")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, wfile, x, y, out, arch, opts), nprocs=2, join=True)
    got = np.load(out)
    eng = HipUNet(32, 32, 1, dropout_rate=0.0, arch=arch, options=opts or None); eng.set_weights(wts)
    loss = eng.forward_backward(x, y).cpu().numpy()
    r0, r1 = np.load(out + ".r0.npz"), np.load(out + ".r1.npz")
    print(arch, opts, "loss single", loss, "dp", r0["loss"], r1["loss"])
    for nm in os.environ.get("DBG_TAPS", "").split(","):
        if nm:
            gr = nm.startswith("d:")
            ref = eng.tap(x.shape[0], nm[2:] if gr else nm, grad=gr)
            dp = np.concatenate([r0["tap/" + nm], r1["tap/" + nm]], 0)
            if not gr:
                print(f"  tap {nm:12s} sign flips {int(((dp > 0) != (ref > 0)).sum())} of {dp.size}; smallest |value| at a flip {float(np.abs(np.where((dp > 0) != (ref > 0), np.maximum(np.abs(dp), np.abs(ref)), np.inf)).min()):.2e}")
            print(f"  tap {nm:12s} rel {np.linalg.norm(dp - ref) / (np.linalg.norm(ref) + 1e-30):.2e}  (DP gradients are per-rank shares: x world for d: taps)" if gr else f"  tap {nm:12s} rel {np.linalg.norm(dp - ref) / (np.linalg.norm(ref) + 1e-30):.2e}")
    for k, v in eng.get_grads().items():
        rel = np.linalg.norm(got[k] - v) / (np.linalg.norm(v) + 1e-30)
        print(f"  {k:16s} rel {rel:.2e}")
        if rel > 1e-4:
            d = np.abs(got[k] - v).reshape(-1); o = np.argsort(-d)[:4]
            print("      worst", [(int(i), float(got[k].reshape(-1)[i]), float(v.reshape(-1)[i])) for i in o], "ratio got/ref median", float(np.median(got[k].reshape(-1) / (v.reshape(-1) + 1e-30))))
