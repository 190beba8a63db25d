"""-m gpu: the data-parallel path of the HIP engine with world_size 2 on ONE GPU (two processes, gloo
process group; the engine stages gloo reductions through the host).  With sync-BN + batch-global Dice +
SUM-reduced gradients, two ranks holding half the batch each must reproduce the single-process full-batch
step (the property the 8-GPU RCCL run relies on)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, wfile, x, y, out, arch="unet", dtype="fp32"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from covidseg_amd.engine import HipUNet
    wts = dict(np.load(wfile))
    eng = HipUNet(x.shape[1], x.shape[2], 1, device=0, process_group=dist.group.WORLD, dropout_rate=0.0, arch=arch, dtype=dtype)
    eng.set_weights(wts)
    n = x.shape[0] // world
    xs, ys = x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]
    eng.forward_backward(xs, ys)
    grads = eng.get_grads()                                            # SUM-reduced over ranks with the global normaliser
    losses = [eng.train_batch(xs, ys).cpu().numpy() for _ in range(2)]
    p, ld = eng.predict_batch(xs, ys)
    sums = eng.threshold_sums(p, ys, [0.3, 0.5]).cpu().numpy() if arch != "classifier" else np.zeros(1)
    if rank == 0:
        np.savez(out, losses=np.array(losses), ld=ld.cpu().numpy(), sums=sums, **{"w/" + k: v for k, v in eng.get_weights().items()},
                 **{"g/" + k: v for k, v in grads.items()})
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("arch", ["unet", "unetpp", "classifier"])
def test_two_ranks_equal_single_process_full_batch(tmp_path, arch):
    import torch.multiprocessing as mp
    from covidseg_amd import weights as W
    from covidseg_amd.data import synthetic_classification, synthetic_ct
    from covidseg_amd.engine import HipUNet
    if arch == "classifier":
        x, y = synthetic_classification(8, 32, seed=5); y = y.astype(np.float32)
    else:
        x, y = synthetic_ct(4, 32, seed=5)
    wts = W.init_weights(4, 1, arch, (32, 32))
    wfile = str(tmp_path / "w.npz"); np.savez(wfile, **wts)
    out = str(tmp_path / "dp.npz")
    mp.get_context("spawn")
    mp.spawn(_worker, args=(2, _free_port(), wfile, x, y, out, arch), nprocs=2, join=True)
    got = np.load(out)
    eng = HipUNet(32, 32, 1, dropout_rate=0.0, arch=arch); eng.set_weights(wts)
    eng.forward_backward(x, y)
    for k, v in eng.get_grads().items():                              # the reduced gradient of 2 half batches == full-batch gradient
        a = got["g/" + k]
        assert np.linalg.norm(a - v) <= 2e-4 * np.linalg.norm(v) + 2e-8 * np.sqrt(v.size), k
    ref_losses = np.array([eng.train_batch(x, y).cpu().numpy() for _ in range(2)])
    p, ld = eng.predict_batch(x, y)
    sums = eng.threshold_sums(p, y, [0.3, 0.5]).cpu().numpy() if arch != "classifier" else np.zeros(1)
    assert np.abs(got["losses"] - ref_losses).max() < 2e-5            # batch-global loss / dice on every rank
    # after 2 Adam steps (sign-like updates while v is tiny) fp32 summation-order noise of the gradients shows up at ~4e-5 in
    # the classifier's small-batch loss; the weights themselves are compared below
    assert np.abs(got["ld"] - ld.cpu().numpy()).max() < (1e-4 if arch == "classifier" else 2e-5) and np.allclose(got["sums"], sums, rtol=1e-5)
    wref = eng.get_weights()
    for k, v in wref.items():                                         # identical replicas after 2 optimizer steps
        a = got["w/" + k]
        # (Adam turns round-off in a near-zero gradient into an O(lr) step: absolute term for the classifier's dead / BN-shadowed units)
        assert np.linalg.norm(a - v) <= 2e-4 * np.linalg.norm(v) + (1e-4 if arch == "classifier" else 1e-6) * np.sqrt(v.size), k


def test_two_ranks_bf16_storage_match_full_batch(tmp_path):
    """bf16-storage mode under data parallelism: the collectives are the fp32 path's (fp64 BN / Dice sums, fp32 gradient buckets), so
    two half batches are the same computation as the full batch up to the summation order of the BN statistics and weight gradients."""
    import torch.multiprocessing as mp
    from covidseg_amd import weights as W
    from covidseg_amd.data import synthetic_ct
    from covidseg_amd.engine import HipUNet
    x, y = synthetic_ct(4, 32, seed=5)
    wts = W.init_weights(4, 1, "unet", (32, 32))
    wfile = str(tmp_path / "w.npz"); np.savez(wfile, **wts)
    out = str(tmp_path / "dp.npz")
    mp.spawn(_worker, args=(2, _free_port(), wfile, x, y, out, "unet", "bf16"), nprocs=2, join=True)
    got = np.load(out)
    eng = HipUNet(32, 32, 1, dropout_rate=0.0, dtype="bf16"); eng.set_weights(wts)
    eng.forward_backward(x, y)
    # measured: the two evaluations decorrelate to the bf16 noise floor exactly like engine-vs-oracle (test_gpu_bf16_model.py: one
    # flipped rounding feeds 20 more layers and flips ReLU masks), 0.13 relative at c1a/kernel.  What data parallelism must still
    # guarantee is checked: direction, and the SUM-with-global-normaliser scale (a missing all-reduce halves the norm).
    for k, v in eng.get_grads().items():
        if k.startswith("u") and k.endswith("/bias"):
            continue                                                  # true gradient 0 (ConvT bias in front of a BatchNorm)
        a = got["g/" + k].astype(np.float64).ravel(); v = v.astype(np.float64).ravel()
        na, nv = np.linalg.norm(a), np.linalg.norm(v)
        assert a @ v / (na * nv + 1e-300) > 0.9 and 0.8 < na / (nv + 1e-300) < 1.25, k
    ref_losses = np.array([eng.train_batch(x, y).cpu().numpy() for _ in range(2)])
    assert np.abs(got["losses"] - ref_losses).max() < 5e-3            # batch-global loss / Dice on every rank
