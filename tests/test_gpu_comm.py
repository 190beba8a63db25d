"""-m gpu: the device-side small all-reduce (include/unet_hip.h unet_comm_*, csrc/comm.hip) through the C ABI.

The GPU box has ONE device: two (and four) ranks are as many processes on it, each mapping the others' receive areas through HIP IPC -- the same code path the ranks of an
8-GPU node take (there the mapped area lives behind an xGMI link).  Checked: sums exact in fp64 and bit-identical on both ranks, vectors longer than one
launch, many back-to-back calls of changing length (the two alternating areas), a missing peer ends the kernel with the error word set instead of hanging."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make(lib, ctx, rank, world, gather):
    comm = C.c_void_p()
    h = (C.c_ubyte * 64)()
    ctx.check(lib.unet_comm_create(ctx.handle, rank, world, C.byref(comm), h), "comm_create")
    ctx.check(lib.unet_comm_connect(comm, b"".join(gather(bytes(h)))), "comm_connect")
    return comm


def test_world_one_is_the_identity_and_long_vectors_are_chunked():
    from covidseg_amd import _lib
    lib, ctx = _lib.load(), _lib.Context.get(0)
    comm = _make(lib, ctx, 0, 1, lambda h: [h])
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(3)
    for n in (1, 7, 256, 2048, 2049, 5000):
        a = torch.randn(n, dtype=torch.float64, generator=g)
        d = a.cuda()
        ctx.check(lib.unet_comm_allreduce_f64(comm, d.data_ptr(), n, st), "allreduce")
        assert torch.equal(d.cpu(), a), n
    err = C.c_int32(-1)
    ctx.check(lib.unet_comm_status(comm, C.byref(err), st), "status")
    assert err.value == 0
    assert lib.unet_comm_allreduce_f64(comm, 0, 4, st) != 0 and lib.unet_comm_allreduce_f64(comm, d.data_ptr(), 0, st) != 0
    lib.unet_comm_destroy(comm)


def _worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from covidseg_amd import _lib
    lib, ctx = _lib.load(), _lib.Context.get(0)

    def gather(h):
        hs = [None] * world
        dist.all_gather_object(hs, h)
        return hs
    comm = _make(lib, ctx, rank, world, gather)
    st = torch.cuda.current_stream().cuda_stream
    results, expect = [], []
    counts = [1, 2048, 64, 1024, 3, 2048, 2048, 4100, 128] * 12          # 108 back-to-back calls: both areas reused many times, lengths changing under stale words
    for k, n in enumerate(counts):
        mine = torch.randn(n, dtype=torch.float64, generator=torch.Generator().manual_seed(1000 * k + rank))
        other = [torch.randn(n, dtype=torch.float64, generator=torch.Generator().manual_seed(1000 * k + r)) for r in range(world)]
        d = mine.cuda()
        ctx.check(lib.unet_comm_allreduce_f64(comm, d.data_ptr(), n, st), "allreduce")
        results.append(d); expect.append(sum(other[1:], other[0]))                                    # rank order 0, 1, ...: the kernel's order
    err = C.c_int32(-1)
    ctx.check(lib.unet_comm_status(comm, C.byref(err), st), "status")
    ok = err.value == 0 and all(torch.equal(r.cpu(), e) for r, e in zip(results, expect))
    # a peer that never shows up: rank 0 reduces alone with a 300 ms budget
    dist.barrier()
    timed_out = 0
    if rank == 0:
        lib.unet_comm_set_timeout_ms(comm, 300)
        d = torch.ones(8, dtype=torch.float64, device="cuda")
        ctx.check(lib.unet_comm_allreduce_f64(comm, d.data_ptr(), 8, st), "allreduce")
        ctx.check(lib.unet_comm_status(comm, C.byref(err), st), "status")
        timed_out = err.value
        assert bool(torch.isnan(d).all())                             # the rank-local values do not pass for sums
    dist.barrier()
    np.savez(out + f".{rank}.npz", ok=ok, timed_out=timed_out, last=results[-1].cpu().numpy())
    lib.unet_comm_destroy(comm)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_on_one_gpu(tmp_path, world):
    import torch.multiprocessing as mp
    out = str(tmp_path / "comm")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    res = [np.load(out + f".{r}.npz") for r in range(world)]
    assert all(bool(r["ok"]) for r in res)
    assert all(np.array_equal(res[0]["last"], r["last"]) for r in res[1:])          # the same bits on every rank
    assert int(res[0]["timed_out"]) == 2                              # 1 + the first rank that was missing


def _engine_worker(rank, world, port, out, mode):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from covidseg_amd import weights as W
    from covidseg_amd.data import synthetic_ct
    from covidseg_amd.engine import HipUNet
    x, y = synthetic_ct(4, 32, seed=5)
    eng = HipUNet(32, 32, 1, device=0, process_group=dist.group.WORLD, dropout_rate=0.0, small_allreduce=mode, options={"deterministic": 1})
    assert (eng._comm is not None) == (mode == "device")
    eng.set_weights(W.init_weights(4, 1, "unet", (32, 32)))
    n = 2
    losses = [eng.train_batch(x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]).cpu().numpy() for _ in range(3)]
    assert eng.comm_status() == 0
    if rank == 0:
        np.savez(out, losses=np.array(losses), **eng.get_weights())
    eng.close()
    dist.barrier(); dist.destroy_process_group()


def test_engine_device_reductions_equal_the_collective_library_bit_for_bit(tmp_path):
    """the same 2-rank training steps with the sums reduced by comm.hip and by torch.distributed: both add rank 0 + rank 1 in fp64 -> identical bits"""
    import torch.multiprocessing as mp
    got = {}
    for mode in ("device", "rccl"):
        out = str(tmp_path / f"{mode}.npz")
        mp.spawn(_engine_worker, args=(2, _free_port(), out, mode), nprocs=2, join=True)
        got[mode] = dict(np.load(out))
    for k in got["device"]:
        assert np.array_equal(got["device"][k], got["rccl"][k]), k
