"""CPU, world_size 2, gloo: the data-parallel orchestration (covidseg_amd.dp.run_program) on a stand-in
op program with the same sync-point structure as the U-Net programs (batch-global statistics, global
loss sums, gradient buckets).  Checks that the 2-rank result equals the single-process full-batch result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from covidseg_amd import dp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class ToyProgram:
    """y = (x - mean_global) ; loss = mean_global(y^2 * w) ; grad_w, grad_b in two buckets.  ops:
    0 stats -> [sync kind0: sums]  1 normalise  2 loss sums -> [sync kind1]  3 grad bucket A -> [sync kind3]
    4 grad bucket B -> [sync kind3]"""

    def __init__(self, x, w, world):
        self.x, self.w, self.world = x, w, world
        self.sums = torch.zeros(2, dtype=torch.float64); self.lsum = torch.zeros(1, dtype=torch.float64)
        self.grads = torch.zeros(2 * w.numel(), dtype=torch.float32)
        self.log = []
        n = w.numel()
        self.sync = [(0, 0, "sums", 2), (2, 1, "lsum", 1), (3, 3, ("g", 0), n), (4, 3, ("g", n), n)]

    def run_range(self, b, e):
        for op in range(b, e):
            self.log.append(op)
            count = self.x.numel() * self.world
            if op == 0:
                self.sums[0] = self.x.double().sum(); self.sums[1] = self.x.numel()
            elif op == 1:
                self.y = self.x - (self.sums[0] / self.sums[1]).float()
            elif op == 2:
                self.lsum[0] = (self.y.double() ** 2 * self.w.double()).sum()
            elif op == 3:
                self.grads[:self.w.numel()] = (self.y ** 2).sum(0) / count          # d loss / d w (local part, global normaliser)
            elif op == 4:
                self.grads[self.w.numel():] = (2 * self.y * self.w).sum(0) / count

    def buf(self, h, count):
        if h == "sums": return self.sums
        if h == "lsum": return self.lsum
        return self.grads[h[1]:h[1] + count]


def _worker(rank, world, port, x_all, w, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = x_all.chunk(world)[rank]
    prog = ToyProgram(shard, w, world)
    pending = []
    dp.run_program(prog.run_range, 5, prog.sync,
                   lambda h, c: dist.all_reduce(prog.buf(h, c)),
                   lambda h, c: pending.append(dist.all_reduce(prog.buf(h, c), async_op=True)),
                   lambda: [p.wait() for p in pending])
    if rank == 0:
        torch.save({"grads": prog.grads, "loss": prog.lsum / (x_all.numel()), "log": prog.log}, out)
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_program_equals_full_batch(tmp_path):
    torch.manual_seed(0)
    x_all = torch.randn(8, 6); w = torch.rand(6) + 0.5
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), x_all, w, out), nprocs=2, join=True)
    got = torch.load(out)
    ref = ToyProgram(x_all, w, 1)
    dp.run_program(ref.run_range, 5, ref.sync, lambda h, c: None, lambda h, c: None, lambda: None)
    assert got["log"] == [0, 1, 2, 3, 4]
    assert torch.allclose(got["grads"], ref.grads, rtol=1e-6, atol=1e-7)
    assert torch.allclose(got["loss"], ref.lsum / x_all.numel(), rtol=1e-12)


def test_run_program_slicing_and_kind_filter():
    calls = []
    sync = [(1, 0, "a", 4), (1, 2, "b", 4), (3, 3, "g0", 10), (5, 3, "g1", 10)]
    dp.run_program(lambda b, e: calls.append(("run", b, e)), 8, sync, lambda h, c: calls.append(("small", h)),
                   lambda h, c: calls.append(("bucket", h)), lambda: calls.append(("finish",)))
    assert calls == [("run", 0, 2), ("small", "a"), ("small", "b"), ("run", 2, 4), ("bucket", "g0"), ("run", 4, 6), ("bucket", "g1"),
                     ("run", 6, 8), ("finish",)]
    calls.clear()                                                     # local-BN mode: only gradient buckets are reduced
    dp.run_program(lambda b, e: calls.append(("run", b, e)), 8, sync, lambda h, c: calls.append(("small", h)),
                   lambda h, c: calls.append(("bucket", h)), lambda: calls.append(("finish",)), enabled_kinds=(3,))
    assert calls == [("run", 0, 4), ("bucket", "g0"), ("run", 4, 6), ("bucket", "g1"), ("run", 6, 8), ("finish",)]


def test_deferred_small_reduction_waits_right_before_its_reader():
    """A small reduction whose first reader is not the next op (use_op > after_op + 1: the backward programs put an independent weight gradient behind the op
    that produces BatchNorm-backward sums) is started on a side stream and waited for just before the reader; inline ones (use_op == after_op + 1) and the
    gradient buckets keep their order; without the async hooks every small reduction stays inline."""
    sync = [(1, 2, "a", 4, 4),        # produced by op 1, read by op 4: ops 2, 3 may run beside it
            (2, 2, "b", 4, 3),        # produced by op 2, read by op 3: inline
            (5, 3, "g", 8, 8),        # bucket after op 5
            (6, 2, "c", 4, 9)]        # reader beyond the end of the program: waited for at the end
    log = []
    dp.run_program(lambda b, e: log.append(("run", b, e)), 8, sync, lambda h, c: log.append(("inline", h)), lambda h, c: log.append(("bucket", h)),
                   lambda: log.append(("finish",)), (0, 1, 2, 3), lambda h, c: (log.append(("async", h)), h)[1], lambda tok: log.append(("wait", tok)))
    assert log == [("run", 0, 2), ("async", "a"), ("run", 2, 3), ("inline", "b"), ("run", 3, 4), ("wait", "a"), ("run", 4, 6), ("bucket", "g"), ("run", 6, 7),
                   ("async", "c"), ("run", 7, 8), ("wait", "c"), ("finish",)], log
    log2 = []
    dp.run_program(lambda b, e: log2.append(("run", b, e)), 8, sync, lambda h, c: log2.append(("inline", h)), lambda h, c: log2.append(("bucket", h)),
                   lambda: log2.append(("finish",)))
    assert [t for t in log2 if t[0] != "run"] == [("inline", "a"), ("inline", "b"), ("bucket", "g"), ("inline", "c"), ("finish",)]
    assert [t for t in log2 if t[0] == "run"] == [("run", 0, 2), ("run", 2, 3), ("run", 3, 6), ("run", 6, 7), ("run", 7, 8)]
