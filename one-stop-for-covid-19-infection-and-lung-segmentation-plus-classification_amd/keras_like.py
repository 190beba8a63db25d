"""Host-side mirror of the Keras surface the reference's runners consume
(task1_preprocessing_plus_unet_with_comments.py): Model construction T1:853-916,
``compile`` T1:1053, ``fit`` T1:1059-1061 with two ModelCheckpoint(save_best_only) slots
T1:1044-1047, ``load_weights``/``save_weights``/``to_json`` T1:1073-1093, ``evaluate`` T1:1101,
threshold sweeps with segmentation_models metrics T1:1196-1330, ``predict`` T1:1137.

The arithmetic runs in a *backend* (default: engine.HipUNet on an MI355X).  The backend is an
explicit constructor argument so host logic can be unit-tested; there is no implicit fallback.
"""
from __future__ import annotations

import time
import warnings

import numpy as np

from . import weights as W

SM_SMOOTH = 1e-5


def sm_scores(tp, spr, sgt, smooth=SM_SMOOTH):
    """segmentation_models FScore(beta=1) / IOUScore / Precision / Recall from batch sums
    (gt is NOT thresholded: soft labels, as in the reference)."""
    tp, spr, sgt = (np.asarray(a, np.float64) for a in (tp, spr, sgt))
    fp, fn = spr - tp, sgt - tp
    return {"dice": (2 * tp + smooth) / (2 * tp + fn + fp + smooth), "iou": (tp + smooth) / (sgt + spr - tp + smooth),
            "precision": (tp + smooth) / (tp + fp + smooth), "recall": (tp + smooth) / (tp + fn + smooth)}


def _host(v):
    """device tensor / array / tuple -> numpy float64 array"""
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, np.float64)


def dp_info(backend):
    """(world, rank) of a data-parallel backend (engine.HipUNet built with a process group), else (1, 0)."""
    return (int(backend.world), int(backend.rank)) if getattr(backend, "_dp", False) else (1, 0)


def check_comm(backend, where):
    """A device-side all-reduce that timed out (csrc/comm.hip: a rank more than the timeout behind) leaves the BatchNorm / loss sums rank-local from then on --
    sync-BN training would silently turn into unsynchronised statistics.  Read the communicator's sticky error word wherever the host synchronises anyway
    (per epoch in fit, per call in evaluate) and fail loudly.  Backends without the hook (CPU test backends) have nothing to check."""
    st = backend.comm_status() if hasattr(backend, "comm_status") else 0
    if st:
        raise RuntimeError(f"{where}: a device-side all-reduce did not receive rank {st - 1}'s contribution within the timeout; the BatchNorm / loss sums of this "
                           f"run are not the global batch's (HipUNet(small_allreduce='rccl') waits instead of timing out)")


def dp_shard(idx, world, rank):
    """How one global mini-batch is run on `world` ranks so that the step equals the single-process step on the whole batch
    (the engine reduces BatchNorm sums, Dice sums and gradients over the ranks, engine.py): a batch whose size divides by the world size
    is cut into equal contiguous shards, rank r takes the r-th -> (shard, {}); any other size (the short last batch of an epoch,
    T1:1059: 1130 = 35 x 32 + 10) is run by every rank in full with no reduction -> (idx, {"replicated": True}): identical replicas
    apply the identical update."""
    if world == 1:
        return idx, {}
    m = len(idx)
    if m % world == 0:
        q = m // world
        return idx[rank * q:(rank + 1) * q], {}
    return idx, {"replicated": True}


def load_model(path, backend=None, custom_objects=None, compile=True, **backend_kw):
    """keras.models.load_model(path) (imported by every script of the reference, T1:67): the graph named by the file's `model_config`, its weights, and -- when
    the file carries them and compile is true -- the compiled state: Adam's learning rate, iteration count and moment slots, so a further fit() continues the
    saved one.  custom_objects is accepted for signature compatibility (the losses / metrics of the path are built in)."""
    import json
    from . import hdf5_min as H5
    from . import keras_graph as KG
    root = H5.read_file(path)
    if "model_config" not in root.attrs:
        raise ValueError(f"{path}: no `model_config` attribute -- a weights-only file (use load_weights)")
    cfg = json.loads(H5._strs(root.attrs["model_config"])[0])
    layers = cfg["config"]["layers"] if isinstance(cfg["config"], dict) else cfg["config"]
    shape = next((l["config"]["batch_input_shape"] for l in layers if "batch_input_shape" in l.get("config", {})), None)
    if shape is None:
        raise ValueError(f"{path}: model_config carries no batch_input_shape")
    h, w, in_ch = int(shape[1]), int(shape[2]), int(shape[3])
    names = [l["config"]["name"] for l in layers]
    arch = next((a for a in (("classifier",) if cfg["class_name"] == "Sequential" else ("unet", "unetpp"))
                 if [l["name"] for l in KG.keras_layers(in_ch, a, (h, w))] == names), None)
    if arch is None:
        raise ValueError(f"{path}: the saved graph ({cfg['class_name']}, {len(names)} layers, input {shape}) is none of the three this engine runs")
    if h != w:
        raise ValueError(f"{path}: the saved graph is {h} x {w}; the Keras-shaped model classes of this package take ONE input_size (new_dim, T1:479) -- "
                         f"non-square inputs are supported by the engine (HipUNet(h, w)) but not by load_model")
    if arch == "classifier":
        from .classifier import ClassifierModel
        model = ClassifierModel(h, in_ch, backend=backend, **backend_kw)
        own_loss, own_metrics = "binary_crossentropy", ["f1"]
    else:
        model = UNetModel(h, in_ch, backend=backend, arch=arch, **backend_kw)
        own_loss, own_metrics = "bce_dice_loss", ["dice_coeff"]
    model.load_weights(path)
    opt = W.load_optimizer(path, in_ch, arch, (h, w)) if compile else None
    if opt is not None:
        # the compiled state names its loss / metrics: this package implements exactly one pair per graph (T1:1053, T2:829) -- a file trained on another loss must
        # not silently continue on ours
        if opt.get("loss") not in (None, own_loss):
            raise ValueError(f"{path}: compiled with loss {opt['loss']!r}; this engine implements {own_loss!r} for the {arch} graph (load with compile=False for the weights only)")
        if opt.get("metrics") and [m_ for m_ in opt["metrics"] if m_ not in own_metrics]:
            warnings.warn(f"{path}: saved metrics {opt['metrics']} -- only {own_metrics} are computed here")
        model.compile(lr=opt["lr"])
        if hasattr(model.backend, "set_optimizer_state"):
            model.backend.set_optimizer_state(opt)
        else:
            warnings.warn(f"{path}: the backend keeps no optimizer state -- weights loaded, Adam starts fresh")
    return model


class BatchSource:
    """x[idx] / y[idx] of model.fit / evaluate / predict (T1:1059-1061, 1101, 1137).  With a backend that can keep a dataset in HBM (engine.HipUNet.resident)
    the arrays are uploaded ONCE and a mini-batch is a device-side gather (or a view, for a contiguous range); otherwise -- a set too large for the
    device, or a backend without the hook (the CPU test backends) -- batches are cut on the host from an fp32 copy made once and travel through the
    backend's staging (pinned, asynchronous in the engine).  device_resident: True / False / "auto" (upload when it fits)."""

    def __init__(self, backend, *arrays, device_resident="auto"):
        self.backend = backend
        self.dev = []
        for a in arrays:
            d = None
            if a is not None and device_resident and hasattr(backend, "resident") and hasattr(backend, "take"):
                d = backend.resident(a) if device_resident == "auto" else backend.resident(a, max_fraction=1.0)
            if d is None and a is not None and not hasattr(a, "detach"):
                a = np.asarray(a)
                if a.dtype != np.float32:
                    a = a.astype(np.float32)                              # once, not per batch (the reference feeds float64, T1:520)
            self.dev.append((d, a if d is None else True))          # (a device copy: the host array is not held -- the caller's reference is the only one)
        self.all_resident = all(d is not None for d, a in self.dev if a is not None)

    def __call__(self, idx):
        out = []
        for d, a in self.dev:
            if a is None:
                out.append(None)
            elif d is not None:
                out.append(self.backend.take(d, idx))
            else:
                idx = np.asarray(idx)
                out.append(a[int(idx[0]):int(idx[0]) + len(idx)] if len(idx) and np.all(np.diff(idx) == 1) else a[idx])
        return out


class History:
    def __init__(self):
        self.history = {"loss": [], "dice_coeff": [], "val_loss": [], "val_dice_coeff": []}
        self.epoch_seconds = []                       # wall time of every epoch incl. its validation pass (not a Keras field; bench.py reads it)


class UNetModel:
    def __init__(self, input_size: int = 224, in_ch: int = 1, backend=None, seed: int = 0, arch: str = "unet", **backend_kw):
        self.h = self.w = int(input_size)
        self.in_ch = in_ch
        self.arch = arch
        if backend is None:
            from .engine import HipUNet                      # raises loudly without GPU / library
            backend = HipUNet(self.h, self.w, in_ch, seed=seed, arch=arch, **backend_kw)
        self.backend = backend
        self.backend.set_weights(W.init_weights(seed, in_ch, arch))
        self.compiled = False
        self.verbose = 1

    # --- Keras-shaped surface ---------------------------------------------------------
    def count_params(self):
        return W.count_params(self.in_ch, self.arch)[0]

    def summary(self, print_fn=print):
        total, train = W.count_params(self.in_ch, self.arch)
        for n, k, ci, co in W.layer_table(self.in_ch, self.arch):
            print_fn(f"{n:6s} {k:6s} {ci:4d} -> {co:4d}")
        print_fn(f"Total params: {total:,}\nTrainable params: {train:,}\nNon-trainable params: {total - train:,}")

    def compile(self, lr: float = 0.0005, loss: str = "bce_dice_loss", metrics=("dice_coeff",)):
        """model.compile(optimizer=Adam(lr), loss=bce_dice_loss, metrics=[dice_coeff]) T1:1053.
        Re-compiling keeps the weights and resets the optimizer state (as Keras does, T1:1208)."""
        if loss != "bce_dice_loss":
            raise ValueError("only bce_dice_loss (T1:797-799) is implemented on the hot path")
        self.backend.lr = float(lr)
        self.backend.reset_optimizer()
        self.compiled = True

    def get_weights(self):
        return self.backend.get_weights()

    def set_weights(self, w):
        self.backend.set_weights(w)

    def save_weights(self, path):
        """model.save_weights(path) T1:1079: Keras HDF5 weight file (weights.save_weights)."""
        W.save_weights(path, self.backend.get_weights(), self.in_ch, self.arch, (self.h, self.w))

    def save(self, path):
        """model.save(path) -- what ModelCheckpoint(save_weights_only=False) calls (T1:1046-1047): `model_weights/` + `model_config` and, for a compiled
        model, the optimizer (`training_config`, `optimizer_weights/`: Adam's iteration count and moment slots) so that load_model resumes the fit."""
        opt = self.backend.get_optimizer_state() if self.compiled and hasattr(self.backend, "get_optimizer_state") else None
        W.save_weights(path, self.backend.get_weights(), self.in_ch, self.arch, (self.h, self.w), full_model=True, optimizer=opt)

    def load_weights(self, path):
        self.backend.set_weights(W.load_weights(path, self.in_ch, self.arch, (self.h, self.w)))

    def to_json(self):
        return W.to_json(self.h, self.w, self.in_ch, self.arch)

    def fit(self, x, y, batch_size=32, epochs=1, validation_data=None, checkpoint_dice=None, checkpoint_loss=None,
            shuffle=True, shuffle_seed=0, dropout=True, device_resident="auto"):
        """model.fit(...) T1:1059-1061.  Per epoch: shuffle, bs-`batch_size` steps with a short
        last batch, loss = sample-weighted mean of batch losses, dice_coeff = mean of per-batch
        values; then a full validation pass; ModelCheckpoint(save_best_only) on val_dice_coeff
        (max) and val_loss (min) T1:1046-1047."""
        assert self.compiled, "call compile() first"
        hist = History()
        n = len(x)
        best_dice, best_loss = -np.inf, np.inf
        rng = np.random.RandomState(shuffle_seed)
        world, rank = dp_info(self.backend)       # data parallel (T3:989-1009 on N GPUs): every rank walks the same shuffled batches, each takes its shard
        src = BatchSource(self.backend, x, y, device_resident=device_resident)          # the training set goes to HBM once (when it fits), not batch by batch
        val = BatchSource(self.backend, validation_data[0], validation_data[1], device_resident=device_resident) if validation_data is not None else None
        for ep in range(epochs):
            t_ep = time.perf_counter()
            order = rng.permutation(n) if shuffle else np.arange(n)
            outs, sizes = [], []
            for i in range(0, n, batch_size):
                idx = order[i:i + batch_size]
                sel, kw = dp_shard(idx, world, rank)
                xb, yb = src(sel)
                outs.append(self.backend.train_batch(xb, yb, dropout, **kw))                  # [loss, dice_coeff] of the WHOLE batch on every rank
                sizes.append(len(idx))
            vals = np.stack([_host(o) for o in outs])                     # one host sync per epoch
            check_comm(self.backend, "fit")
            hist.history["loss"].append(float(np.average(vals[:, 0], weights=sizes)))
            hist.history["dice_coeff"].append(float(vals[:, 1].mean()))
            line = f"Epoch {ep + 1}/{epochs} - loss: {hist.history['loss'][-1]:.4f} - dice_coeff: {hist.history['dice_coeff'][-1]:.4f}"
            if validation_data is not None:
                ev = self.evaluate(validation_data[0], validation_data[1], batch_size=batch_size, verbose=0, _source=val)
                hist.history["val_loss"].append(ev["loss"]); hist.history["val_dice_coeff"].append(ev["dice_coeff"])
                line += f" - val_loss: {ev['loss']:.4f} - val_dice_coeff: {ev['dice_coeff']:.4f}"
                if checkpoint_dice and ev["dice_coeff"] > best_dice:
                    if self.verbose:
                        print(f"\nEpoch {ep + 1:05d}: val_dice_coeff improved from {best_dice:.5f} to {ev['dice_coeff']:.5f}, saving model to {checkpoint_dice}")
                    best_dice = ev["dice_coeff"]
                    if rank == 0:
                        self.save(checkpoint_dice)                   # replicas are identical: rank 0 writes the file
                if checkpoint_loss and ev["loss"] < best_loss:
                    if self.verbose:
                        print(f"\nEpoch {ep + 1:05d}: val_loss improved from {best_loss:.5f} to {ev['loss']:.5f}, saving model to {checkpoint_loss}")
                    best_loss = ev["loss"]
                    if rank == 0:
                        self.save(checkpoint_loss)
            hist.epoch_seconds.append(time.perf_counter() - t_ep)
            if self.verbose:
                print(line)
        if world > 1:
            self.backend.barrier()                                    # the checkpoint files exist before any rank goes on to load_weights (T1:1073)
        return hist

    def evaluate(self, x, y, batch_size=32, thresholds=None, verbose=0, device_resident="auto", _source=None):
        """model.evaluate T1:1101: loss = sample-weighted mean over batches; every metric = mean of
        the per-batch values.  With `thresholds`, ONE forward pass per batch feeds all thresholds
        (the reference re-compiles and re-runs evaluate per threshold, T1:1205-1211)."""
        losses, dices, sizes, per_batch = [], [], [], []
        world, rank = dp_info(self.backend)
        src = _source if _source is not None else self._eval_source(x, y, device_resident)
        for i in range(0, len(x), batch_size):
            sel, kw = dp_shard(np.arange(i, min(i + batch_size, len(x))), world, rank)
            xb, yb = src(sel)
            p, ld = self.backend.predict_batch(xb, yb, **kw)          # loss / dice_coeff and the threshold sums are the whole batch's on every rank
            losses.append(ld); sizes.append(min(i + batch_size, len(x)) - i)
            if thresholds is not None and len(thresholds):
                per_batch.append(self.backend.threshold_sums(p, yb, thresholds, **kw))
        vals = np.stack([_host(v) for v in losses])
        check_comm(self.backend, "evaluate")
        out = {"loss": float(np.average(vals[:, 0], weights=sizes)), "dice_coeff": float(vals[:, 1].mean())}
        if per_batch:
            sc = [sm_scores(s[:, 0], s[:, 1], s[:, 2]) for s in (_host(b) for b in per_batch)]
            for k in ("dice", "iou", "precision", "recall"):
                out[k] = np.mean([b[k] for b in sc], axis=0)
        return out

    def _eval_source(self, x, y, device_resident):
        """The BatchSource of an evaluate() call, kept while the SAME arrays come back (the runners sweep thresholds / call evaluate repeatedly on one hold-out
        set, T1:1196-1330): the set is uploaded to HBM once, not per call.  Reused only for NumPy arrays whose identity AND whole contents are unchanged (a 64-bit
        hash of every byte -- ~10 GB/s, cheap next to the upload it saves; an in-place edit between two calls is always seen); anything else (torch tensors, lists)
        gets a fresh source.  The cached set is released as soon as either array dies."""
        import weakref

        def digest(a):
            buf = memoryview(np.ascontiguousarray(a)).cast("B")
            try:
                import xxhash
                return xxhash.xxh3_64_intdigest(buf)
            except ImportError:
                import zlib
                return (zlib.crc32(buf) << 32) | zlib.adler32(buf)
        if not (isinstance(x, np.ndarray) and isinstance(y, np.ndarray)) or x.size == 0:
            return BatchSource(self.backend, x, y, device_resident=device_resident)
        key = (id(x), id(y), device_resident, x.shape, y.shape, str(x.dtype), str(y.dtype), digest(x), digest(y))
        c = getattr(self, "_eval_cache", None)
        if c is not None and c[0] == key and all(r() is o for r, o in zip(c[1], (x, y))):
            return c[2]
        self._eval_cache = None
        selfref = weakref.ref(self)

        def drop(_):                                           # (either array is gone: the device copy goes with it)
            m = selfref()
            if m is not None:
                m._eval_cache = None
        try:
            refs = (weakref.ref(x, drop), weakref.ref(y, drop))
        except TypeError:
            return BatchSource(self.backend, x, y, device_resident=device_resident)
        src = BatchSource(self.backend, x, y, device_resident=device_resident)
        if src.all_resident:                                   # (a host-side source saves no upload and would keep the arrays alive)
            self._eval_cache = (key, refs, src)
        return src

    def intermediate_output(self, layer_name, x, batch_size=32):
        """Model(inputs=model.input, outputs=model.get_layer(layer_name).output).predict(x)  (T1:1385-1387, layer 'conv2d_9' = the
        first bottleneck conv): the activation of a named layer in inference mode.  `layer_name` is a Keras auto-name
        ('conv2d_9', 'batch_normalization_3', 'conv2d_transpose_1') or an engine name ('c5a', 'bn3', 'u6', 'p2')."""
        rev = {v.split("/")[0]: k.split("/")[0] for k, v in W.keras_names(self.in_ch, self.arch).items()}
        name = rev.get(layer_name, layer_name)
        outs = []
        kw = {"replicated": True} if dp_info(self.backend)[0] > 1 else {}
        for i in range(0, len(x), batch_size):
            xb = x[i:i + batch_size]
            self.backend.predict_batch(xb, **kw)
            outs.append(self.backend.tap(len(xb), name, **kw))
        return np.concatenate(outs, 0)

    def predict(self, x, batch_size=32):
        """model.predict T1:1137."""
        outs = []
        kw = {"replicated": True} if dp_info(self.backend)[0] > 1 else {}          # every rank predicts everything (no gather needed)
        for i in range(0, len(x), batch_size):
            p, _ = self.backend.predict_batch(x[i:i + batch_size], **kw)
            outs.append(p)
        return np.concatenate([(_o.detach().cpu().numpy() if hasattr(_o, "detach") else np.asarray(_o)) for _o in outs], 0)
