"""Host-side mirror of what the reference's classification runner does around its Keras model
(task2_covid19_classifcation.py, `T2`): the Sequential CNN T2:747-776, ``compile`` T2:829, ``fit`` with the RocCallback +
ModelCheckpoint(val_loss) + class_weight T2:707-741, 811-835, ``evaluate`` T2:884, ``predict`` T2:910, and the sklearn / plot-metric
helpers it calls (StratifiedShuffleSplit T2:647, compute_class_weight T2:801, roc_auc_score T2:727, confusion-matrix report T2:930-967).

The arithmetic runs in a *backend* (default engine.HipUNet(arch="classifier") on an MI355X); there is no implicit fallback.
"""
from __future__ import annotations

import math

import numpy as np

from . import weights as W
from .keras_like import BatchSource, check_comm, dp_info, dp_shard


# ----------------------------------------------------------------------------------------- sklearn / plot-metric restatements
def _approximate_mode(class_counts, n_draws, rng):
    """sklearn.utils._approximate_mode: floor of the proportional allocation, leftovers by largest remainder, ties drawn with rng."""
    continuous = class_counts / class_counts.sum() * n_draws
    floored = np.floor(continuous)
    need = int(n_draws - floored.sum())
    if need > 0:
        remainder = continuous - floored
        for value in np.sort(np.unique(remainder))[::-1]:
            (inds,) = np.where(remainder == value)
            add_now = min(len(inds), need)
            inds = rng.choice(inds, size=add_now, replace=False)
            floored[inds] += 1
            need -= add_now
            if need == 0:
                break
    return floored.astype(int)


def stratified_shuffle_split(y, test_size: float = 0.3, random_state: int = 42, use_sklearn: bool = True):
    """One split of sklearn.model_selection.StratifiedShuffleSplit(n_splits=1, test_size=.3, random_state=42).split(X, y) (T2:647-650):
    returns (train_index, test_index).  scikit-learn itself where it is installed (the reference's own call); the restatement below is what runs where it is
    not (tests/test_classifier_host.py holds the two together)."""
    y = np.asarray(y)
    if use_sklearn:
        try:
            from sklearn.model_selection import StratifiedShuffleSplit
            return next(StratifiedShuffleSplit(n_splits=1, test_size=test_size, random_state=random_state).split(np.zeros(len(y)), y))
        except ImportError:
            pass
    n = len(y)
    n_test = int(math.ceil(test_size * n)); n_train = n - n_test
    classes, y_idx = np.unique(y, return_inverse=True)
    counts = np.bincount(y_idx)
    if counts.min() < 2:
        raise ValueError("stratified_shuffle_split: the least populated class has only 1 member")
    if n_train < len(classes) or n_test < len(classes):
        raise ValueError("stratified_shuffle_split: train/test size smaller than the number of classes")
    class_indices = np.split(np.argsort(y_idx, kind="mergesort"), np.cumsum(counts)[:-1])
    rng = np.random.RandomState(random_state)
    n_i = _approximate_mode(counts, n_train, rng)
    t_i = _approximate_mode(counts - n_i, n_test, rng)
    train, test = [], []
    for i in range(len(classes)):
        perm = class_indices[i].take(rng.permutation(counts[i]), mode="clip")
        train.extend(perm[: n_i[i]]); test.extend(perm[n_i[i]: n_i[i] + t_i[i]])
    return rng.permutation(train), rng.permutation(test)


def compute_class_weight_balanced(y, use_sklearn: bool = True):
    """sklearn.utils.class_weight.compute_class_weight('balanced', np.unique(y), y) (T2:801-803): n / (n_classes * bincount); scikit-learn's own where installed."""
    y = np.asarray(y)
    if use_sklearn:
        try:
            from sklearn.utils.class_weight import compute_class_weight
            return compute_class_weight(class_weight="balanced", classes=np.unique(y), y=y)
        except ImportError:
            pass
    classes, y_idx = np.unique(y, return_inverse=True)
    return len(y) / (len(classes) * np.bincount(y_idx).astype(np.float64))


def roc_auc_score(y_true, y_score):
    """sklearn.metrics.roc_auc_score for binary labels (T2:727, 729): area under the ROC polyline = Mann-Whitney U with
    average ranks for tied scores."""
    y_true = np.asarray(y_true).reshape(-1); s = np.asarray(y_score, np.float64).reshape(-1)
    pos = y_true == np.max(y_true)
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(s, kind="mergesort")
    ss = s[order]
    ranks = np.empty(len(s), np.float64)
    i = 0
    while i < len(ss):
        j = i
        while j + 1 < len(ss) and ss[j + 1] == ss[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def confusion_report(y_true, y_pred, threshold):
    """The numbers the reference prints from plot-metric's BinaryClassification(...).plot_confusion_matrix() (T2:930-967):
    class = p > threshold; returns dict(tn, fp, fn, tp, accuracy, precision, recall, f1)."""
    t = np.asarray(y_true).reshape(-1).astype(int); c = (np.asarray(y_pred).reshape(-1) > threshold).astype(int)
    tn = int(((t == 0) & (c == 0)).sum()); fp = int(((t == 0) & (c == 1)).sum())
    fn = int(((t == 1) & (c == 0)).sum()); tp = int(((t == 1) & (c == 1)).sum())
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = np.float64(tp) / (tp + fp); recall = np.float64(tp) / (tp + fn)
        f1 = 2 * precision * recall / (precision + recall)
    return dict(tn=tn, fp=fp, fn=fn, tp=tp, accuracy=(tp + tn) / max(tp + tn + fp + fn, 1), precision=float(precision), recall=float(recall),
                f1=float(f1))


def _host(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v, np.float64)


class History:
    def __init__(self):
        self.history = {"loss": [], "f1": [], "val_loss": [], "val_f1": [], "roc_auc_train": [], "roc_auc_val": []}


class ClassifierModel:
    """Sequential([...]) of T2:747-776 with the Keras calls the runner makes."""

    def __init__(self, input_size: int = 224, in_ch: int = 1, backend=None, seed: int = 0, **backend_kw):
        self.h = self.w = int(input_size)
        self.in_ch = in_ch
        if backend is None:
            from .engine import HipUNet                      # raises loudly without GPU / library
            backend = HipUNet(self.h, self.w, in_ch, seed=seed, arch="classifier", dropout_rate=0.4, **backend_kw)
        self.backend = backend
        self.backend.set_weights(W.init_weights(seed, in_ch, "classifier", (self.h, self.w)))
        self.compiled = False
        self.verbose = 1
        self.best_val_auc = -1.0                              # the reference's global best_val_auc (T2:813)

    @property
    def _hw(self):
        return (self.h, self.w)

    def count_params(self):
        return W.count_params(self.in_ch, "classifier", self._hw)[0]

    def summary(self, print_fn=print):
        total, train = W.count_params(self.in_ch, "classifier", self._hw)
        for n, k, ci, co in W.layer_table(self.in_ch, "classifier", self._hw):
            print_fn(f"{n:6s} {k:6s} {ci:6d} -> {co:4d}")
        print_fn(f"Total params: {total:,}\nTrainable params: {train:,}\nNon-trainable params: {total - train:,}")

    def compile(self, lr: float = 0.0005, loss: str = "binary_crossentropy", metrics=("f1",)):
        """model.compile(loss='binary_crossentropy', optimizer=Adam(lr=0.0005), metrics=[f1]) T2:829."""
        if loss != "binary_crossentropy":
            raise ValueError("only binary_crossentropy (T2:829) is implemented")
        self.backend.lr = float(lr)
        self.backend.reset_optimizer()
        self.compiled = True

    def get_weights(self):
        return self.backend.get_weights()

    def set_weights(self, w):
        self.backend.set_weights(w)

    def save_weights(self, path):
        W.save_weights(path, self.backend.get_weights(), self.in_ch, "classifier", self._hw)

    def save(self, path):
        """model.save(path): what ModelCheckpoint(filepath_loss, ...) writes (T2:820) -- `model_weights/` + `model_config` and the optimizer of a compiled model."""
        opt = self.backend.get_optimizer_state() if self.compiled and hasattr(self.backend, "get_optimizer_state") else None
        if opt is not None:
            opt = dict(opt, loss="binary_crossentropy", metrics=["f1"])
        W.save_weights(path, self.backend.get_weights(), self.in_ch, "classifier", self._hw, full_model=True, optimizer=opt)

    def load_weights(self, path):
        self.backend.set_weights(W.load_weights(path, self.in_ch, "classifier", self._hw))

    def to_json(self):
        return W.to_json(self.h, self.w, self.in_ch, "classifier")

    @staticmethod
    def _class_weight_pair(class_weight, honour_array):
        """Keras 2.3 applies class_weight only when it is a dict (training_utils.standardize_weights); the reference passes the
        ndarray from sklearn (T2:801, 835), which Keras silently ignores.  Default: reproduce that; honour_array=True applies it."""
        if class_weight is None:
            return (1.0, 1.0), False
        if isinstance(class_weight, dict):
            return (float(class_weight.get(0, 1.0)), float(class_weight.get(1, 1.0))), True
        if honour_array:
            cw = np.asarray(class_weight, np.float64).reshape(-1)
            return (float(cw[0]), float(cw[1])), True
        return (1.0, 1.0), False

    def fit(self, x, y, batch_size=32, epochs=1, validation_data=None, class_weight=None, honour_array_class_weight=False,
            roc_callback=True, best_auc_path=None, checkpoint_loss=None, shuffle=True, shuffle_seed=0, dropout=True):
        """model.fit(x_train, y_train, batch_size=32, epochs=25, validation_data=..., callbacks=[roc, checkpoint_loss],
        class_weight=weights) T2:833-835.  Per epoch: shuffled mini-batches (short last batch), loss = sample-weighted mean of the
        batch losses, f1 = mean of the per-batch values; validation pass; RocCallback.on_epoch_end (T2:724-736: AUC on the whole
        train and validation sets, best validation AUC -> best_auc_path); ModelCheckpoint(val_loss, save_best_only) T2:820."""
        assert self.compiled, "call compile() first"
        y = np.asarray(y, np.float32).reshape(-1)
        cw, applied = self._class_weight_pair(class_weight, honour_array_class_weight)
        if class_weight is not None and not applied and self.verbose:
            print("note: class_weight given as an array -> ignored, as Keras 2.3 does (pass a dict or honour_array_class_weight=True)")
        hist = History()
        n = len(x)
        best_loss = np.inf
        rng = np.random.RandomState(shuffle_seed)
        world, rank = dp_info(self.backend)
        src = BatchSource(self.backend, x, y)                         # the training set goes to HBM once (keras_like.BatchSource)
        for ep in range(epochs):
            order = rng.permutation(n) if shuffle else np.arange(n)
            self.backend.set_class_weights(*cw)
            outs, sizes = [], []
            for i in range(0, n, batch_size):
                idx = order[i:i + batch_size]
                sel, kw = dp_shard(idx, world, rank)                  # data parallel: keras_like.dp_shard
                xb, yb = src(sel)
                outs.append(self.backend.train_batch(xb, yb, dropout, **kw)); sizes.append(len(idx))
            vals = np.stack([_host(o) for o in outs])
            check_comm(self.backend, "fit")
            hist.history["loss"].append(float(np.average(vals[:, 0], weights=sizes))); hist.history["f1"].append(float(vals[:, 1].mean()))
            line = f"Epoch {ep + 1}/{epochs} - loss: {hist.history['loss'][-1]:.4f} - f1: {hist.history['f1'][-1]:.4f}"
            if validation_data is not None:
                xv, yv = validation_data
                ev = self.evaluate(xv, yv, batch_size=batch_size)
                hist.history["val_loss"].append(ev[0]); hist.history["val_f1"].append(ev[1])
                line += f" - val_loss: {ev[0]:.4f} - val_f1: {ev[1]:.4f}"
                if roc_callback:
                    roc_train = roc_auc_score(y, self.predict_proba(x)); roc_val = roc_auc_score(yv, self.predict_proba(xv))
                    hist.history["roc_auc_train"].append(roc_train); hist.history["roc_auc_val"].append(roc_val)
                    if self.verbose:
                        print('\rroc-auc_train: %s - roc-auc_val: %s' % (str(round(roc_train, 4)), str(round(roc_val, 4))), end=100 * ' ' + '\n')
                    if self.best_val_auc < roc_val:
                        self.best_val_auc = roc_val
                        if best_auc_path and rank == 0:
                            self.save_weights(best_auc_path)
                        if self.verbose:
                            print("Saving best validation AUC weights")
                if checkpoint_loss and ev[0] < best_loss:
                    if self.verbose:
                        print(f"\nEpoch {ep + 1:05d}: val_loss improved from {best_loss:.5f} to {ev[0]:.5f}, saving model to {checkpoint_loss}")
                    best_loss = ev[0]
                    if rank == 0:
                        self.save(checkpoint_loss)
            if self.verbose:
                print(line)
        if world > 1:
            self.backend.barrier()
        return hist

    def evaluate(self, x, y, batch_size=32):
        """model.evaluate(x_valid, y_valid, batch_size=32) T2:884 -> [loss, f1] (loss without class weights; f1 = mean of batch values)."""
        y = np.asarray(y, np.float32).reshape(-1)
        self.backend.set_class_weights(1.0, 1.0)
        vals, sizes = [], []
        world, rank = dp_info(self.backend)
        for i in range(0, len(x), batch_size):
            sel, kw = dp_shard(np.arange(i, min(i + batch_size, len(x))), world, rank)
            _, ld = self.backend.predict_batch(x[sel], y[sel], **kw)
            vals.append(ld); sizes.append(min(i + batch_size, len(x)) - i)
        v = np.stack([_host(a) for a in vals])
        check_comm(self.backend, "evaluate")
        return [float(np.average(v[:, 0], weights=sizes)), float(v[:, 1].mean())]

    def predict(self, x, batch_size=32):
        """model.predict(x_valid) T2:910 -> [n, 1]."""
        outs = []
        kw = {"replicated": True} if dp_info(self.backend)[0] > 1 else {}
        for i in range(0, len(x), batch_size):
            p, _ = self.backend.predict_batch(x[i:i + batch_size], **kw)
            outs.append(p)
        return np.concatenate([(_o.detach().cpu().numpy() if hasattr(_o, "detach") else np.asarray(_o)) for _o in outs], 0).reshape(-1, 1)

    predict_proba = predict                                   # Sequential.predict_proba (T2:726, 728)
