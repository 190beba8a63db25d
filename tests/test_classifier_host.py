"""CPU tests of the classification runner's host side (task2_covid19_classifcation.py): the sklearn helpers it calls are
restated in covidseg_amd.classifier and pinned here against the REAL sklearn of this image; the Keras-shaped ClassifierModel
and runner_classification are driven through the CPU oracle backend (test infrastructure)."""
import os

import numpy as np
import pytest
import torch

from covidseg_amd import classifier as C
from covidseg_amd import weights as W
from covidseg_amd.data import synthetic_classification
from oracle import unet_oracle as O
from tests.oracle_backend import ClsOracleBackend


def test_split_class_weight_auc_match_sklearn():
    from sklearn.metrics import roc_auc_score
    from sklearn.model_selection import StratifiedShuffleSplit
    from sklearn.utils import class_weight
    rng = np.random.default_rng(0)
    for n in (10, 33, 57, 64, 200, 1001, 3520):
        y = (rng.random(n) < rng.uniform(0.2, 0.8)).astype(int); y[:4] = (0, 1, 0, 1)
        tr, te = C.stratified_shuffle_split(y, 0.3, 42, use_sklearn=False)               # T2:647 -- the RESTATEMENT (what a box without scikit-learn runs) against the library
        a, b = next(StratifiedShuffleSplit(n_splits=1, test_size=0.3, random_state=42).split(np.zeros(n), y))
        assert np.array_equal(tr, a) and np.array_equal(te, b), n
        assert np.allclose(C.compute_class_weight_balanced(y, use_sklearn=False),
                           class_weight.compute_class_weight(class_weight="balanced", classes=np.unique(y), y=y), rtol=0, atol=1e-15)
        for s in (rng.random(n), np.round(rng.random(n), 1), np.zeros(n)):              # continuous, heavily tied, all tied
            assert abs(C.roc_auc_score(y, s) - roc_auc_score(y, s)) < 1e-12
    with pytest.raises(ValueError):
        C.roc_auc_score(np.ones(5), np.arange(5))
    for use in (False, True):
        with pytest.raises(ValueError):
            C.stratified_shuffle_split(np.array([0, 0, 0, 1]), 0.3, 42, use_sklearn=use)


def test_confusion_report_and_f1_closure():
    from sklearn.metrics import confusion_matrix
    rng = np.random.default_rng(1)
    y = (rng.random(200) < 0.5).astype(int); p = np.clip(y * 0.4 + rng.random(200) * 0.6, 0, 1)
    for thr in (0.5, 0.81):
        r = C.confusion_report(y, p, thr)
        tn, fp, fn, tp = confusion_matrix(y, (p > thr).astype(int)).ravel()
        assert (r["tn"], r["fp"], r["fn"], r["tp"]) == (tn, fp, fn, tp)
        assert abs(r["f1"] - 2 * tp / (2 * tp + fp + fn)) < 1e-12
    # the reference's f1 closure (T2:688-703) on a case worked by hand: tp=2, predicted=3, possible=3 -> p=r=2/3
    t = torch.tensor([1., 1., 1., 0., 0.]); q = torch.tensor([0.9, 0.6, 0.2, 0.7, 0.5])      # round(0.5) = 0 (half to even)
    assert abs(float(O.cls_f1(t, q)) - 2 * (2 / 3) ** 2 / (4 / 3)) < 1e-6


def test_f1_closure_pinned_by_reference_goldens():
    """tests/golden/f1_goldens.npz was produced by executing the reference's own recall / precision / f1 closures (T2:688-703,
    tests/golden/make_f1_goldens.py): the oracle's restatement must reproduce them."""
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f1_goldens.npz"))
    for i in range(int(z["n_cases"])):
        t, p = torch.as_tensor(z[f"t{i}"]), torch.as_tensor(z[f"p{i}"])
        assert abs(float(O.cls_f1(t, p)) - float(z[f"f1{i}"])) < 1e-12, i
        tp = float(torch.sum(torch.round(torch.clamp(t * p, 0, 1))))
        assert abs(tp / (float(torch.sum(torch.round(torch.clamp(p, 0, 1)))) + 1e-7) - float(z[f"precision{i}"])) < 1e-12
        assert abs(tp / (float(torch.sum(torch.round(torch.clamp(t, 0, 1)))) + 1e-7) - float(z[f"recall{i}"])) < 1e-12


def test_tables_and_param_count():
    assert W.count_params(1, "classifier") == (1_678_385, 1_677_937)                   # model.summary() of T2:747-776
    assert list(W.weight_shapes(1, "classifier").items()) == list(O.cls_weight_shapes(1, (224, 224)).items())
    kn = W.keras_names(1, "classifier")
    assert kn["fc1/kernel"] == "dense_1/kernel:0" and kn["bn3b/var"] == "batch_normalization_6/moving_variance:0" and kn["c3b/bias"] == "conv2d_6/bias:0"
    assert W.weight_shapes(1, "classifier", (32, 48))["fc1/kernel"] == (4 * 6 * 64, 32)


def test_oracle_classifier_grads_finite_difference():
    """The oracle's autograd path against central differences (fp64) on a few entries of every parameter kind."""
    rng = np.random.default_rng(2)
    w = {k: v.astype(np.float64) for k, v in O.cls_init_weights(3, 1, (16, 16)).items()}
    for k in w:
        if k.endswith("/bias") or k.endswith("/beta"):
            w[k] = rng.standard_normal(w[k].shape) * 0.1
    x = rng.random((6, 16, 16, 1)); y = np.array([0, 1, 1, 0, 1, 0], np.float64)
    keep = (rng.random((6, 32)) > 0.4).astype(np.float64)
    r = O.cls_loss_and_grads(w, x, y, keep_mask=keep, class_weights=(0.7, 1.6), dtype=torch.float64)
    for name in ("c1a/kernel", "bn2a/gamma", "c3b/bias", "fc1/kernel", "fc2/kernel", "fc2/bias"):
        idx = tuple(rng.integers(0, s) for s in w[name].shape)
        eps = 1e-6
        vals = []
        for sgn in (+1, -1):
            w2 = dict(w); a = w[name].copy(); a[idx] += sgn * eps; w2[name] = a
            p = O.cls_forward(w2, x, training=True, keep_mask=keep, dtype=torch.float64)[0]
            vals.append(float(O.cls_loss(torch.as_tensor(y), p, (0.7, 1.6))))
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - r["grads"][name][idx]) < 1e-6 * max(1, abs(fd)), (name, fd, r["grads"][name][idx])


def test_fit_evaluate_roc_checkpoints_on_oracle_backend(tmp_path, capsys):
    x, y = synthetic_classification(24, 16, seed=0)
    be = ClsOracleBackend(16, 16)
    m = C.ClassifierModel(16, 1, backend=be, seed=0)
    m.verbose = 1
    m.compile(lr=0.0005)
    tr, te = C.stratified_shuffle_split(y, 0.3, 42)
    fa, fl = str(tmp_path / "best_val_auc_weights.h5"), str(tmp_path / "covid_weights_val_loss.hdf5")
    h = m.fit(x[tr], y[tr], batch_size=8, epochs=3, validation_data=(x[te], y[te]), class_weight=C.compute_class_weight_balanced(y[tr]),
              best_auc_path=fa, checkpoint_loss=fl, shuffle_seed=0)
    out = capsys.readouterr().out
    assert "roc-auc_train:" in out and "Saving best validation AUC weights" in out and "class_weight given as an array -> ignored" in out
    assert len(h.history["loss"]) == 3 and len(h.history["roc_auc_val"]) == 3 and os.path.exists(fa) and os.path.exists(fl)
    assert m.best_val_auc == max(h.history["roc_auc_val"])
    ev = m.evaluate(x[te], y[te], batch_size=8)
    ref = be.tr.evaluate(x[te], y[te], batch_size=8)
    assert abs(ev[0] - ref["loss"]) < 1e-6 and abs(ev[1] - ref["f1"]) < 1e-6
    # dict class weights ARE applied (Keras semantics): first batch loss = weighted mean
    be2 = ClsOracleBackend(16, 16); m2 = C.ClassifierModel(16, 1, backend=be2, seed=0); m2.verbose = 0; m2.compile()
    w0 = m2.get_weights()
    h2 = m2.fit(x[:8], y[:8], batch_size=8, epochs=1, class_weight={0: 0.5, 1: 2.0}, shuffle=False, dropout=False)
    exp = O.cls_loss_and_grads(w0, x[:8], y[:8], class_weights=(0.5, 2.0))["loss"]
    assert abs(h2.history["loss"][0] - exp) < 1e-6
    # weights round trip through the Keras-named container
    m.save_weights(str(tmp_path / "w.h5")); wa = m.get_weights()
    m2b = C.ClassifierModel(16, 1, backend=ClsOracleBackend(16, 16), seed=5); m2b.load_weights(str(tmp_path / "w.h5"))
    assert all(np.array_equal(wa[k], m2b.get_weights()[k]) for k in wa)
    assert m.predict(x[:5]).shape == (5, 1)


def test_runner_classification_on_oracle_backend(tmp_path, capsys):
    from covidseg_amd.runners import runner_classification
    x, y = synthetic_classification(20, 16, seed=1)
    out = runner_classification(data=(x, y), epochs=2, batch_size=8, backend=ClsOracleBackend(16, 16), workdir=str(tmp_path), verbose=0)
    txt = capsys.readouterr().out
    for s in ("(20, 16, 16, 1) (20,)", "Best saved AUCROC on validation set :", "test loss:", "test f1 score:", "Accuracy:", "F1 score:"):
        assert s in txt, s
    assert set(out["reports"]) == {0.5, 0.81} and len(out["predictions"]) == 6 and os.path.exists(tmp_path / "best_val_auc_weights.json")
    assert np.allclose(out["class_weights"], C.compute_class_weight_balanced(y[C.stratified_shuffle_split(y, 0.3, 42)[0]]))
