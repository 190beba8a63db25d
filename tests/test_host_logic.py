"""CPU tests of the HOST logic (fit / evaluate / checkpoints / runners / weight files) with the CPU oracle
injected as the backend (tests/oracle_backend.py).  The product's default backend is the HIP engine."""
import json
import os

import numpy as np
import pytest

from covidseg_amd import weights as W
from covidseg_amd.data import synthetic_ct
from covidseg_amd.keras_like import UNetModel, sm_scores
from oracle import unet_oracle as O
from oracle_backend import OracleBackend


def small_model(size=16):
    m = UNetModel(size, backend=OracleBackend(size, size), seed=1)
    m.verbose = 0
    return m


def test_fit_history_batches_and_checkpoints(tmp_path):
    x, y = synthetic_ct(7, 16, seed=0)
    m = small_model()
    m.compile(lr=0.0005)
    fd, fl = str(tmp_path / "best_dice.hdf5"), str(tmp_path / "best_loss.hdf5")
    h = m.fit(x[:5], y[:5], batch_size=2, epochs=3, validation_data=(x[5:], y[5:]), checkpoint_dice=fd, checkpoint_loss=fl, shuffle_seed=3)
    assert set(h.history) == {"loss", "dice_coeff", "val_loss", "val_dice_coeff"} and all(len(v) == 3 for v in h.history.values())
    assert m.backend.tr.t == 9                                   # 3 epochs x ceil(5/2) steps, short last batch included
    assert h.history["loss"][2] < h.history["loss"][0]
    # best-slot semantics of ModelCheckpoint(save_best_only) T1:1046-1047
    best = int(np.argmax(h.history["val_dice_coeff"]))
    wd = W.load_weights(fd)
    if best == 2:
        cur = m.get_weights()
        assert all(np.array_equal(wd[k], cur[k]) for k in wd)
    assert os.path.exists(fl)
    # epoch metrics: loss is sample-weighted, dice is the mean of per-batch values -- replay epoch 1 by hand
    m2 = small_model(); m2.compile()
    order = np.random.RandomState(3).permutation(5)
    vals, sizes = [], []
    for i in range(0, 5, 2):
        idx = order[i:i + 2]; vals.append(m2.backend.train_batch(x[:5][idx], y[:5][idx])); sizes.append(len(idx))
    vals = np.array(vals)
    assert h.history["loss"][0] == pytest.approx(np.average(vals[:, 0], weights=sizes), rel=1e-6)
    assert h.history["dice_coeff"][0] == pytest.approx(vals[:, 1].mean(), rel=1e-6)


def test_model_save_carries_the_optimizer_and_load_model_resumes_the_fit(tmp_path):
    """model.save / ModelCheckpoint write a compiled model's optimizer (T1:1046-1047: `training_config`, `optimizer_weights/` = Adam's iteration count and moment
    slots in optimizer.weights order); keras.models.load_model (T1:67) restores it: 2 + 3 steps through a file == 5 steps in one go, bit for bit (oracle backend)."""
    from covidseg_amd.keras_like import load_model
    from covidseg_amd import hdf5_min as H5
    x, y = synthetic_ct(4, 16, seed=1)
    a = small_model(); a.compile(lr=0.0005)
    for _ in range(2):
        a.backend.train_batch(x, y)
    f = str(tmp_path / "unet_covid_weights_dice_coeff.hdf5")
    a.save(f)
    root = H5.read_file(f)
    tc = json.loads(H5._strs(root.attrs["training_config"])[0])
    assert tc["optimizer_config"]["class_name"] == "Adam" and tc["optimizer_config"]["config"]["learning_rate"] == 0.0005 and tc["loss"] == "bce_dice_loss"
    names = H5._strs(root["optimizer_weights"].attrs["weight_names"])
    P = len(W.trainable_names(1, "unet", (16, 16)))
    assert len(names) == 1 + 3 * P and names[0] == "Adam/iterations:0" and names[1] == "training/Adam/m_0:0" and names[1 + 2 * P] == "training/Adam/vhat_0:0"
    assert int(root["optimizer_weights"]["Adam/iterations:0"]) == 2 and root["optimizer_weights"]["Adam/iterations:0"].dtype == np.int64
    b = load_model(f, backend=OracleBackend(16, 16))
    assert b.compiled and b.backend.tr.t == 2 and b.arch == "unet" and (b.h, b.w, b.in_ch) == (16, 16, 1)
    for _ in range(3):
        a.backend.train_batch(x, y); b.backend.train_batch(x, y)
    wa, wb = a.get_weights(), b.get_weights()
    assert all(np.array_equal(wa[k], wb[k]) for k in wa)
    # compile=False / a model saved before compile(): weights only, a fresh optimizer
    c = load_model(f, backend=OracleBackend(16, 16), compile=False)
    assert not c.compiled and c.backend.tr.t == 0
    d = small_model(); d.save(f)
    assert "optimizer_weights" not in H5.read_file(f) and not load_model(f, backend=OracleBackend(16, 16)).compiled
    with pytest.raises(ValueError, match="model_config"):
        d.save_weights(f); load_model(f, backend=OracleBackend(16, 16))
    # a file compiled with another loss does not silently continue on bce_dice_loss; its weights alone still load
    opt = a.backend.get_optimizer_state()
    W.save_weights(f, a.get_weights(), 1, "unet", (16, 16), full_model=True, optimizer=dict(opt, loss="mean_squared_error"))
    with pytest.raises(ValueError, match="mean_squared_error"):
        load_model(f, backend=OracleBackend(16, 16))
    assert not load_model(f, backend=OracleBackend(16, 16), compile=False).compiled
    # a backend without optimizer-state support: weights load, a warning says Adam starts fresh
    a.save(f)

    class NoOptState(OracleBackend):
        set_optimizer_state = property()                                       # hasattr(...) is False
    with pytest.warns(UserWarning, match="starts fresh"):
        e = load_model(f, backend=NoOptState(16, 16))
    assert e.compiled and all(np.array_equal(e.get_weights()[k], a.get_weights()[k]) for k in wa)


def test_evaluate_is_mean_of_batch_metrics_and_weighted_loss():
    x, y = synthetic_ct(5, 16, seed=2)
    m = small_model()
    thr = np.array([0.3, 0.5])
    ev = m.evaluate(x, y, batch_size=2, thresholds=thr)
    ref = O.OracleTrainer(m.get_weights()).evaluate(x, y, batch_size=2, thresholds=thr.astype(np.float32))
    assert ev["loss"] == pytest.approx(ref["loss"], rel=1e-6) and ev["dice_coeff"] == pytest.approx(ref["dice_coeff"], rel=1e-6)
    for k in ("dice", "iou", "precision", "recall"):
        np.testing.assert_allclose(ev[k], ref[k], rtol=1e-6)
    p = m.predict(x, batch_size=2)
    assert p.shape == (5, 16, 16, 1)
    s = O.threshold_sums(y[:2], p[:2], thr.astype(np.float32))
    np.testing.assert_allclose(sm_scores(s[:, 0], s[:, 1], s[:, 2])["iou"], O.sm_scores(s[:, 0], s[:, 1], s[:, 2])["iou"])


def test_evaluate_reuses_its_uploaded_set_only_while_every_byte_is_unchanged():
    """evaluate() keeps the BatchSource of the arrays it was last given (the runners call it once per threshold sweep on one hold-out set, T1:1196-1330).  The reuse is
    keyed by identity AND a hash of the whole buffers: ANY in-place edit between two calls -- one mask pixel, far from any sampling stride -- is scored, not the stale copy;
    the cached set goes when the arrays die."""
    import gc
    x, y = synthetic_ct(5, 16, seed=4)
    m = small_model()
    be = m.backend
    uploads = []
    be.resident = lambda a, max_fraction=0.5: (uploads.append(id(a)), np.array(a, np.float32))[1]          # a backend with a resident set: the "device" copy
    be.take = lambda d, idx: d[np.asarray(idx)]
    a = m.evaluate(x, y, batch_size=2)
    src = m._eval_cache[2]
    assert m.evaluate(x, y, batch_size=2) == a and m._eval_cache[2] is src and len(uploads) == 2          # same bytes: the same source, nothing uploaded again
    y[3, 7, 9, 0] = 1.0 - y[3, 7, 9, 0]                                              # one element edited in place
    b = m.evaluate(x, y, batch_size=2)
    assert m._eval_cache[2] is not src and b["loss"] != a["loss"] and len(uploads) == 4
    ref = O.OracleTrainer(m.get_weights()).evaluate(x, y, batch_size=2)
    assert b["loss"] == pytest.approx(ref["loss"], rel=1e-6)
    m.evaluate(list(x), list(y), batch_size=2)                                       # (no array identity to hold on to: a fresh source)
    del x, y; gc.collect()
    assert m._eval_cache is None                                                     # the arrays died: the resident copy is released with them
    del be.resident, be.take
    x, y = synthetic_ct(5, 16, seed=4)
    m.evaluate(x, y, batch_size=2)
    assert m._eval_cache is None                                                     # a host-side source is not kept (it would save nothing and hold the arrays)


def test_weight_file_roundtrip_and_keras_names(tmp_path):
    w = W.init_weights(3)
    f = str(tmp_path / "unet_0.8954_cosine_annealer.h5")          # reference file name T1:1079
    W.save_weights(f, w)
    from covidseg_amd import hdf5_min as H5
    assert H5.is_hdf5(f)                                           # a Keras HDF5 weight file, not an .npz under another name
    layers, _ = H5.load_keras_weights(f)
    assert len(layers) == 44 and list(layers)[:3] == ["input_1", "conv2d_1", "conv2d_2"] and list(layers)[-1] == "conv2d_19"
    assert list(layers["conv2d_1"]) == ["conv2d_1/kernel:0", "conv2d_1/bias:0"] and not layers["dropout_2"]
    assert list(layers["batch_normalization_8"]) == [f"batch_normalization_8/{p}:0" for p in ("gamma", "beta", "moving_mean", "moving_variance")]
    assert sum(len(v) for v in layers.values()) == len(w) == 78
    w2 = W.load_weights(f)
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    g = str(tmp_path / "w.npz")                                    # explicit .npz: archive keyed by the Keras weight names
    W.save_weights(g, w)
    z = np.load(g)
    assert "conv2d_1/kernel:0" in z.files and "batch_normalization_8/moving_variance:0" in z.files and len(z.files) == 78
    assert all(np.array_equal(w[k], v) for k, v in W.load_weights(g).items())
    import json
    js = json.loads(W.to_json(224, 224))                           # model.to_json() T1:1091: Keras 2.3 functional-model schema
    assert js["class_name"] == "Model" and js["keras_version"] == "2.3.1" and len(js["config"]["layers"]) == 44
    l1 = js["config"]["layers"][1]
    assert l1["name"] == "conv2d_1" and l1["class_name"] == "Conv2D" and l1["config"]["filters"] == 32 and l1["inbound_nodes"] == [[["input_1", 0, 0, {}]]]
    assert js["config"]["layers"][0]["config"]["batch_input_shape"] == [None, 224, 224, 1] and js["config"]["output_layers"] == [["conv2d_19", 0, 0]]


def test_init_statistics():
    w = W.init_weights(0)
    k = w["c3a/kernel"]                                            # he_normal, truncated at 2 sigma
    std = np.sqrt(2.0 / (9 * 64))
    assert abs(k.std() - std) < 0.02 * std and np.abs(k).max() <= 2.0 * std / 0.87962566 + 1e-6
    u = w["u6/kernel"]; lim = np.sqrt(6.0 / (4 * 256 + 4 * 512))
    assert np.abs(u).max() <= lim and abs(u.std() - lim / np.sqrt(3)) < 0.02 * lim
    assert (w["bn1/gamma"] == 1).all() and (w["bn1/var"] == 1).all() and (w["c1a/bias"] == 0).all()


def test_runner_prints_reference_summary_and_returns_tables(tmp_path, capsys):
    from covidseg_amd.runners import holdout_runner_unet_infection_segmentation, runner_lung_segmentation
    x, y = synthetic_ct(8, 16, seed=0)
    out = holdout_runner_unet_infection_segmentation(data=(x, y), epochs=2, batch_size=4, workdir=str(tmp_path), verbose=0,
                                                     backend=OracleBackend(16, 16))
    txt = capsys.readouterr().out
    for label in ("(5, 16, 16, 1) (3, 16, 16, 1)", "test loss, test dice coefficient:", "DICES:", "IOUS:", "Best Threshold:", "Best dice score:",
                  "Best iou score:", "We just checked for 80 steps between 0.52 and 0.6", "NEW DICES:", "NEW IOUS:", "New Best Threshold:",
                  "PRECISIONS:", "RRECALLS:", "Best Threshold for Precision:", "Best recall score:"):
        assert label in txt, label
    assert len(out["dices"]) == 14 and len(out["new_dices"]) == 80 and len(out["precisions"]) == 20     # T1:1196, 1250, 1304
    assert os.path.exists(tmp_path / "unet_covid_weights_dice_coeff.hdf5") and os.path.exists(tmp_path / "unet_covid_weights_val_loss.hdf5")
    out3 = runner_lung_segmentation(data=(x, y), epochs=1, batch_size=4, workdir=str(tmp_path), verbose=0, backend=OracleBackend(16, 16))
    assert len(out3["new_dices"]) == len(np.arange(0.43, 0.53, 0.001)) and abs(out3["new_range"][0] - 0.43) < 1e-9                      # T3:1206


def test_kfold_indices_match_sklearn():
    from sklearn.model_selection import KFold
    from covidseg_amd.data import kfold_indices
    for n, k in ((10, 3), (12, 4), (37, 4), (64, 3)):
        want = list(KFold(n_splits=k, random_state=42, shuffle=True).split(np.arange(n)))
        got = kfold_indices(n, k, 42)
        assert len(got) == k
        for (a, b), (c, d) in zip(want, got):
            assert a.tolist() == c.tolist() and b.tolist() == d.tolist()


def test_kfold_runner_reference_flow_and_quirks(tmp_path, capsys):
    from covidseg_amd.runners import four_fold_runner_unet_infection_segmentation, three_fold_runner_unet_infection_segmentation
    x, y = synthetic_ct(9, 16, seed=3)
    out = three_fold_runner_unet_infection_segmentation(data=(x, y), epochs=1, batch_size=4, workdir=str(tmp_path), verbose=0,
                                                        backend=OracleBackend(16, 16))
    txt = capsys.readouterr().out
    for label in ("Current fold number going: 3", "Shapes: (6, 16, 16, 1) (3, 16, 16, 1)", "Time of 3-fold cross validation:",
                  "test loss, test dice coefficient:", "Calculating for threshold:", "3-fold Dices dataframe",
                  "Maximum validation iou on each of the 3 splits (any threshold chosen):", "Mean of all obtained recalls:"):
        assert label in txt, label
    assert out["table_dice"].shape == (len(np.arange(0.30, 0.80, 0.05)), 3) and len(out["scores"]) == 3
    # reference quirk CV4:1105-1108: the final weights overwrite every fold file -> all fold files identical
    w1, w3 = W.load_weights(out["paths"][0]), W.load_weights(out["paths"][2])
    assert all(np.array_equal(w1[k], w3[k]) for k in w1)
    # the model is NOT re-initialised between folds (CV4:1051-1090): 3 folds x ceil(6/4) steps accumulated in one trajectory
    assert out["model"].backend.tr.t == 2                         # optimizer state is reset by every compile(), weights are not
    os.makedirs(tmp_path / "f4", exist_ok=True)
    out4 = four_fold_runner_unet_infection_segmentation(data=(x[:8], y[:8]), epochs=1, batch_size=8, workdir=str(tmp_path / "f4"), verbose=0,
                                                        backend=OracleBackend(16, 16), reinit_each_fold=True, overwrite_fold_files=False)
    assert out4["table_iou"].shape[1] == 4 and len(out4["paths"]) == 4


def test_intermediate_output_by_keras_layer_name():
    """T1:1385-1387: Model(inputs, outputs=model.get_layer('conv2d_9').output) -- conv2d_9 is the first bottleneck conv (c5a)."""
    from covidseg_amd.data import synthetic_ct
    from covidseg_amd.keras_like import UNetModel
    from oracle import unet_oracle as O
    from oracle_backend import OracleBackend
    x, _ = synthetic_ct(3, 32, seed=4)
    m = UNetModel(32, backend=OracleBackend(32, 32), seed=2)
    f = m.intermediate_output("conv2d_9", x, batch_size=2)
    assert f.shape == (3, 2, 2, 512)
    import torch
    with torch.no_grad():
        acts = O.forward(m.get_weights(), x, training=False, want_acts=True)[1]
    assert np.allclose(f, acts["c5a"].numpy(), atol=1e-6) and np.allclose(m.intermediate_output("c5a", x), f, atol=1e-6)
    assert m.intermediate_output("batch_normalization_3", x).shape == (3, 8, 8, 128)
