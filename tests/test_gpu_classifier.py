"""-m gpu: the slice classifier (task2_covid19_classifcation.py:747-776) on the HIP engine vs the CPU oracle: the dense /
head ops through the C ABI, the whole model (forward, loss, f1, every gradient, dropout with the engine's mask, class weights),
the optimizer trajectory and the runner."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(max(np.linalg.norm(a - b) - 1e-8 * np.sqrt(a.size), 0.0) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from gpu_util import Ops
    return Ops()


@pytest.mark.parametrize("shape", [(5, 64, 32), (32, 50176, 32), (3, 1000, 8), (130, 516, 16), (1, 4, 4)])
def test_dense_fwd_bwd(ops, shape):
    b, k, n = shape
    rng = np.random.default_rng(b + k)
    x = rng.standard_normal((b, k)).astype(np.float32); w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    bias = rng.standard_normal(n).astype(np.float32); dy = rng.standard_normal((b, n)).astype(np.float32)
    nb = ops.lib.unet_dense_ws_bytes(b, k, n)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    for act in (0, 1):
        y = ops.z(b, n)
        ops.ck(ops.lib.unet_dense_fwd(ops.h, ops.d(x).data_ptr(), ops.d(w).data_ptr(), ops.d(bias).data_ptr(), y.data_ptr(), b, k, n, act, 0.0, 0,
                                      ws.data_ptr(), nb, ops.s), "dense fwd")
        want = x64 @ w64 + bias
        assert relerr(y.cpu().numpy(), np.maximum(want, 0) if act else want) < 2e-5
    if (b * n) % 4 == 0 and b * n >= 64:                                          # fused dropout: zeros or value/(1-rate), reproducible
        y1, y2 = ops.z(b, n), ops.z(b, n)
        for yy in (y1, y2):
            ops.ck(ops.lib.unet_dense_fwd(ops.h, ops.d(x).data_ptr(), ops.d(w).data_ptr(), ops.d(bias).data_ptr(), yy.data_ptr(), b, k, n, 0, 0.4, 99,
                                          ws.data_ptr(), nb, ops.s), "dense fwd drop")
        a1 = y1.cpu().numpy(); keep = a1 != 0
        assert np.array_equal(a1, y2.cpu().numpy()) and relerr(a1[keep], ((x64 @ w64 + bias) / 0.6)[keep]) < 2e-5
        if b * n >= 1000:
            assert 0.5 < keep.mean() < 0.7
    dx, dw = ops.z(b, k), ops.z(k, n)
    ops.ck(ops.lib.unet_dense_bwd(ops.h, ops.d(x).data_ptr(), ops.d(w).data_ptr(), ops.d(dy).data_ptr(), dx.data_ptr(), dw.data_ptr(), b, k, n, ops.s), "dense bwd")
    assert relerr(dx.cpu().numpy(), dy.astype(np.float64) @ w64.T) < 2e-5
    assert relerr(dw.cpu().numpy(), x64.T @ dy.astype(np.float64)) < 2e-5


@pytest.mark.parametrize("b", [1, 7, 32, 300])
def test_cls_head_fwd_bwd(ops, b):
    n = 32
    rng = np.random.default_rng(b)
    a = rng.standard_normal((b, n)); keep = rng.random((b, n)) > 0.4
    h = (np.maximum(a, 0) * keep / 0.6).astype(np.float32)
    w = (rng.standard_normal((n, 1)) * 0.5).astype(np.float32); bias = np.array([0.1], np.float32)
    t = (rng.random(b) > 0.5).astype(np.float32)
    cw = (0.7, 1.9)
    p = ops.z(b); sums = ops.z(4, dtype=torch.float64); out = ops.z(2)
    ops.ck(ops.lib.unet_cls_head_fwd(ops.h, ops.d(h).data_ptr(), ops.d(w).data_ptr(), ops.d(bias).data_ptr(), p.data_ptr(), ops.d(t).data_ptr(), cw[0], cw[1],
                                     sums.data_ptr(), b, n, ops.s), "cls head fwd")
    ops.ck(ops.lib.unet_cls_loss_finalize(ops.h, sums.data_ptr(), float(b), out.data_ptr(), ops.s), "cls finalize")
    ht = torch.tensor(h.astype(np.float64), requires_grad=True); wt = torch.tensor(w.astype(np.float64), requires_grad=True)
    bt = torch.tensor(bias.astype(np.float64), requires_grad=True)
    pt = torch.sigmoid(ht @ wt + bt).reshape(-1); tt = torch.tensor(t.astype(np.float64))
    loss = O.cls_loss(tt, pt, cw); f1 = O.cls_f1(tt, pt)
    assert np.abs(p.cpu().numpy() - pt.detach().numpy()).max() < 2e-6
    o = out.cpu().numpy()
    assert abs(o[0] - float(loss)) < 2e-5 * max(1, float(loss)) and abs(o[1] - float(f1)) < 1e-6
    loss.backward()
    dh, dw, db, db1 = ops.z(b, n), ops.z(n), ops.z(1), ops.z(n)
    ops.ck(ops.lib.unet_cls_head_bwd(ops.h, ops.d(h).data_ptr(), ops.d(w).data_ptr(), p.data_ptr(), ops.d(t).data_ptr(), cw[0], cw[1], float(b), 0.4,
                                     dh.data_ptr(), dw.data_ptr(), db.data_ptr(), db1.data_ptr(), b, n, ops.s), "cls head bwd")
    want_dh = ht.grad.numpy() * (h > 0) / 0.6                              # dL/da through dropout(relu(a)): kept & positive -> 1/(1-rate)
    assert relerr(dh.cpu().numpy(), want_dh) < 2e-5 and relerr(dw.cpu().numpy(), wt.grad.numpy().reshape(-1)) < 2e-5
    assert relerr(db.cpu().numpy(), bt.grad.numpy()) < 2e-5 and relerr(db1.cpu().numpy(), want_dh.sum(0)) < 2e-5


def make(h, w=None, **kw):
    from covidseg_amd.engine import HipUNet
    kw.setdefault("dropout_rate", 0.0)
    return HipUNet(h, w or h, 1, arch="classifier", **kw)


def rand_weights(seed, hw):
    rng = np.random.default_rng(seed)
    wts = O.cls_init_weights(seed, 1, hw)
    for k in wts:
        if k.endswith("/bias") or k.endswith("/beta"):
            wts[k] = (rng.standard_normal(wts[k].shape) * 0.1).astype(np.float32)
        if k.endswith("/gamma"):
            wts[k] = rng.uniform(0.5, 1.5, wts[k].shape).astype(np.float32)
    return wts


def count_flips(eng, n, acts, names, thr=1e-5):
    """ReLU pre-activations the fp32 engine and the fp64 oracle put on different sides of 0 (each one perturbs upstream gradients
    by ~1e-3 relative)."""
    flips = 0
    for nm in names:
        a, b = eng.tap(n, nm), acts[nm]
        flips += int(((a > 0) != (b > 0)).sum())
    return flips


@pytest.mark.parametrize("algo", [0, 1, "fold16"])          # fold16: UNET_OPT_BN_FOLD = 3, the 16-channel first block folded too (opt-in)
@pytest.mark.parametrize("hw,n", [((32, 32), 4), ((24, 40), 5), ((96, 128), 24)])
def test_model_fwd_bwd_all_grads(hw, n, algo):
    h, w_ = hw
    opts = None
    if algo == "fold16":
        algo, opts = 0, {"bn_fold": 3}
    rng = np.random.default_rng(h + n)
    wts = rand_weights(h, hw)
    x = rng.random((n, h, w_, 1)).astype(np.float32); y = (rng.random(n) > 0.5).astype(np.float32)
    cw = (0.8, 1.4)
    r = O.cls_loss_and_grads(wts, x, y, class_weights=cw, dtype=torch.float64, want_acts=True)
    eng = make(h, w_, conv_algo=algo, options=opts)
    eng.set_weights(wts); eng.set_class_weights(*cw)
    ld = eng.forward_backward(x, y).cpu().numpy()
    if opts:
        names = [o[0] for o in eng.op_profile(n, 0)] + [o[0] for o in eng.op_profile(n, 1)]
        assert "bn_fold_prepare:c1b" in names and "conv3x3_dgrad_bn_bwd:c1b" in names and "bn_apply:bn1a" not in names, names
        ld = eng.forward_backward(x, y).cpu().numpy()
    assert abs(ld[0] - r["loss"]) < 1e-5 and abs(ld[1] - r["f1"]) < 1e-6
    convs = [f"c{k}{ab}" for k in (1, 2, 3) for ab in "ab"]
    for name in convs + ["bn1a", "bn2b", "p1", "p3"]:
        assert relerr(eng.tap(n, name), r["acts"][name]) < 2e-5, name
    assert relerr(eng.tap(n, "h1").reshape(n, 32), r["acts"]["h1"]) < 2e-5
    # ReLU pre-activations the fp32 engine and the fp64 oracle put on different sides of 0 are discontinuities of the gradient, not arithmetic
    # errors: the gradient reference is the oracle evaluated on the ENGINE's sign pattern (z * mask; identical wherever the signs agree), so the
    # tight tolerance holds with or without flips (the 96 x 128 x 24 case has them)
    flips = count_flips(eng, n, r["acts"], convs)
    assert flips <= 1e-5 * sum(r["acts"][c].size for c in convs) + 8, flips
    if flips:
        # ... and on the engine's max-pool choices: a 2 x 2 window whose two largest entries differ by less than the fp32 round-off routes its gradient
        # to the other element -- the same kind of discontinuity (ONE such window in p3 moves every upstream gradient of this case by 3e-4)
        r = O.cls_loss_and_grads(wts, x, y, class_weights=cw, dtype=torch.float64, relu_masks={c: (eng.tap(n, c) > 0).astype(np.float64) for c in convs},
                                 pool_sel={f"p{k}": O.pool_selection(eng.tap(n, f"bn{k}b")) for k in (1, 2, 3)})
        assert abs(ld[0] - r["loss"]) < 1e-5
    tol = 3e-4
    g = eng.get_grads()
    assert set(g) == set(r["grads"])
    for k in g:
        assert relerr(g[k], r["grads"][k]) < tol, (k, flips)
    eng.set_weights(wts); eng.set_class_weights(1.0, 1.0)
    p, l2 = eng.predict_batch(x, y)
    with torch.no_grad():
        pw = O.cls_forward(wts, x, training=False, dtype=torch.float64)[0]
        lw = float(O.cls_loss(torch.as_tensor(y, dtype=torch.float64), pw))
    assert p.shape == (n, 1) and np.abs(p.cpu().numpy().reshape(-1) - pw.numpy()).max() < 2e-5 and abs(l2.cpu().numpy()[0] - lw) < 1e-5


def test_dropout_mask_and_trajectory():
    n, h = 8, 32
    rng = np.random.default_rng(9)
    wts = rand_weights(9, (h, h))
    x = rng.random((n, h, h, 1)).astype(np.float32); y = (rng.random(n) > 0.5).astype(np.float32)
    eng = make(h, dropout_rate=0.4, seed=3)
    eng.set_weights(wts)
    ld = eng.forward_backward(x, y, training_dropout=True).cpu().numpy()
    h1 = eng.tap(n, "h1").reshape(n, 32)
    keep = (h1 != 0).astype(np.float64)                        # dropped or ReLU-dead units are both exact zeros; either way no gradient
    r = O.cls_loss_and_grads(wts, x, y, keep_mask=keep, dtype=torch.float64, want_acts=True)
    # the oracle's pre-dropout activations tell which zeros are ReLU zeros: the rest is the Dropout(0.4) mask
    alive = np.maximum(r["acts"]["p3"].reshape(n, -1) @ wts["fc1/kernel"].astype(np.float64) + wts["fc1/bias"], 0) > 1e-6
    assert 0.4 < keep[alive].mean() < 0.8
    assert abs(ld[0] - r["loss"]) < 1e-5
    g = eng.get_grads()
    convs = [f"c{k}{ab}" for k in (1, 2, 3) for ab in "ab"]
    tol = 3e-4 if count_flips(eng, n, r["acts"], convs) == 0 else 2e-2
    for k in g:
        assert relerr(g[k], r["grads"][k]) < tol, k
    # optimizer trajectory, dropout off
    tr = O.ClsOracleTrainer({k: v.astype(np.float64) for k, v in wts.items()}, torch.float64)
    eng2 = make(h); eng2.set_weights(wts)
    for step in range(3):
        a = eng2.train_batch(x, y).cpu().numpy(); b = tr.train_step(x, y)
        assert abs(a[0] - b[0]) < 3e-4 and abs(a[1] - b[1]) < 1e-6, (step, a, b)


def test_reference_size_224_and_runner(tmp_path, capsys):
    from covidseg_amd.classifier import ClassifierModel
    from covidseg_amd.data import synthetic_classification
    from covidseg_amd.runners import runner_classification
    x, y = synthetic_classification(12, 224, seed=0)
    m = ClassifierModel(224, 1, seed=1)
    assert m.count_params() == 1_678_385
    m.verbose = 0; m.compile()
    w0 = m.get_weights()
    ev = m.evaluate(x, y, batch_size=8)
    tr = O.ClsOracleTrainer(w0, torch.float32)
    ref = tr.evaluate(x, y, batch_size=8)
    assert abs(ev[0] - ref["loss"]) < 2e-5 and abs(ev[1] - ref["f1"]) < 1e-6
    assert np.abs(m.predict(x[:5]).reshape(-1) - tr.predict(x[:5])).max() < 2e-5
    a = m.backend.train_batch(x[:8], y[:8], False).cpu().numpy(); b = tr.train_step(x[:8], y[:8])
    assert abs(a[0] - b[0]) < 2e-5
    xs, ys = synthetic_classification(24, 32, seed=2)
    out = runner_classification(data=(xs, ys), epochs=3, batch_size=8, workdir=str(tmp_path), verbose=0)
    txt = capsys.readouterr().out
    for s in ("Best saved AUCROC on validation set :", "test loss:", "Accuracy:", "F1 score:"):
        assert s in txt, s
    assert os.path.exists(tmp_path / "best_val_auc_weights.h5") and os.path.exists(tmp_path / "covid_weights_val_loss.hdf5")
    assert 0.0 <= out["best_val_auc"] <= 1.0 and len(out["predictions"]) == 8
