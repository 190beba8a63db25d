"""-m gpu: U-Net++ (task1_unet_plus_plus.py:858-950, BASELINE config 4's model at the reference's fp32) on the HIP engine vs
the CPU oracle: forward / loss / every gradient, dropout with the engine's own masks, optimizer trajectory, runner."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(max(np.linalg.norm(a - b) - 1e-8 * np.sqrt(a.size), 0.0) / (np.linalg.norm(b) + 1e-30))


def make(h, w=None, **kw):
    from covidseg_amd.engine import HipUNet
    return HipUNet(h, w or h, 1, arch="unetpp", **kw)


def rand_weights(seed):
    rng = np.random.default_rng(seed)
    wts = O.pp_init_weights(seed=seed)
    for k in wts:
        if k.endswith("/bias") or k.endswith("/beta"):
            wts[k] = (rng.standard_normal(wts[k].shape) * 0.1).astype(np.float32)
        if k.endswith("/gamma"):
            wts[k] = rng.uniform(0.5, 1.5, wts[k].shape).astype(np.float32)
    return wts


@pytest.mark.parametrize("algo", [0, 1])
@pytest.mark.parametrize("hw,n", [((32, 32), 2), ((24, 40), 3)])
def test_fwd_bwd_all_grads(hw, n, algo):
    h, w_ = hw
    rng = np.random.default_rng(h)
    wts = rand_weights(h)
    x = rng.random((n, h, w_, 1)).astype(np.float32); y = (np.round(rng.random((n, h, w_, 1)) ** 4 * 255) / 255).astype(np.float32)
    r = O.pp_loss_and_grads(wts, x, y, dtype=torch.float64, want_acts=True)
    eng = make(h, w_, dropout_rate=0.0, conv_algo=algo)
    eng.set_weights(wts)
    ld = eng.forward_backward(x, y).cpu().numpy()
    assert abs(ld[0] - r["loss"]) < 1e-5 and abs(ld[1] - r["dice"]) < 1e-5
    if algo == 0:
        # auto algorithm: the first BatchNorm of every conv_block is folded into the block's second conv -- x1_2b's output, its kernel
        # gradient, x1_2abn's gamma / beta and everything upstream are checked through that path ("x1_2abn" below is materialised by the tap)
        bwd = [o[0] for o in eng.op_profile(n, 1)]
        assert "conv3x3_dgrad_bn_bwd:x1_2b" in bwd and "bn_bwd_apply:x1_2abn" not in bwd and "bn_bwd_apply:x1_2bbn" in bwd, bwd
    for name in ("c1a", "c1b", "bn1", "p1", "c4b", "bn4", "u1_2", "x1_2a", "x1_2abn", "x1_2b", "x1_2bbn", "x2_3bbn", "x1_3a", "x1_4bbn"):
        assert relerr(eng.tap(n, name), r["acts"][name]) < 2e-5, name
    g = eng.get_grads()
    assert set(g) == set(r["grads"])
    for k in g:                                  # ELU is C1: no ReLU-flip discontinuities -> tight tolerance on every tensor
        assert relerr(g[k], r["grads"][k]) < 3e-4, k
    # inference forward (moving statistics; the training step above advanced them, so restore the oracle's state)
    eng.set_weights(wts)
    p, l2 = eng.predict_batch(x, y)
    with torch.no_grad():
        pw = O.pp_forward(wts, x, training=False, dtype=torch.float64)[0]
    assert np.abs(p.cpu().numpy() - pw.numpy()).max() < 2e-5


def test_dropout_with_engine_masks():
    n, h = 2, 32
    rng = np.random.default_rng(5)
    wts = rand_weights(5)
    x = rng.random((n, h, h, 1)).astype(np.float32); y = (rng.random((n, h, h, 1)) > 0.7).astype(np.float32)
    eng = make(h, dropout_rate=0.25, seed=11)          # any rate > 0 switches the reference's fixed .2 / .4 dropouts on
    eng.set_weights(wts)
    ld = eng.forward_backward(x, y, training_dropout=True).cpu().numpy()
    drops = ["c1a", "c2a", "c3a", "c4a"] + [f"{nm}{ab}" for nm in ("x1_2", "x2_2", "x1_3", "x3_2", "x2_3", "x1_4") for ab in "ab"]
    masks = {k: (eng.tap(n, k) != 0).astype(np.float32) for k in drops}
    for k in ("c1a", "c3a"):
        assert 0.7 < masks[k].mean() < 0.9, (k, masks[k].mean())          # Dropout(0.2) UPP:877
    for k in ("x1_2a", "x2_3b"):
        assert 0.5 < masks[k].mean() < 0.7, (k, masks[k].mean())          # Dropout(0.4) UPP:858
    r = O.pp_loss_and_grads(wts, x, y, keep_masks=masks, dtype=torch.float64)
    assert abs(ld[0] - r["loss"]) < 1e-5
    g = eng.get_grads()
    for k in g:
        assert relerr(g[k], r["grads"][k]) < 3e-4, k


def test_training_trajectory_and_runner(tmp_path, capsys):
    rng = np.random.default_rng(3)
    wts = O.pp_init_weights(seed=7)
    x = rng.random((4, 32, 32, 1)).astype(np.float32); y = (rng.random((4, 32, 32, 1)) > 0.8).astype(np.float32)
    tr = O.OracleTrainer({k: v.astype(np.float64) for k, v in wts.items()}, torch.float64, arch="unetpp")
    eng = make(32, dropout_rate=0.0)
    eng.set_weights(wts)
    for step in range(3):
        a = eng.train_batch(x, y).cpu().numpy(); b = tr.train_step(x, y)
        assert abs(a[0] - b[0]) < 3e-4 and abs(a[1] - b[1]) < 3e-4, (step, a, b)
    from covidseg_amd.data import synthetic_ct
    from covidseg_amd.runners import holdout_runner_unetplusplus_infection_segmentation
    xs, ys = synthetic_ct(8, 32, seed=0)
    out = holdout_runner_unetplusplus_infection_segmentation(data=(xs, ys), epochs=2, batch_size=4, workdir=str(tmp_path), verbose=0)
    txt = capsys.readouterr().out
    assert "test loss, test dice coefficient:" in txt and "We just checked for" in txt and len(out["new_dices"]) == len(np.arange(0.40, 0.50, 0.001))
    assert out["model"].count_params() == 2_209_697 and os.path.exists(tmp_path / "unet_covid_weights_dice_coeff.hdf5")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_level_one_tensors_live_in_their_first_concat_and_gradient_slices_are_summed_once(dtype):
    """UPP:884-921: c1 / x1_2 / x1_3 go into several concats.  The program writes each straight into the slice of its FIRST consumer (no copy op for that
    pair, the later concats still get theirs) and sums the gradient slices of all consumers in ONE multi-source launch right before the tensor's own backward;
    the taps through the aliased buffers still equal the oracle's tensors."""
    n, h = 2, 32
    eng = make(h, dtype=dtype, dropout_rate=0.0)
    wts = rand_weights(3); eng.set_weights(wts)
    rng = np.random.default_rng(5)
    x = rng.random((n, h, h, 1)).astype(np.float32); y = (rng.random((n, h, h, 1)) > 0.7).astype(np.float32)
    eng.forward_backward(x, y)
    fwd = [o[0] for o in eng.op_profile(n, 0)]; bwd = [o[0] for o in eng.op_profile(n, 1)]
    for gone in ("copy_slice:c1>x1_2", "copy_slice:x1_2>x1_3", "copy_slice:x1_3>x1_4"):
        assert gone not in fwd, gone
    for kept in ("copy_slice:c1>x1_3", "copy_slice:c1>x1_4", "copy_slice:x1_2>x1_4", "copy_slice:c2>x2_2"):
        assert kept in fwd, kept
    acc = [o for o in bwd if o.startswith("accum_slice:")]
    assert "accum_slice:x1_4+x1_3+x1_2>c1" in acc and "accum_slice:x1_4+x1_3>x1_2" in acc and "accum_slice:x1_4>x1_3" in acc and len(acc) == 6, acc
    assert bwd.index("accum_slice:x1_4+x1_3>x1_2") < bwd.index("bn_bwd_stats:x1_2bbn")
    if dtype == "fp32":
        r = O.pp_loss_and_grads(wts, x, y, dtype=torch.float64, want_acts=True)
        for name, ref in (("c1", "bn1"), ("x1_2", "x1_2bbn"), ("x1_3", "x1_3bbn"), ("x1_4", "x1_4bbn")):          # (oracle names: the BatchNorm that produces the tensor)
            assert relerr(eng.tap(n, name), r["acts"][ref]) < 2e-5, name
