"""CPU tests: the oracle against the reference-derived golden vectors, analytic known answers and
independent restatements; the host-side helpers; the C-ABI library's exported symbols."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import unet_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_loss_goldens_from_reference_closures():
    """tests/golden/loss_goldens.npz was produced by executing the reference's dice_coeff (T1:784),
    dice_loss (T1:792) and weighted_bce_loss (T1:819, weight=1) -- the oracle must reproduce them."""
    z = np.load(os.path.join(HERE, "golden", "loss_goldens.npz"))
    n = int(z["n_cases"])
    assert n >= 8
    for i in range(n):
        t = torch.as_tensor(z[f"t{i}"], dtype=torch.float64)
        p = torch.as_tensor(z[f"p{i}"], dtype=torch.float64)
        assert float(O.dice_coeff(t, p)) == pytest.approx(float(z[f"dice_coeff{i}"]), rel=1e-12)
        assert 1 - float(O.dice_coeff(t, p)) == pytest.approx(float(z[f"dice_loss{i}"]), abs=1e-12)
        assert float(O.binary_crossentropy_mean(t, p)) == pytest.approx(float(z[f"bce_mean{i}"]), rel=1e-10, abs=1e-12)
        want = 0.5 * float(z[f"bce_mean{i}"]) + 0.5 * float(z[f"dice_loss{i}"])
        assert float(O.bce_dice_loss(t, p)) == pytest.approx(want, rel=1e-10)


def test_param_count_matches_keras_summary():
    total, train = O.count_params(1)
    assert total == 7_765_281 and train == 7_762_401          # Keras count for T1:853-916
    from covidseg_amd import weights as W
    assert W.count_params(1) == (total, train)
    assert list(W.weight_shapes(1).items()) == list(O.weight_shapes(1).items())
    kn = W.keras_names(1)
    assert kn["c5a/kernel"] == "conv2d_9/kernel:0"            # T1:1386 taps 'conv2d_9' == c5a
    assert kn["out/kernel"] == "conv2d_19/kernel:0" and kn["bn6/mean"] == "batch_normalization_5/moving_mean:0"


def test_unetpp_table_matches_reference_count():
    """U-Net++ (task1_unet_plus_plus.py:858-950): 2,209,697 parameters, same tables in product and oracle."""
    from covidseg_amd import weights as W
    assert W.count_params(1, "unetpp") == (2_209_697, 2_207_329)
    assert list(W.weight_shapes(1, "unetpp").items()) == list(O.pp_weight_shapes(1).items())
    assert W.weight_shapes(1, "unetpp")["x1_4a/kernel"] == (3, 3, 128, 32) and W.weight_shapes(1, "unetpp")["x2_3a/kernel"] == (3, 3, 192, 64)
    w = O.pp_init_weights(0)
    rng = np.random.default_rng(0)
    x = rng.random((2, 16, 16, 1)).astype(np.float32); y = (rng.random((2, 16, 16, 1)) > 0.7).astype(np.float32)
    r = O.pp_loss_and_grads(w, x, y, dtype=torch.float64)
    assert r["p"].shape == (2, 16, 16, 1) and all(np.isfinite(g).all() for g in r["grads"].values())


def _conv_loop(x, k, b):
    n, h, w, ci = x.shape
    co = k.shape[3]
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    y = np.zeros((n, h, w, co))
    for a in range(3):
        for c in range(3):
            y += np.einsum("nhwc,co->nhwo", xp[:, a:a + h, c:c + w, :], k[a, c])
    return np.maximum(y + b, 0)


def test_conv3x3_matches_loop_restatement():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 5, 7, 3)); k = rng.standard_normal((3, 3, 3, 4)); b = rng.standard_normal(4)
    got = O.conv3x3_bias_relu(torch.as_tensor(x), torch.as_tensor(k), torch.as_tensor(b)).numpy()
    np.testing.assert_allclose(got, _conv_loop(x, k, b), atol=1e-12)


def test_convT_matches_definition():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 3, 4, 5)); k = rng.standard_normal((2, 2, 6, 5)); b = rng.standard_normal(6)
    got = O.convT2x2s2_bias(torch.as_tensor(x), torch.as_tensor(k), torch.as_tensor(b)).numpy()
    want = np.zeros((2, 6, 8, 6))
    for a in range(2):
        for c in range(2):
            want[:, a::2, c::2, :] = np.einsum("nijc,oc->nijo", x, k[a, c]) + b
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_bn_known_answers():
    x = torch.full((2, 4, 4, 3), 5.0, dtype=torch.float64)
    g = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64); be = torch.tensor([0.5, -1.0, 0.0], dtype=torch.float64)
    y, mu, va = O.batchnorm(x, g, be, None, None, True)            # constant image -> y == beta
    np.testing.assert_allclose(y.numpy(), np.broadcast_to(be.numpy(), y.shape), atol=1e-12)
    assert torch.allclose(mu, torch.full((3,), 5.0, dtype=torch.float64)) and float(va.abs().max()) == 0
    rng = np.random.default_rng(3)
    x = torch.as_tensor(rng.standard_normal((3, 4, 4, 3)))
    y, mu, va = O.batchnorm(x, g, be, None, None, True)
    xn = x.numpy().reshape(-1, 3)
    np.testing.assert_allclose(y.numpy().reshape(-1, 3), (xn - xn.mean(0)) / np.sqrt(xn.var(0) + 1e-3) * g.numpy() + be.numpy(), atol=1e-12)
    nm, nv = O.bn_moving_update(np.zeros(3), np.ones(3), mu.numpy(), va.numpy(), 48)
    np.testing.assert_allclose(nm, 0.01 * xn.mean(0)); np.testing.assert_allclose(nv, 0.99 + 0.01 * xn.var(0, ddof=1))


def test_maxpool_tie_goes_to_first():
    x = torch.zeros((1, 2, 2, 1), dtype=torch.float64, requires_grad=True)
    O.maxpool2x2(x).sum().backward()
    assert x.grad.flatten().tolist() == [1.0, 0.0, 0.0, 0.0]


def test_adam_keras_form():
    p = {"a": np.array([1.0, -2.0])}; g = {"a": np.array([0.5, -0.25])}
    m = {"a": np.zeros(2)}; v = {"a": np.zeros(2)}
    O.adam_keras(p, g, m, v, 1)
    lr_t = 5e-4 * np.sqrt(1 - 0.999) / (1 - 0.9)
    mm, vv = 0.1 * g["a"], 0.001 * g["a"] ** 2
    np.testing.assert_allclose(p["a"], np.array([1.0, -2.0]) - lr_t * mm / (np.sqrt(vv) + 1e-7), rtol=1e-12)


def test_sm_scores_and_threshold_sums():
    gt = np.array([0.0, 1.0, 0.5, 1.0]); p = np.array([0.2, 0.9, 0.6, 0.4], np.float32)
    s = O.threshold_sums(gt, p, [0.5])
    assert s.tolist() == [[1.5, 2.0, 2.5]]
    sc = O.sm_scores(s[:, 0], s[:, 1], s[:, 2])
    assert sc["dice"][0] == pytest.approx((3.0 + 1e-5) / (4.5 + 1e-5)) and sc["iou"][0] == pytest.approx((1.5 + 1e-5) / (3.0 + 1e-5))
    assert sc["precision"][0] == pytest.approx((1.5 + 1e-5) / (2.0 + 1e-5)) and sc["recall"][0] == pytest.approx((1.5 + 1e-5) / (2.5 + 1e-5))


def test_forward_shapes_and_grad_flow_16px():
    w = O.init_weights(0)
    rng = np.random.default_rng(0)
    x = rng.random((2, 16, 16, 1)).astype(np.float32); y = (rng.random((2, 16, 16, 1)) > 0.7).astype(np.float32)
    r = O.loss_and_grads(w, x, y, dtype=torch.float64)
    assert r["p"].shape == (2, 16, 16, 1) and 0 < r["loss"] < 5
    assert set(r["grads"]) == set(O.trainable_names()) and all(np.isfinite(g).all() for g in r["grads"].values())


def test_train_test_split_matches_sklearn():
    from sklearn.model_selection import train_test_split as sk
    from covidseg_amd.data import train_test_split
    for n in (8, 10, 37, 100):
        x = np.arange(n); y = np.arange(n) * 10
        a = sk(x, y, test_size=0.3, random_state=42); b = train_test_split(x, y, 0.3, 42)
        for u, v in zip(a, b):
            assert u.tolist() == v.tolist()
    xtr, xva, _, _ = train_test_split(np.arange(8), np.arange(8))
    assert len(xtr) == 5 and len(xva) == 3                      # BASELINE config 1: 8 -> 5 / 3


def test_synthetic_data_is_quantised_and_deterministic():
    from covidseg_amd.data import synthetic_ct
    x, y = synthetic_ct(3, 64, seed=0); x2, y2 = synthetic_ct(3, 64, seed=0)
    assert x.shape == (3, 64, 64, 1) and x.dtype == np.float32 and (x == x2).all() and (y == y2).all()
    assert np.allclose(x * 255, np.round(x * 255)) and np.allclose(y * 255, np.round(y * 255))
    assert 0 <= x.min() and x.max() <= 1 and 0.005 < y.mean() < 0.4


def test_dropin_modules_export_reference_names():
    import importlib
    import sys
    d = os.path.join(ROOT, "covidseg_amd", "..", "one-stop-for-covid-19-infection-and-lung-segmentation-plus-classification_amd", "dropin")
    sys.path.insert(0, os.path.abspath(d))
    try:
        want = {"task1_crossval_3folds_unet": "three_fold_runner_unet_infection_segmentation",
                "task1_crossval_4folds_unet": "four_fold_runner_unet_infection_segmentation",
                "task1_preprocessing_plus_unet_with_comments": "holdout_runner_unet_infection_segmentation",
                "task1_unet_plus_plus": "holdout_runner_unetplusplus_infection_segmentation",
                "task2_covid19_classifcation": "runner_classification",
                "task3_lung_segmentation_unet": "runner_lung_segmentation"}                # app.py:7-12, 37-57
        for mod, fn in want.items():
            m = importlib.import_module(mod)
            assert m.__all__ == [fn] and callable(getattr(m, fn))
    finally:
        sys.path.pop(0)


def test_c_abi_library_exports_every_declared_symbol():
    """The C-ABI shared library loads without a GPU and exports every symbol include/unet_hip.h declares."""
    from covidseg_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "unet_hip.h")).read()
    declared = set(re.findall(r"\b(unet_[a-zA-Z0-9_]+)\s*\(", hdr)) - {"unet_last_error"} | {"unet_last_error"}
    assert os.path.exists(_lib.LIB_PATH), "build the extension first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in unet_hip.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    assert _lib.load().unet_abi_version() == _lib.ABI_VERSION == 16


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from covidseg_amd import _lib
    from covidseg_amd.keras_like import UNetModel
    with pytest.raises(_lib.UNetHipError):
        UNetModel(16)
    h = ctypes.c_void_p()
    assert _lib.load().unet_ctx_create(0, ctypes.byref(h)) == -5      # UNET_E_NODEV, no CPU fallback


def test_fullsize_measured_errors_stay_inside_the_independent_bound():
    """tests/golden/fullsize_measured.json (what the engine measured on an MI355X) only tightens the full-size gradient bounds of tests/test_gpu_fullsize.py;
    here, without a GPU: every figure in it lies inside the independent bound max(3e-4, 4 x E_k) of its fixture (E_k = the fp32-CPU evaluation's distance
    from fp64, capped at 2.5e-3), so the file cannot drift above it unnoticed."""
    import json
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import fullsize_cases as FC
    meas = json.load(open(os.path.join(here, "golden", "fullsize_measured.json")))
    seen = 0
    for name, tensors in meas.items():
        if name.startswith("_"):
            continue
        z = np.load(os.path.join(here, "golden", f"fullsize_{name}.npz"))
        for k, v in tensors.items():
            if k.startswith("_") or float(z["gnorm/" + k]) < 1e-10:          # (a ConvT bias in front of a BatchNorm: true gradient 0, both sides return noise)
                continue
            bound = max(3e-4, 4.0 * min(float(z["fp32ref_relerr/" + k]), FC.EK_CAP.get(name, 2.5e-3)))
            assert max(v.get("norm_rel", 0.0), v.get("relerr", 0.0)) <= bound, (name, k, v, bound)
            seen += 1
    assert seen > 150
