"""The input path of model.fit (T1:1059-1061): a host-resident float64 set is uploaded ONCE (pinned staging, engine.HipUNet.resident), every shuffled
mini-batch is a device-side gather (unet_gather_samples through the C ABI); the fallback cuts host batches and sends them through the pinned ring.
Both must train exactly as the plain per-batch upload does."""
import numpy as np
import pytest
import torch

from covidseg_amd.data import synthetic_ct
from covidseg_amd.engine import HipUNet
from covidseg_amd.keras_like import BatchSource, UNetModel

pytestmark = pytest.mark.gpu


def test_resident_and_take_equal_numpy_indexing():
    eng = HipUNet(32, 32, 1, dropout_rate=0.0)
    rng = np.random.default_rng(0)
    a = rng.random((37, 32, 32, 1))                                   # float64 on the host, as the reference feeds it
    d = eng.resident(a)
    assert d.dtype == torch.float32 and tuple(d.shape) == a.shape
    assert np.array_equal(d.cpu().numpy(), a.astype(np.float32))
    for idx in (rng.permutation(37)[:16], np.arange(5, 21), np.array([3]), rng.permutation(37)):
        got = eng.take(d, idx).cpu().numpy()
        assert np.array_equal(got, a.astype(np.float32)[idx])
    lab = rng.random(37)                                              # [n] vectors (the classifier's labels): one float per sample
    dl = eng.resident(lab)
    assert np.array_equal(eng.take(dl, np.array([7, 2, 30])).cpu().numpy(), lab.astype(np.float32)[[7, 2, 30]])
    # the staging ring: many different host batches in a row land intact (a buffer is reused only after its copy finished)
    outs = [eng._to_dev(a[i:i + 8]) for i in range(0, 32, 8) for _ in range(3)]
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        i = (k // 3) * 8
        assert np.array_equal(o.cpu().numpy(), a[i:i + 8].astype(np.float32))


def test_resident_refuses_what_does_not_fit():
    eng = HipUNet(32, 32, 1, dropout_rate=0.0)
    a = np.zeros((4, 32, 32, 1))
    assert eng.resident(a, max_fraction=0.0) is None
    src = BatchSource(eng, a, None, device_resident=False)
    xb, yb = src(np.array([2, 0]))
    assert isinstance(xb, np.ndarray) and xb.dtype == np.float32 and yb is None


@pytest.mark.parametrize("resident", [True, False])
def test_fit_through_the_resident_set_equals_per_batch_upload(resident):
    x, y = synthetic_ct(11, 32, seed=4)
    x, y = x.astype(np.float64), y.astype(np.float64)
    hists, weights = [], []
    for mode in ("reference", "tested"):
        m = UNetModel(32, 1, seed=2, dropout_rate=0.25, options={"deterministic": 1})
        m.verbose = 0
        m.compile(lr=5e-4)
        if mode == "reference":                                       # the plain path: every batch cut on the host and handed over as an array
            h = _fit_plain(m, x, y)
        else:
            h = m.fit(x, y, batch_size=4, epochs=2, validation_data=(x[:5], y[:5]), shuffle=True, shuffle_seed=3, device_resident=resident).history
        hists.append(h); weights.append(m.get_weights())
    assert hists[0] == hists[1]                                       # bit for bit (deterministic reductions)
    for k in weights[0]:
        assert np.array_equal(weights[0][k], weights[1][k]), k


def _fit_plain(m, x, y):
    """what fit did before the resident path: x[sel] per batch"""
    be = m.backend
    hist = {"loss": [], "dice_coeff": [], "val_loss": [], "val_dice_coeff": []}
    rng = np.random.RandomState(3)
    for ep in range(2):
        order = rng.permutation(len(x))
        outs, sizes = [], []
        for i in range(0, len(x), 4):
            idx = order[i:i + 4]
            outs.append(be.train_batch(x[idx], y[idx], True)); sizes.append(len(idx))
        vals = np.stack([o.cpu().numpy().astype(np.float64) for o in outs])
        hist["loss"].append(float(np.average(vals[:, 0], weights=sizes))); hist["dice_coeff"].append(float(vals[:, 1].mean()))
        lds = [be.predict_batch(x[i:min(i + 4, 5)], y[i:min(i + 4, 5)])[1] for i in (0, 4)]
        v = np.stack([l.cpu().numpy().astype(np.float64) for l in lds])
        hist["val_loss"].append(float(np.average(v[:, 0], weights=[4, 1]))); hist["val_dice_coeff"].append(float(v[:, 1].mean()))
    return hist
