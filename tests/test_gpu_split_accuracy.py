"""-m gpu: the accuracy of the fp16-split (h2) convolution kernels, MEASURED and written down -- profiles/r03_split_accuracy.json is this test's output
(it writes gpurun_out/r03_split_accuracy.json; the committed copy is taken from a gpurun of this test).

For every case three numbers per op (forward, data gradient, weight gradient), each against float64 (torch CPU conv2d in double):
    l2        relative L2 error over the tensor
    elem      max over OUTPUT ELEMENTS of |err| / (2^-22 * sum_k |a_k b_k|)     -- the per-element yardstick: 1.0 = one part in 2^22 of the element's own sum of |products|
for the h2 family (UNET_ALGO_AUTO: three v_mfma_f32_32x32x16_f16 products of a block-scaled two-term split) and for the STRICT family (UNET_ALGO_MFMA:
v_mfma_f32_32x32x2_f32, exact fp32 multiply-add) on the same device buffers.  Cases:
    synthetic  post-ReLU-like activations (half zeros) x He-normal weights, K = 9 * cin = 144 ... 4608, gradients at 1e-8 (what a 512 x 512 x 16 batch produces)
    real       the tensors of the 512 x 512 x 16 golden step (tests/golden/fullsize_cases.py) at three layers: c1b (512^2 x 32 -> 32), c5b (32^2 x 512 -> 512),
               c9a (512^2 x 64 -> 32, the raw concat): activations, weights and the gradients the engine's own backward produced
Asserted: h2 elem <= 4 (the bound of tests/gpu_util.py) on every case; h2 l2 <= 2 x strict l2 + 2e-7 (the same accuracy class as the fp32 matrix pipe it replaces)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


@pytest.fixture(scope="module")
def ops():
    from gpu_util import Ops
    return Ops()


def _ref64(x, k, dy):
    """float64 conv3x3 'same' forward (no bias / activation), data gradient and weight gradient on the CPU."""
    import torch.nn.functional as F
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2).requires_grad_(True)
    kt = torch.tensor(k, dtype=torch.float64).permute(3, 2, 0, 1).requires_grad_(True)
    y = F.conv2d(xt, kt, padding=1)
    y.backward(torch.tensor(dy, dtype=torch.float64).permute(0, 3, 1, 2))
    return y.detach().permute(0, 2, 3, 1).numpy(), xt.grad.permute(0, 2, 3, 1).numpy(), kt.grad.permute(2, 3, 1, 0).numpy()


def _run(ops, x, k, dy, algo):
    n, h, w, ci = x.shape; co = k.shape[3]
    xd, kd, dyd = ops.d(x), ops.d(k), ops.d(dy)
    y = ops.z(n, h, w, co); dx = ops.z(n, h, w, ci); dw = ops.z(3, 3, ci, co); db = ops.z(co)
    ops.ck(ops.lib.unet_conv3x3_fwd(ops.h, xd.data_ptr(), kd.data_ptr(), None, y.data_ptr(), n, h, w, ci, co, 0, 0.0, 0, algo, ops.wws(ci, co), ops.s), "fwd")
    ops.ck(ops.lib.unet_conv3x3_bwd_data(ops.h, dyd.data_ptr(), kd.data_ptr(), None, 0, 0.0, 0, dx.data_ptr(), ops.wws(ci, co), n, h, w, ci, co, algo, ops.s), "dgrad")
    nb = ops.lib.unet_conv3x3_bwd_weights_ws_bytes(n, h, w, ci, co); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    ops.ck(ops.lib.unet_conv3x3_bwd_weights(ops.h, xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, n, h, w, ci, co, algo, ops.s), "wgrad")
    return y.cpu().numpy(), dx.cpu().numpy(), dw.cpu().numpy()


def _measure(ops, name, x, k, dy):
    from gpu_util import conv_abs_sums, relerr
    want = _ref64(x, k, dy)
    ab = conv_abs_sums(x, k, dy, with_floor=False)
    a1 = (ab["y_a1"], ab["dx_a1"], ab["dw_a1"])
    row = {"case": name, "shape_nhwc_cin_cout": [int(v) for v in x.shape] + [int(k.shape[3])], "K": 9 * int(k.shape[2])}
    for fam, algo in (("h2", 0), ("strict_fp32_mfma", 2)):
        got = _run(ops, x, k, dy, algo)
        for op, g, wv, a in zip(("fwd", "dgrad", "wgrad"), got, want, a1):
            err = np.abs(g.astype(np.float64) - wv)
            ok = a > 0
            row[f"{fam}/{op}/l2"] = relerr(g, wv)
            row[f"{fam}/{op}/elem"] = float((err[ok] / (2.0 ** -22 * a[ok])).max())
    return row


def test_split_accuracy_profile(ops):
    rows = []
    rng = np.random.default_rng(2026)
    for cin, cout, hw in ((16, 64, 24), (32, 32, 24), (64, 64, 24), (128, 128, 16), (256, 256, 16), (512, 512, 16)):
        x = np.maximum(rng.standard_normal((2, hw, hw, cin)), 0).astype(np.float32)
        k = (rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(np.float32)
        dy = (rng.standard_normal((2, hw, hw, cout)) * 1e-8).astype(np.float32)
        rows.append(_measure(ops, f"synthetic_cin{cin}", x, k, dy))
    # ---- the real tensors of the 512 x 512 x 16 golden step
    import fullsize_cases as FC
    from covidseg_amd.engine import HipUNet
    _, w, xin, yin = FC.build("unet_512_bs16")
    eng = HipUNet(512, 512, 1, dropout_rate=0.0)
    eng.set_weights(w)
    eng.forward_backward(xin, yin)
    for layer, src, nimg in (("c1b", "c1a", 2), ("c5b", "c5a", 4), ("c9a", "cat9", 2)):
        x = np.ascontiguousarray(eng.tap(16, src)[:nimg]); dy = np.ascontiguousarray(eng.tap(16, layer, grad=True)[:nimg])
        rows.append(_measure(ops, f"real_{layer}", x, w[layer + "/kernel"], dy))
        rows[-1]["max_abs_x"] = float(np.abs(x).max()); rows[-1]["max_abs_dy"] = float(np.abs(dy).max()); rows[-1]["rms_dy"] = float(np.sqrt((dy.astype(np.float64) ** 2).mean()))
    del eng
    torch.cuda.empty_cache()
    out = {"what": "fp32 conv3x3 on MI355X: fp16-split h2 kernels (algo 0) and strict fp32 MFMA kernels (algo 2) against float64; l2 = relative L2 error, "
                   "elem = max over output elements of |err| / (2^-22 * sum_k |a_k b_k|); written by tests/test_gpu_split_accuracy.py", "rows": rows}
    try:
        os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
        with open(os.path.join(os.path.dirname(HERE), "gpurun_out", "r03_split_accuracy.json"), "w") as f:
            json.dump(out, f, indent=1)
    except OSError:
        pass
    for r in rows:
        print(r["case"], " ".join(f"{k.split('/', 1)[1]}={v:.3g}" for k, v in r.items() if k.startswith("h2/")), "| strict", " ".join(f"{v:.3g}" for k, v in r.items() if k.startswith("strict")))
        for op in ("fwd", "dgrad", "wgrad"):
            assert r[f"h2/{op}/elem"] <= 4.0, (r["case"], op, r[f"h2/{op}/elem"])
            assert r[f"h2/{op}/l2"] <= 2.0 * r[f"strict_fp32_mfma/{op}/l2"] + 2e-7, (r["case"], op, r[f"h2/{op}/l2"], r[f"strict_fp32_mfma/{op}/l2"])
