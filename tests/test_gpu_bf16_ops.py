"""-m gpu: the bf16-storage ops (include/unet_hip.h, ABI v5) against the CPU oracle on bf16-exact inputs.

Tolerances (written here, per the task's floating-point rule):
  * tensors the op STORES as bf16 (conv / convT outputs, data gradients): every element is an fp32-accumulated value rounded
    once to bf16 (8 significant bits): |err| <= 2^-9 relative per element  ->  norm-wise BF16_OUT = 4e-3;
  * fp32 outputs computed from bf16 inputs (weight / bias gradients): products of bf16 values are exact in fp32, only the
    summation order differs from the fp64 oracle  ->  F32_OUT = 2e-4.
"""
import numpy as np
import pytest
import torch

import oracle.unet_oracle as O

pytestmark = pytest.mark.gpu
BF16_OUT, F32_OUT = 4e-3, 2e-4


@pytest.fixture(scope="module")
def ops():
    from gpu_util import Ops
    return Ops()


def T64(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


def q(a):
    """round to bf16 (nearest even, as the kernels do) and return (device bf16 tensor, fp32 numpy of the rounded values)"""
    t = torch.from_numpy(np.ascontiguousarray(a, np.float32)).bfloat16()
    return t.cuda(), t.float().numpy()


def f32(t):
    return t.float().cpu().numpy()


CONV_SHAPES = [(1, 16, 32, 16, 32), (2, 9, 37, 32, 64), (1, 33, 34, 64, 32), (2, 16, 16, 128, 128), (1, 5, 70, 96, 64), (1, 40, 8, 48, 96),
               (1, 64, 64, 32, 32), (2, 28, 28, 16, 16), (1, 20, 36, 32, 16), (1, 8, 8, 16, 48)]          # 16-channel layers: the task-2 classifier


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_fwd_bf16(ops, shape):
    from gpu_util import relerr
    n, h, w, ci, co = shape
    rng = np.random.default_rng(hash(shape) % 1000)
    xd, x = q(rng.standard_normal((n, h, w, ci))); _, k = q(rng.standard_normal((3, 3, ci, co)) * 0.2)
    b = rng.standard_normal(co).astype(np.float32)
    for act in (1, 0):
        y = ops.z(n, h, w, co, dtype=torch.bfloat16); y.fill_(7.0)
        ops.ck(ops.lib.unet_conv3x3_fwd_bf16(ops.h, xd.data_ptr(), ops.d(k).data_ptr(), ops.d(b).data_ptr(), y.data_ptr(), n, h, w, ci, co, act, 0.0, 0, ops.wws(ci, co), ops.s), "conv fwd bf16")
        want = O.conv3x3_bias_relu(T64(x), T64(k), T64(b), relu=bool(act)).numpy()
        assert relerr(f32(y), want) < BF16_OUT
        # rounding the oracle the same way leaves only the accumulation-order flips
        wq = torch.from_numpy(want.astype(np.float32)).bfloat16().float().numpy()
        assert (f32(y) != wq).mean() < 0.02


# the data gradient swaps the roles of cin / cout: multiples of 16 on both sides
@pytest.mark.parametrize("shape", [s for s in CONV_SHAPES if s[3] != 48] + [(1, 40, 8, 96, 96), (2, 12, 20, 256, 64)])
def test_conv3x3_bwd_bf16(ops, shape):
    from gpu_util import relerr
    n, h, w, ci, co = shape
    rng = np.random.default_rng(7 + hash(shape) % 1000)
    xd, x = q(rng.standard_normal((n, h, w, ci))); _, k = q(rng.standard_normal((3, 3, ci, co)) * 0.2)
    dyd, dy = q(rng.standard_normal((n, h, w, co)))
    xt, kt = T64(x).requires_grad_(True), T64(k).requires_grad_(True)
    bt = torch.zeros(co, dtype=torch.float64, requires_grad=True)
    O.conv3x3_bias_relu(xt, kt, bt, relu=False).backward(T64(dy))
    for masked in (False, True):
        dx = ops.z(n, h, w, ci, dtype=torch.bfloat16); dx.fill_(3.0)
        ops.ck(ops.lib.unet_conv3x3_bwd_data_bf16(ops.h, dyd.data_ptr(), ops.d(k).data_ptr(), xd.data_ptr() if masked else None, 1 if masked else 0, 0.0, 0, dx.data_ptr(),
                                                  ops.wws(ci, co), n, h, w, ci, co, ops.s), "conv bwd data bf16")
        want = xt.grad.numpy() * ((x > 0) if masked else 1.0)
        assert relerr(f32(dx), want) < BF16_OUT
    nb = ops.lib.unet_conv3x3_bwd_weights_ws_bytes_bf16(n, h, w, ci, co)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    dw = ops.z(3, 3, ci, co); db = ops.z(co)
    dw.fill_(123.0); db.fill_(-7.0)                                  # must be overwritten, not accumulated
    ops.ck(ops.lib.unet_conv3x3_bwd_weights_bf16(ops.h, xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, n, h, w, ci, co, ops.s), "conv bwd w bf16")
    assert relerr(dw.cpu().numpy(), kt.grad.numpy()) < F32_OUT
    assert relerr(db.cpu().numpy(), bt.grad.numpy()) < F32_OUT


@pytest.mark.parametrize("shape", [(2, 9, 37, 32), (1, 64, 64, 32), (1, 5, 6, 64)])
def test_conv3x3_first_layer_bf16(ops, shape):
    """cin = 1: the fp32 image goes in, bf16 activations come out; the weight gradient reads the fp32 image and a bf16 gradient"""
    from gpu_util import relerr
    n, h, w, co = shape
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, h, w, 1)).astype(np.float32); k = (rng.standard_normal((3, 3, 1, co)) * 0.3).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32); dyd, dy = q(rng.standard_normal((n, h, w, co)))
    y = ops.z(n, h, w, co, dtype=torch.bfloat16)
    ops.ck(ops.lib.unet_conv3x3_first_fwd_bf16(ops.h, ops.d(x).data_ptr(), ops.d(k).data_ptr(), ops.d(b).data_ptr(), y.data_ptr(), n, h, w, co, 1, 0.0, 0, ops.s), "c1 fwd bf16")
    assert relerr(f32(y), O.conv3x3_bias_relu(T64(x), T64(k), T64(b)).numpy()) < BF16_OUT
    xt, kt, bt = T64(x), T64(k).requires_grad_(True), torch.zeros(co, dtype=torch.float64, requires_grad=True)
    O.conv3x3_bias_relu(xt, kt, bt, relu=False).backward(T64(dy))
    nb = ops.lib.unet_conv3x3_bwd_weights_ws_bytes_bf16(n, h, w, 1, co)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    dw = ops.z(3, 3, 1, co); db = ops.z(co)
    ops.ck(ops.lib.unet_conv3x3_first_bwd_weights_bf16(ops.h, ops.d(x).data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, n, h, w, co, ops.s), "c1 wgrad bf16")
    assert relerr(dw.cpu().numpy(), kt.grad.numpy()) < F32_OUT and relerr(db.cpu().numpy(), bt.grad.numpy()) < F32_OUT


@pytest.mark.parametrize("shape", [(1, 4, 4, 512, 256), (2, 8, 8, 64, 32), (1, 7, 9, 128, 64), (2, 5, 37, 64, 64), (1, 33, 34, 64, 32), (1, 16, 32, 256, 128)])
def test_convT_bf16(ops, shape):
    from gpu_util import relerr
    n, h, w, ci, co = shape
    rng = np.random.default_rng(11)
    xd, x = q(rng.standard_normal((n, h, w, ci))); _, k = q(rng.standard_normal((2, 2, co, ci)) * 0.2)
    b = rng.standard_normal(co).astype(np.float32); _, dy = q(rng.standard_normal((n, 2 * h, 2 * w, co)))
    ld = 2 * co
    xt, kt, bt = T64(x).requires_grad_(True), T64(k).requires_grad_(True), T64(b).requires_grad_(True)
    yt = O.convT2x2s2_bias(xt, kt, bt)
    cat = ops.z(n, 2 * h, 2 * w, ld, dtype=torch.bfloat16); cat.fill_(9.0)
    ops.ck(ops.lib.unet_convT2x2_fwd_bf16(ops.h, xd.data_ptr(), ops.d(k).data_ptr(), ops.d(b).data_ptr(), cat.data_ptr(), ld, n, h, w, ci, co, ops.wws(ci, co), ops.s), "convT fwd bf16")
    got = f32(cat)
    assert relerr(got[..., :co], yt.detach().numpy()) < BF16_OUT and (got[..., co:] == 9.0).all()      # only the slice is written
    yt.backward(T64(dy))
    dcat = np.full((n, 2 * h, 2 * w, ld), 5.0, np.float32); dcat[..., :co] = dy
    dcd, _ = q(dcat)
    for masked in (False, True):
        dx = ops.z(n, h, w, ci, dtype=torch.bfloat16)
        ops.ck(ops.lib.unet_convT2x2_bwd_data_bf16(ops.h, dcd.data_ptr(), ld, ops.d(k).data_ptr(), xd.data_ptr() if masked else None, dx.data_ptr(), n, h, w, ci, co, ops.wws(ci, co), ops.s), "convT bwd data bf16")
        assert relerr(f32(dx), xt.grad.numpy() * ((x > 0) if masked else 1.0)) < BF16_OUT
    nb = ops.lib.unet_convT2x2_bwd_weights_ws_bytes_bf16(n, h, w, ci, co)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    dw = ops.z(2, 2, co, ci); db = ops.z(co); dw.fill_(3.0); db.fill_(-2.0)
    ops.ck(ops.lib.unet_convT2x2_bwd_weights_bf16(ops.h, xd.data_ptr(), dcd.data_ptr(), ld, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nb, n, h, w, ci, co, ops.s), "convT bwd w bf16")
    assert relerr(dw.cpu().numpy(), kt.grad.numpy()) < F32_OUT and relerr(db.cpu().numpy(), bt.grad.numpy()) < F32_OUT


def test_dense_bf16(ops):
    """Flatten -> Dense(32): bf16 activations in, fp32 hidden units out; backward returns a bf16 dx and fp32 dW"""
    from gpu_util import relerr
    b, k, n = 6, 3136, 32
    rng = np.random.default_rng(5)
    xd, x = q(rng.standard_normal((b, k))); w = (rng.standard_normal((k, n)) * 0.05).astype(np.float32); bias = rng.standard_normal(n).astype(np.float32)
    dy = rng.standard_normal((b, n)).astype(np.float32)
    nb = ops.lib.unet_dense_ws_bytes(b, k, n); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    y = ops.z(b, n)
    ops.ck(ops.lib.unet_dense_fwd_bf16(ops.h, xd.data_ptr(), ops.d(w).data_ptr(), ops.d(bias).data_ptr(), y.data_ptr(), b, k, n, 1, 0.0, 0, ws.data_ptr(), nb, ops.s), "dense fwd bf16")
    assert relerr(y.cpu().numpy(), np.maximum(x.astype(np.float64) @ w + bias, 0)) < F32_OUT
    dx = ops.z(b, k, dtype=torch.bfloat16); dw = ops.z(k, n)
    ops.ck(ops.lib.unet_dense_bwd_bf16(ops.h, xd.data_ptr(), ops.d(w).data_ptr(), ops.d(dy).data_ptr(), dx.data_ptr(), dw.data_ptr(), b, k, n, ops.s), "dense bwd bf16")
    assert relerr(f32(dx), dy.astype(np.float64) @ w.T) < BF16_OUT and relerr(dw.cpu().numpy(), x.astype(np.float64).T @ dy) < F32_OUT


def test_unsupported_channel_counts_fail_loudly(ops):
    xd, _ = q(np.zeros((1, 8, 8, 8))); y = ops.z(1, 8, 8, 32, dtype=torch.bfloat16)
    rc = ops.lib.unet_conv3x3_fwd_bf16(ops.h, xd.data_ptr(), ops.z(3, 3, 8, 32).data_ptr(), None, y.data_ptr(), 1, 8, 8, 8, 32, 0, 0.0, 0, ops.wws(8, 32), ops.s)
    assert rc == -3          # UNET_E_SHAPE, not a silent fallback


def test_cast_round_trip(ops):
    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32)
    b = ops.z(4096, dtype=torch.bfloat16); back = ops.z(4096)
    ops.ck(ops.lib.unet_cast_f32_to_bf16(ops.h, ops.d(x).data_ptr(), b.data_ptr(), 4096, ops.s), "cast")
    ops.ck(ops.lib.unet_cast_bf16_to_f32(ops.h, b.data_ptr(), back.data_ptr(), 4096, ops.s), "cast back")
    want = torch.from_numpy(x).bfloat16()
    assert torch.equal(b.cpu(), want) and np.array_equal(back.cpu().numpy(), want.float().numpy())          # bit-exact: same RNE rounding as torch


@pytest.mark.parametrize("shape", [(2, 512, 512, 32, 32), (1, 512, 512, 64, 32), (4, 64, 64, 256, 128)])
def test_full_size_bf16_conv_agrees_with_the_fp32_kernels(ops, shape):
    """BASELINE-size tensors (the oracle would take minutes): the bf16 kernels against the fp32 GPU kernels -- which the oracle pins --
    on the same bf16-exact inputs.  Exercises the large grids (XCD block map, > 65535 workgroups, multi-unit wgrad splits)."""
    from gpu_util import relerr
    n, h, w, ci, co = shape
    g = torch.Generator(device="cuda").manual_seed(1)
    xb = torch.randn((n, h, w, ci), device="cuda", generator=g).bfloat16(); dyb = torch.randn((n, h, w, co), device="cuda", generator=g).bfloat16()
    kb = (torch.randn((3, 3, ci, co), device="cuda", generator=g) * 0.1).bfloat16().float(); b = torch.randn(co, device="cuda", generator=g)
    x32, dy32 = xb.float(), dyb.float()
    y = ops.z(n, h, w, co, dtype=torch.bfloat16); yr = ops.z(n, h, w, co)
    ops.ck(ops.lib.unet_conv3x3_fwd_bf16(ops.h, xb.data_ptr(), kb.data_ptr(), b.data_ptr(), y.data_ptr(), n, h, w, ci, co, 1, 0.0, 0, ops.wws(ci, co), ops.s), "fwd bf16")
    ops.ck(ops.lib.unet_conv3x3_fwd(ops.h, x32.data_ptr(), kb.data_ptr(), b.data_ptr(), yr.data_ptr(), n, h, w, ci, co, 1, 0.0, 0, 2, None, ops.s), "fwd fp32 direct")
    assert float((y.float() - yr).norm() / yr.norm()) < BF16_OUT
    dx = ops.z(n, h, w, ci, dtype=torch.bfloat16); dxr = ops.z(n, h, w, ci)
    ops.ck(ops.lib.unet_conv3x3_bwd_data_bf16(ops.h, dyb.data_ptr(), kb.data_ptr(), xb.data_ptr(), 1, 0.0, 0, dx.data_ptr(), ops.wws(ci, co), n, h, w, ci, co, ops.s), "dgrad bf16")
    ops.ck(ops.lib.unet_conv3x3_bwd_data(ops.h, dy32.data_ptr(), kb.data_ptr(), x32.data_ptr(), 1, 0.0, 0, dxr.data_ptr(), ops.z(int(ops.lib.unet_conv3x3_w_ws_floats(ci, co))).data_ptr(), n, h, w, ci, co, 2, ops.s), "dgrad fp32")
    assert float((dx.float() - dxr).norm() / dxr.norm()) < BF16_OUT
    nb = ops.lib.unet_conv3x3_bwd_weights_ws_bytes_bf16(n, h, w, ci, co); nr = ops.lib.unet_conv3x3_bwd_weights_ws_bytes(n, h, w, ci, co)
    ws = torch.empty(max(nb, nr, 16), dtype=torch.uint8, device="cuda")
    dw, db, dwr, dbr = ops.z(3, 3, ci, co), ops.z(co), ops.z(3, 3, ci, co), ops.z(co)
    ops.ck(ops.lib.unet_conv3x3_bwd_weights_bf16(ops.h, xb.data_ptr(), dyb.data_ptr(), dw.data_ptr(), db.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, ci, co, ops.s), "wgrad bf16")
    ops.ck(ops.lib.unet_conv3x3_bwd_weights(ops.h, x32.data_ptr(), dy32.data_ptr(), dwr.data_ptr(), dbr.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w, ci, co, 2, ops.s), "wgrad fp32")
    assert float((dw - dwr).norm() / dwr.norm()) < F32_OUT and float((db - dbr).norm() / dbr.norm()) < F32_OUT


@pytest.mark.parametrize("shape", [(2, 24, 40, 32, 64), (1, 16, 16, 64, 32)])
def test_bf16_general_epilogue_elu_dropout_and_masks_agree_with_fp32_kernels(ops, shape):
    """U-Net++ epilogue variants of the bf16 conv (ELU, fused Philox dropout, ELU / ELU+dropout data-gradient masks) against the fp32
    kernels on the same bf16-exact inputs: the keep decisions must be IDENTICAL (the counter RNG indexes elements, not bytes)."""
    n, h, w, ci, co = shape
    g = torch.Generator(device="cuda").manual_seed(3)
    xb = torch.randn((n, h, w, ci), device="cuda", generator=g).bfloat16(); kb = (torch.randn((3, 3, ci, co), device="cuda", generator=g) * 0.15).bfloat16().float()
    b = torch.randn(co, device="cuda", generator=g) * 0.1
    for rate, seed in ((0.0, 0), (0.4, 1234)):
        y = ops.z(n, h, w, co, dtype=torch.bfloat16); yr = ops.z(n, h, w, co)
        ops.ck(ops.lib.unet_conv3x3_fwd_bf16(ops.h, xb.data_ptr(), kb.data_ptr(), b.data_ptr(), y.data_ptr(), n, h, w, ci, co, 2, rate, seed, ops.wws(ci, co), ops.s), "fwd elu bf16")
        x32 = xb.float()
        ops.ck(ops.lib.unet_conv3x3_fwd(ops.h, x32.data_ptr(), kb.data_ptr(), b.data_ptr(), yr.data_ptr(), n, h, w, ci, co, 2, rate, seed, 2, None, ops.s), "fwd elu fp32")
        assert torch.equal(y == 0, yr == 0) or rate == 0.0                     # same dropped elements
        assert float((y.float() - yr).norm() / yr.norm()) < BF16_OUT
        # data gradient through this layer's input side: mask = stored (dropped-out) ELU output of the producer
        dyb = torch.randn((n, h, w, co), device="cuda", generator=g).bfloat16()
        mode = 3 if rate > 0 else 2                                             # MASK_ELU_DROP / MASK_ELU
        # the mask tensor must be a valid output of the same epilogue: reuse y of a conv whose output has `ci` channels
        mb = ops.z(n, h, w, ci, dtype=torch.bfloat16); k2 = (torch.randn((3, 3, ci, ci), device="cuda", generator=g) * 0.15).bfloat16().float()
        ops.ck(ops.lib.unet_conv3x3_fwd_bf16(ops.h, xb.data_ptr(), k2.data_ptr(), None, mb.data_ptr(), n, h, w, ci, ci, 2, rate, seed + 7, ops.wws(ci, ci), ops.s), "mask producer")
        dx = ops.z(n, h, w, ci, dtype=torch.bfloat16); dxr = ops.z(n, h, w, ci)
        ops.ck(ops.lib.unet_conv3x3_bwd_data_bf16(ops.h, dyb.data_ptr(), kb.data_ptr(), mb.data_ptr(), mode, rate, seed + 7, dx.data_ptr(), ops.wws(ci, co), n, h, w, ci, co, ops.s), "dgrad bf16")
        dy32, m32, wt = dyb.float(), mb.float(), ops.z(16 * ci * co)          # keep the fp32 copies alive: the C ABI only sees raw pointers
        ops.ck(ops.lib.unet_conv3x3_bwd_data(ops.h, dy32.data_ptr(), kb.data_ptr(), m32.data_ptr(), mode, rate, seed + 7, dxr.data_ptr(), wt.data_ptr(), n, h, w, ci, co, 2, ops.s), "dgrad fp32")
        assert float((dx.float() - dxr).norm() / dxr.norm()) < BF16_OUT
