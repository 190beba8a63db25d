"""-m gpu: the persistent two-half schedule of the shallow conv3x3 launches (csrc/kernels_conv_pp.hip; context option CONV_PP) through the SAME C-ABI entry points
as every other conv test, against torch-CPU float64 with the per-element bound of the h2 kernels, and against conv_h2_kernel on the same inputs (T1:859-860, 910-911).
CONV_PP = 2 forces the schedule onto launches too small to fill its persistent grid, so ragged shapes (tiles that overhang, XCDs without a tile, halves without a tile)
are covered; the full-size launch (512 x 512, what the option exists for) runs with CONV_PP = 1."""
import numpy as np
import pytest
import torch

from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5


def T64(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float64)


class PPOps:
    def __init__(self, value):
        from gpu_util import Ops
        from covidseg_amd import _lib
        self.base = Ops()
        self.lib = self.base.lib
        self.ctx = _lib.Context.get(torch.cuda.current_device(), {"conv_pp": value})
        self.h = self.ctx.handle
        self.d, self.z, self.wws = self.base.d, self.base.z, self.base.wws

    @property
    def s(self):
        return torch.cuda.current_stream().cuda_stream

    def ck(self, rc, what=""):
        self.ctx.check(rc, what)
        torch.cuda.synchronize()


@pytest.fixture(scope="module")
def pp():
    return PPOps(2)


SHAPES = [(2, 16, 32), (1, 8, 8), (3, 33, 70), (1, 64, 48), (2, 7, 100), (16, 40, 64), (1, 9, 264)]


@pytest.mark.parametrize("shape", SHAPES)
def test_pp_forward_bias_relu_statistics_and_sign_bits(pp, shape):
    from gpu_util import relerr, conv_abs_sums, elem_ratio
    n, h, w = shape
    ci = co = 32
    rng = np.random.default_rng(n * 1000 + h * 10 + w)
    x = rng.standard_normal((n, h, w, ci)).astype(np.float32); k = (rng.standard_normal((3, 3, ci, co)) * 0.2).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    assert pp.lib.unet_ctx_get_option(pp.h, 13) == 2
    for relu in (1, 0):
        y = pp.z(n, h, w, co)
        pp.ck(pp.lib.unet_conv3x3_fwd(pp.h, pp.d(x).data_ptr(), pp.d(k).data_ptr(), pp.d(b).data_ptr(), y.data_ptr(), n, h, w, ci, co, relu, 0.0, 0, 0, pp.wws(ci, co), pp.s), "conv fwd")
        want = O.conv3x3_bias_relu(T64(x), T64(k), T64(b), relu=bool(relu)).numpy()
        assert relerr(y.cpu().numpy(), want) < TOL
        if relu == 0:
            a = conv_abs_sums(x, k, np.zeros((n, h, w, co), np.float32), with_floor=False)
            assert elem_ratio(y.cpu().numpy(), want, a["y_a1"] + np.abs(b)[None, None, None, :]) <= 1.0
    # BatchNorm statistics from the epilogue (unet_request_bn_stats -> unet_bn_stats folds the slot copies without reading the tensor) and, where the width allows, sign bits
    y = pp.z(n, h, w, co); sums = pp.z(2 * co, dtype=torch.float64)
    bits = None
    if w % 8 == 0:
        bits = torch.full((n * h * w * co // 64,), -1, dtype=torch.int64, device="cuda")
        pp.ck(pp.lib.unet_request_relu_bits(pp.h, bits.data_ptr()), "arm bits")
    pp.ck(pp.lib.unet_request_bn_stats(pp.h, co), "arm stats")
    pp.ck(pp.lib.unet_conv3x3_fwd(pp.h, pp.d(x).data_ptr(), pp.d(k).data_ptr(), pp.d(b).data_ptr(), y.data_ptr(), n, h, w, ci, co, 1, 0.0, 0, 0, pp.wws(ci, co), pp.s), "conv fwd + stats")
    pp.ck(pp.lib.unet_bn_stats(pp.h, y.data_ptr(), co, sums.data_ptr(), n * h * w, co, pp.s), "bn stats")
    yv = y.cpu().numpy().astype(np.float64)
    sn = sums.cpu().numpy()
    assert np.allclose(sn[:co], yv.sum((0, 1, 2)), rtol=1e-5, atol=1e-3) and np.allclose(sn[co:], (yv * yv).sum((0, 1, 2)), rtol=1e-5, atol=1e-3)
    if bits is not None:
        words = bits.cpu().numpy().view(np.uint64).reshape(n, h, w // 8, 1, 4)
        pos = (yv > 0).reshape(n, h, w // 8, 8, 1, 8, 4)
        wantb = np.zeros((n, h, w // 8, 1, 4), np.uint64)
        for p in range(8):
            for q in range(8):
                wantb |= pos[:, :, :, p, :, q, :].astype(np.uint64) << np.uint64(p * 8 + q)
        assert (words == wantb).all()


@pytest.mark.parametrize("shape", [(2, 16, 32), (3, 33, 72), (1, 64, 48), (16, 40, 64)])
def test_pp_data_gradient_plain_and_with_the_one_bit_mask(pp, shape):
    from gpu_util import relerr, conv_abs_sums, elem_ratio, Ops
    n, h, w = shape
    ci = co = 32
    rng = np.random.default_rng(n + h + w)
    x = rng.standard_normal((n, h, w, ci)).astype(np.float32); k = (rng.standard_normal((3, 3, ci, co)) * 0.2).astype(np.float32)
    dy = (rng.standard_normal((n, h, w, co)) * 1e-6).astype(np.float32)                  # (gradient-sized values: the block exponent does the work)
    xt = T64(x).requires_grad_(True)
    O.conv3x3_bias_relu(xt, T64(k), torch.zeros(co, dtype=torch.float64), relu=False).backward(T64(dy))
    ab = conv_abs_sums(x, k, dy, with_floor=False)
    dx = pp.z(n, h, w, ci)
    pp.ck(pp.lib.unet_conv3x3_bwd_data(pp.h, pp.d(dy).data_ptr(), pp.d(k).data_ptr(), None, 0, 0.0, 0, dx.data_ptr(), pp.wws(ci, co), n, h, w, ci, co, 0, pp.s), "dgrad")
    assert relerr(dx.cpu().numpy(), xt.grad.numpy()) < TOL and elem_ratio(dx.cpu().numpy(), xt.grad.numpy(), ab["dx_a1"]) <= 1.0
    # the sign bits of x = relu(x0) in the documented layout, then the masked gradient: equal in every bit to masking the unmasked one
    pos = (x > 0).reshape(n, h, w // 8, 8, 1, 8, 4)
    words = np.zeros((n, h, w // 8, 1, 4), np.uint64)
    for p in range(8):
        for q in range(8):
            words |= pos[:, :, :, p, :, q, :].astype(np.uint64) << np.uint64(p * 8 + q)
    bits = torch.from_numpy(words.view(np.int64).reshape(-1)).cuda()
    dxm = pp.z(n, h, w, ci)
    pp.ck(pp.lib.unet_conv3x3_bwd_data(pp.h, pp.d(dy).data_ptr(), pp.d(k).data_ptr(), bits.data_ptr(), 9, 0.0, 0, dxm.data_ptr(), pp.wws(ci, co), n, h, w, ci, co, 0, pp.s), "dgrad bits")
    assert np.array_equal(dxm.cpu().numpy(), np.where(x > 0, dx.cpu().numpy(), np.float32(0)))
    # conv_h2_kernel on the same launch: the two schedules agree to the rounding of their (different) block exponents
    base = Ops()
    dx0 = base.z(n, h, w, ci)
    base.ck(base.lib.unet_conv3x3_bwd_data(base.h, base.d(dy).data_ptr(), base.d(k).data_ptr(), None, 0, 0.0, 0, dx0.data_ptr(), base.wws(ci, co), n, h, w, ci, co, 0, base.s), "dgrad h2")
    assert relerr(dx.cpu().numpy(), dx0.cpu().numpy()) < 2e-6


def test_pp_full_size_launch_takes_the_schedule_by_itself():
    """512 x 512, batch 2: 2048 tiles = four per half-workgroup -- CONV_PP = 1 takes it without being forced.  Against float64 on a sample of rows."""
    from gpu_util import relerr
    p1 = PPOps(1)
    n, h, w, c = 2, 512, 512, 32
    rng = np.random.default_rng(3)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32); k = (rng.standard_normal((3, 3, c, c)) * 0.1).astype(np.float32); b = rng.standard_normal(c).astype(np.float32)
    y = p1.z(n, h, w, c)
    p1.ck(p1.lib.unet_conv3x3_fwd(p1.h, p1.d(x).data_ptr(), p1.d(k).data_ptr(), p1.d(b).data_ptr(), y.data_ptr(), n, h, w, c, c, 1, 0.0, 0, 0, p1.wws(c, c), p1.s), "conv fwd")
    want = O.conv3x3_bias_relu(T64(x), T64(k), T64(b), relu=True).numpy()
    assert relerr(y.cpu().numpy(), want) < TOL


@pytest.mark.parametrize("shape", [(2, 16, 32), (3, 33, 72), (1, 64, 48)])
def test_pp_data_gradient_behind_the_heads_stream(pp, shape):
    """The last conv3x3's data gradient from the 8-byte-per-pixel stream {dz, 32 mask bits} (T1:911-913 backwards; unet_conv3x3_bwd_data_dzm) on the persistent schedule:
    against float64 from the same stream, and against conv_h2_kernel's EPI 3 instance."""
    from gpu_util import relerr, Ops
    n, h, w = shape
    c = 32
    rng = np.random.default_rng(11 + h + w)
    k3 = (rng.standard_normal((3, 3, c, c)) * 0.2).astype(np.float32); kh = rng.standard_normal(c).astype(np.float32)
    dz = (rng.standard_normal((n, h, w)) * 1e-7).astype(np.float32); mbits = rng.integers(0, 2 ** 32, (n, h, w), dtype=np.uint64).astype(np.uint32)
    st = np.stack([dz.view(np.uint32), mbits], -1).reshape(-1).copy()
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    pos = (x > 0).reshape(n, h, w // 8, 8, 1, 8, 4)
    words = np.zeros((n, h, w // 8, 1, 4), np.uint64)
    for p in range(8):
        for q in range(8):
            words |= pos[:, :, :, p, :, q, :].astype(np.uint64) << np.uint64(p * 8 + q)
    bits_in = torch.from_numpy(words.view(np.int64).reshape(-1)).cuda()
    dy = dz[..., None].astype(np.float64) * kh[None, None, None, :].astype(np.float64) * ((mbits[..., None] >> np.arange(c, dtype=np.uint32)) & 1)
    xt = T64(x).requires_grad_(True)
    O.conv3x3_bias_relu(xt, T64(k3), torch.zeros(c, dtype=torch.float64), relu=False).backward(T64(dy))
    want = xt.grad.numpy() * (x > 0)
    dzm = torch.from_numpy(st.view(np.int32)).cuda()
    dx = pp.z(n, h, w, c)
    pp.ck(pp.lib.unet_conv3x3_bwd_data_dzm(pp.h, dzm.data_ptr(), pp.d(k3).data_ptr(), pp.d(kh).data_ptr(), bits_in.data_ptr(), dx.data_ptr(), pp.wws(c, c), n, h, w, c, pp.s), "dgrad of the stream (pp)")
    assert relerr(dx.cpu().numpy(), want) < TOL
    b0 = PPOps(0)
    dx0 = b0.z(n, h, w, c)
    b0.ck(b0.lib.unet_conv3x3_bwd_data_dzm(b0.h, dzm.data_ptr(), b0.d(k3).data_ptr(), b0.d(kh).data_ptr(), bits_in.data_ptr(), dx0.data_ptr(), b0.wws(c, c), n, h, w, c, b0.s), "dgrad of the stream (h2)")
    assert relerr(dx.cpu().numpy(), dx0.cpu().numpy()) < 2e-6
