"""-m gpu: the image steps in front of the path (csrc/kernels_pre.hip) bit-exact against oracle/preprocess_oracle.py."""
import numpy as np
import pytest

from oracle import preprocess_oracle as P

pytestmark = pytest.mark.gpu


def _ct(n, size, seed):
    from covidseg_amd.data import synthetic_ct
    return synthetic_ct(n, size, seed=seed)[0][..., 0]


@pytest.mark.parametrize("shape", [(512, 512), (256, 256), (100, 130), (64, 200), (77, 8 * 9)])
def test_clahe_bit_exact(shape):
    """divisible grids, and sizes OpenCV pads with BORDER_REFLECT_101 (one or both dimensions not a multiple of 8)"""
    from covidseg_amd import preprocess as G
    rng = np.random.default_rng(shape[0])
    imgs = [(rng.random(shape) * 255).astype(np.uint8), (np.clip(rng.normal(90, 20, shape), 0, 255)).astype(np.uint8), np.full(shape, 37, np.uint8),
            (np.arange(shape[0] * shape[1]).reshape(shape) % 251).astype(np.uint8)]
    got = G.clahe_u8(np.stack(imgs))
    for g, im in zip(got, imgs):
        assert np.array_equal(g, P.clahe_u8(im))
    for clip, grid in ((2.0, (4, 4)), (40.0, (8, 8)), (0.0, (2, 3))):           # other hyper-parameters (T1:155: "clip limit ... between 2 to 4"); 0 = plain AHE
        assert np.array_equal(G.clahe_u8(imgs[1], clip, grid), P.clahe_u8(imgs[1], clip, grid))


def test_clahe_enhancer_on_ct_slices_and_round_trip_to_unit():
    from covidseg_amd import preprocess as G
    x = _ct(3, 512, 4)                                          # [0,1] float slices quantised to k/255 like the reference's
    got = G.clahe_enhancer(x, demo=0)
    assert got.dtype == np.uint8 and got.shape == x.shape
    for g, im in zip(got, x):
        assert np.array_equal(g, P.clahe_enhancer(im))
    assert np.array_equal(G.clahe_enhancer(x[0]), got[0])      # single slice == first of the stack
    assert np.array_equal(G.u8_to_unit(got), P.u8_to_unit(got))


@pytest.mark.parametrize("shape", [(512, 512), (37, 91)])
def test_min_max_to_u8_bit_exact(shape):
    from covidseg_amd import preprocess as G
    rng = np.random.default_rng(5)
    hu = (rng.normal(-600, 400, (3,) + shape)).astype(np.float32)          # Hounsfield-like raw slices
    got = G.min_max_to_u8(hu)
    for g, im in zip(got, hu):
        assert np.array_equal(g, P.minmax_to_u8(im))
        assert g.min() == 0 and g.max() == 255


def test_prepare_cts_chain_matches_the_reference_order_of_operations():
    """min-max -> clahe_enhancer -> /255 (T1:336-337, 348, 520) composed on the GPU == composed with the CPU restatement"""
    from covidseg_amd import preprocess as G
    rng = np.random.default_rng(9)
    raw = rng.normal(-500, 350, (2, 256, 256)).astype(np.float32)
    got = G.prepare_cts(raw)
    assert got.shape == (2, 256, 256, 1) and got.dtype == np.float32
    for g, im in zip(got[..., 0], raw):
        a = im.astype(np.float64); mm = (a - a.min()) / (a.max() - a.min())
        assert np.array_equal(g, P.u8_to_unit(P.clahe_enhancer(mm)))


def _smooth(shape, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    a = 120 + 90 * np.sin(yy / rng.uniform(8, 40)) * np.cos(xx / rng.uniform(8, 40)) + rng.normal(0, 6, shape)
    return np.clip(a, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("src,dsize", [((250, 250), (224, 224)), ((250, 250), (512, 512)), ((300, 180), (125, 250)), ((91, 57), (200, 33)), ((64, 64), (64, 64)),
                                       ((500, 250), (125, 250)), ((300, 180), (90, 150)), ((7, 5), (3, 2)), ((1, 9), (4, 4))])
def test_resize_bit_exact_linear_and_area(src, dsize):
    """cv2.resize restated (T1:486-488 INTER_LINEAR, T1:236 INTER_AREA): down-scaling, up-scaling (INTER_AREA then runs the bilinear code with
    its own coefficients), integer scales (the box-sum path; 2x2 INTER_LINEAR is redirected to it), identity, tiny images"""
    from covidseg_amd import preprocess as G
    imgs = np.stack([_smooth(src, 1), (np.random.default_rng(2).random(src) * 255).astype(np.uint8), np.full(src, 201, np.uint8)])
    for interp in (P.INTER_LINEAR, P.INTER_AREA):
        got = G.resize(imgs, dsize, interp)
        assert got.shape == (3, dsize[1], dsize[0]) and got.dtype == np.uint8
        for g, im in zip(got, imgs):
            assert np.array_equal(g, P.resize_u8(im, dsize, interp)), (src, dsize, interp)
        assert np.all(got[2] == 201)                                   # weights sum to one: constants stay constant
    assert np.array_equal(G.resize(imgs[0], dsize), got_single := P.resize_u8(imgs[0], dsize, P.INTER_LINEAR)) and got_single.ndim == 2


def test_crop_resize_fuse_per_slice_rectangles_and_errors():
    from covidseg_amd import preprocess as G, _lib
    rng = np.random.default_rng(3)
    n = 100                                                            # more rectangles than one launch carries
    imgs = np.stack([_smooth((256, 256), 10 + i) for i in range(4)])[rng.integers(0, 4, n)]
    r1 = np.stack([rng.integers(0, 60, n), rng.integers(0, 40, n), rng.integers(40, 130, n), rng.integers(100, 216, n)], 1)          # narrow ones up-scale in x
    r2 = np.stack([rng.integers(120, 140, n), rng.integers(0, 30, n), rng.integers(60, 116, n), rng.integers(150, 226, n)], 1)
    got = G.crop_resize_fuse(imgs, r1, r2)
    assert got.shape == (n, 250, 250)
    for i in range(n):
        assert np.array_equal(got[i], P.crop_resize_fuse(imgs[i], tuple(r1[i]), tuple(r2[i]))), i
    with pytest.raises(_lib.UNetHipError, match="rectangle"):
        G.crop_resize_fuse(imgs[:1], [[200, 0, 100, 50]], [[0, 0, 10, 10]])
    with pytest.raises(_lib.UNetHipError, match="interpolation"):
        G.resize(imgs[0], (10, 10), 2)                                 # INTER_CUBIC is not restated


def test_full_cts_and_infection_chain_with_rectangles():
    """read_nii for 'cts' and 'infections' end to end (T1:336-368, 486-488, 520-521) on the GPU == composed with the CPU restatement"""
    from covidseg_amd import preprocess as G
    rng = np.random.default_rng(11)
    raw = rng.normal(-500, 350, (3, 512, 512)).astype(np.float32)
    msk = (rng.random((3, 512, 512)) > 0.97).astype(np.float32) * rng.integers(1, 4, (3, 1, 1))
    r1 = [(60, 90, 170, 330), (40, 100, 120, 300), (70, 80, 200, 350)]
    r2 = [(280, 95, 180, 320), (270, 100, 190, 310), (300, 60, 150, 260)]
    got = G.prepare_cts(raw, r1, r2, new_dim=224)
    gotm = G.prepare_infections(msk, r1, r2, new_dim=224)
    assert got.shape == gotm.shape == (3, 224, 224, 1)
    for i in range(3):
        a = raw[i].astype(np.float64); mm = (a - a.min()) / (a.max() - a.min())
        e = P.resize_u8(P.crop_resize_fuse(P.clahe_enhancer(mm), r1[i], r2[i]), (224, 224), P.INTER_LINEAR)
        assert np.array_equal(got[i, ..., 0], P.u8_to_unit(e))
        e = P.resize_u8(P.crop_resize_fuse(P.minmax_to_u8(msk[i]), r1[i], r2[i]), (224, 224), P.INTER_LINEAR)
        assert np.array_equal(gotm[i, ..., 0], P.u8_to_unit(e))


@pytest.mark.parametrize("seed", [0, 1, 4])
def test_cropper_end_to_end_matches_the_oracle(seed):
    """cropper(test_img, demo) T1:211-273 on a lung-like mask: uint8 cast on the GPU, contour search in the library's host code (tests/test_contours.py
    checks it bit-exact on its own), both crops INTER_AREA-resized to 125 x 250 and fused on the GPU -- against the oracle's cropper, byte for byte."""
    from test_contours import lung_mask
    m = lung_mask(256, seed)
    from covidseg_amd import preprocess as PP
    fused, r1, r2 = PP.cropper(m, demo=0)
    wf, w1, w2 = P.cropper(m, demo=0)
    assert r1 == w1 and r2 == w2 and isinstance(r1, list) and len(r1) == 4
    assert fused.shape == (250, 250) and fused.dtype == np.uint8 and np.array_equal(fused, wf)
