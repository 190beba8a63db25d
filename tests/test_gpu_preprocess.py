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
