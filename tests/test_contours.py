"""The contour search of `cropper` (T1:211-233, T3:213-236): cv2.findContours(RETR_TREE, CHAIN_APPROX_SIMPLE) -> contourArea -> two largest ->
boundingRect.  (1) the oracle restatement (oracle/preprocess_oracle.py, Suzuki-Abe border following with OpenCV's conventions) against
hand-derived known answers; (2) the library's native host implementation (csrc/host_contours.hip through the C ABI: needs no GPU) bit-exact
against the oracle on blobs with holes, islands, edge-touching components, 1-pixel lines, full and empty images; (3) `lung_rects` on lung-like
masks.  The GPU leg (cropper end to end: uint8 cast, rectangles, crop + INTER_AREA resize + fuse) is in tests/test_gpu_preprocess.py."""
import numpy as np
import pytest

from covidseg_amd import preprocess as PP
from oracle import preprocess_oracle as P


def blobs(h, w, seed, q=0.55):
    r = np.random.default_rng(seed)
    a = r.random((h, w))
    for _ in range(3):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / 5
    return ((a > np.quantile(a, q)) * 255).astype(np.uint8)


def lung_mask(size, seed):
    """two elliptical lungs (sometimes with a vessel hole and stray specks), like the lung masks `cropper` is called on (T1:340)"""
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size] / size
    m = np.zeros((size, size), bool)
    for cx in (0.3 + 0.04 * r.standard_normal(), 0.7 + 0.04 * r.standard_normal()):
        cy, a, b = 0.5 + 0.05 * r.standard_normal(), r.uniform(0.10, 0.16), r.uniform(0.22, 0.32)
        m |= ((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1
        m &= ~(((xx - cx) / 0.02) ** 2 + ((yy - cy + 0.05) / 0.03) ** 2 <= 1)
    for _ in range(r.integers(0, 6)):
        y, x = r.integers(0, size, 2); m[y:y + r.integers(1, 4), x:x + r.integers(1, 4)] = True
    return m.astype(np.float64)


def test_oracle_known_answers():
    a = np.zeros((12, 14), np.uint8); a[2:9, 3:11] = 255
    (c,) = P.find_contours(a)
    # an 8 x 7 pixel rectangle: border through the pixel CENTRES -> polygon 7 x 6 = 42; cv2 lists it top-left, bottom-left, bottom-right, top-right
    assert c["points"].tolist() == [[3, 2], [3, 8], [10, 8], [10, 2]] and c["area"] == 42.0 and c["rect"] == (3, 2, 8, 7) and not c["is_hole"] and c["parent"] == -1
    a[4:7, 5:8] = 0; a[5, 6] = 255                                  # a 3 x 3 hole with a 1-pixel island
    outer, hole, island = P.find_contours(a)
    assert outer["area"] == 42.0 and outer["parent"] == -1
    # a hole border runs over the FOREGROUND pixels around the hole, cutting the corners diagonally (8-connectivity): the 5 x 5 ring minus 4 half-pixel corners
    assert hole["is_hole"] and hole["parent"] == 0 and hole["area"] == 14.0 and hole["rect"] == (4, 3, 5, 5)
    assert hole["points"].tolist() == [[4, 4], [5, 3], [7, 3], [8, 4], [8, 6], [7, 7], [5, 7], [4, 6]]
    assert island["points"].tolist() == [[6, 5]] and island["area"] == 0.0 and island["rect"] == (6, 5, 1, 1) and island["parent"] == 1
    b = np.zeros((6, 6), np.uint8); b[0, 0] = 1; b[0:2, 4:6] = 1; b[3, 1] = 1; b[4, 2] = 1; b[5, 5] = 1
    cs = P.find_contours(b)                                          # top-level siblings come out newest (bottom-most) first
    assert [c["rect"] for c in cs] == [(5, 5, 1, 1), (1, 3, 2, 2), (4, 0, 2, 2), (0, 0, 1, 1)]
    assert [c["area"] for c in cs] == [0.0, 0.0, 1.0, 0.0]          # a diagonal 2-pixel line encloses nothing; 2 x 2 pixels enclose 1
    assert P.find_contours(np.zeros((5, 5), np.uint8)) == []
    (full,) = P.find_contours(np.full((4, 7), 9, np.uint8))         # components touching the image edge keep their border (zero frame outside the image)
    assert full["rect"] == (0, 0, 7, 4) and full["area"] == 18.0
    for c in cs + [outer, hole]:
        assert P.contour_area(c["points"]) == c["area"] and P.bounding_rect(c["points"]) == tuple(c["rect"])


@pytest.mark.parametrize("h,w,seed,q", [(12, 14, 1, .55), (40, 33, 2, .5), (64, 64, 3, .45), (128, 96, 4, .6), (7, 1, 5, .3), (1, 9, 6, .3), (2, 2, 7, .1),
                                        (200, 200, 8, .55), (97, 131, 9, .9), (512, 512, 10, .5)])
def test_native_host_contours_bit_exact_vs_oracle(h, w, seed, q):
    img = blobs(h, w, seed, q)
    cs = P.find_contours(img)
    areas, rects = PP.contours(img, threads=1)
    assert len(cs) == len(areas)
    assert all(c["area"] == a for c, a in zip(cs, areas)) and all(tuple(c["rect"]) == tuple(r) for c, r in zip(cs, rects))      # same values, same (cv2) order
    assert all(P.contour_area(c["points"]) == c["area"] and P.bounding_rect(c["points"]) == tuple(c["rect"]) for c in cs)         # SIMPLE polygon == walked chain


def test_batch_threads_capacity_and_degenerate_images():
    imgs = np.stack([blobs(64, 80, s, 0.5) for s in range(9)] + [np.zeros((64, 80), np.uint8), np.full((64, 80), 255, np.uint8)])
    want = [[(c["area"], tuple(c["rect"])) for c in P.find_contours(im)] for im in imgs]
    for threads in (1, 4, 0):
        got = PP.contours(imgs, threads=threads)
        assert [[(a, tuple(r)) for a, r in zip(*g)] for g in got] == want
    got = PP.contours(imgs, max_contours=3)                          # too small a capacity: the call is repeated with room for every contour
    assert [[(a, tuple(r)) for a, r in zip(*g)] for g in got] == want
    assert len(got[9][0]) == 0 and got[10][1].tolist() == [[0, 0, 80, 64]]
    with pytest.raises(IndexError):
        PP.lung_rects(imgs[9])                                       # no contour at all: the reference's x[x.size - 1] on an empty argsort
    one = np.zeros((20, 20), np.uint8); one[3:9, 4:12] = 1
    assert PP.lung_rects(one) == ([4, 3, 8, 6], [4, 3, 8, 6]) == tuple(P.lung_rects(one))       # a single contour is both "largest" and "second largest" (numpy x[-1])


@pytest.mark.parametrize("seed", range(6))
def test_lung_rects_match_oracle_on_lung_like_masks(seed):
    m = P.to_u8(lung_mask(256, seed))
    r1, r2 = PP.lung_rects(m)
    assert (r1, r2) == tuple(P.lung_rects(m))
    if seed in (0, 1):
        assert r1[2] > 30 and r1[3] > 80 and r2[2] > 30 and r2[3] > 80 and abs(r1[0] - r2[0]) > 50      # the two lungs, not specks
    stack = np.stack([P.to_u8(lung_mask(128, s)) for s in range(5)])
    b1, b2 = PP.lung_rects(stack)
    assert [(a.tolist(), b.tolist()) for a, b in zip(b1, b2)] == [tuple(P.lung_rects(s)) for s in stack]
