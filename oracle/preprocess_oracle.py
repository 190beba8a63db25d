"""CPU restatement of the two cleanly defined steps in front of the U-Net path (TEST INFRASTRUCTURE ONLY -- the product never imports
this module; only tests/ compare the HIP kernels of csrc/kernels_pre.hip against it):

  * min-max normalisation of a slice                 T1:336-337  `img = (img - xmin)/(xmax - xmin)`            (float64 arithmetic)
  * clahe_enhancer(test_img, demo)                   T1:163-171  `np.uint8(test_img*255)` -> `cv2.createCLAHE(clipLimit=3.0,
                                                                  tileGridSize=(8,8)).apply(...)`
  * the later `/255`                                 T1:520

PARITY UNPINNED for the CLAHE step: OpenCV (cv2, version not pinned by the reference: `import cv2` T1:47) is not installed in this
image and its source is not under /root/reference, so `clahe_u8` restates the published algorithm of
opencv/modules/imgproc/src/clahe.cpp (CLAHE_CalcLut_Body + CLAHE_Interpolation_Body, 8-bit path) from its documentation / source as
remembered: BORDER_REFLECT_101 padding up to a multiple of the tile grid (a full extra tile row / column when one dimension already
divides and the other does not -- OpenCV pads `tiles - size % tiles`), per-tile histogram, clip at max(int(clip * area / 256), 1), excess
redistributed as `excess / 256` to every bin plus one count to every `max(256 / residual, 1)`-th bin, LUT = saturate(round-half-even(
cdf * (255 / area))) in float32, bilinear blend of the four neighbouring tile LUTs with float32 weights from `x / tile_w - 0.5`.
Every float32 operation is written out separately (no fused multiply-add), in the order of the C++ expression.
"""
import numpy as np

F = np.float32


def minmax_to_u8(img):
    """T1:336-337 followed by T1:165-166: (img - min)/(max - min) in float64, then np.uint8(x * 255) (truncation)."""
    a = np.asarray(img, np.float64)
    mn, mx = a.min(), a.max()
    return np.uint8((a - mn) / (mx - mn) * 255)


def to_u8(test_img):
    """T1:165-166: np.uint8(test_img * 255) on a [0,1] image (float64 product, truncation toward zero)."""
    return np.uint8(np.asarray(test_img, np.float64) * 255)


def u8_to_unit(img_u8):
    """T1:520: uint8 / 255 -> float64, stored as float32 by the training arrays."""
    return (np.asarray(img_u8, np.float64) / 255).astype(np.float32)


def _reflect101(i, n):
    return np.where(i < n, i, 2 * (n - 1) - i)


def clahe_u8(src, clip_limit=3.0, tiles=(8, 8)):
    """cv2.createCLAHE(clipLimit, tileGridSize=(tiles_x, tiles_y)).apply(src) for a 2-D uint8 image."""
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    tx_n, ty_n = int(tiles[0]), int(tiles[1])
    if w % tx_n == 0 and h % ty_n == 0:
        eh, ew = h, w
    else:
        eh, ew = h + (ty_n - h % ty_n), w + (tx_n - w % tx_n)
    th, tw = eh // ty_n, ew // tx_n
    area = th * tw
    yy = _reflect101(np.arange(eh), h); xx = _reflect101(np.arange(ew), w)
    ext = src[np.ix_(yy, xx)]
    lut_scale = F(255) / F(area)
    clip = max(int(clip_limit * area / 256), 1) if clip_limit > 0 else 0
    luts = np.zeros((ty_n, tx_n, 256), np.uint8)
    for ty in range(ty_n):
        for tx in range(tx_n):
            hist = np.bincount(ext[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw].ravel(), minlength=256).astype(np.int64)
            if clip > 0:
                clipped = int(np.maximum(hist - clip, 0).sum())
                hist = np.minimum(hist, clip)
                batch = clipped // 256; residual = clipped - batch * 256
                hist = hist + batch
                if residual:
                    step = max(256 // residual, 1)
                    i = 0
                    while i < 256 and residual > 0:
                        hist[i] += 1; i += step; residual -= 1
            cdf = np.cumsum(hist)
            v = np.rint(cdf.astype(F) * lut_scale)                       # float32 product, round half to even (cvRound)
            luts[ty, tx] = np.clip(v, 0, 255).astype(np.uint8)
    inv_tw, inv_th = F(1) / F(tw), F(1) / F(th)
    xs = np.arange(w).astype(F); ys = np.arange(h).astype(F)
    txf = xs * inv_tw - F(0.5); tyf = ys * inv_th - F(0.5)
    tx1 = np.floor(txf).astype(np.int64); ty1 = np.floor(tyf).astype(np.int64)
    xa = (txf - tx1.astype(F)).astype(F); ya = (tyf - ty1.astype(F)).astype(F)
    xa1 = (F(1) - xa).astype(F); ya1 = (F(1) - ya).astype(F)
    tx2 = np.minimum(tx1 + 1, tx_n - 1); tx1 = np.maximum(tx1, 0)
    ty2 = np.minimum(ty1 + 1, ty_n - 1); ty1 = np.maximum(ty1, 0)
    v = src.astype(np.int64)
    Y1, X1 = np.meshgrid(ty1, tx1, indexing="ij"); Y2, X2 = np.meshgrid(ty2, tx2, indexing="ij")
    l11 = luts[Y1, X1, v].astype(F); l12 = luts[Y1, X2, v].astype(F); l21 = luts[Y2, X1, v].astype(F); l22 = luts[Y2, X2, v].astype(F)
    XA, XA1 = xa[None, :], xa1[None, :]; YA, YA1 = ya[:, None], ya1[:, None]
    top = ((l11 * XA1).astype(F) + (l12 * XA).astype(F)).astype(F)
    bot = ((l21 * XA1).astype(F) + (l22 * XA).astype(F)).astype(F)
    res = ((top * YA1).astype(F) + (bot * YA).astype(F)).astype(F)
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def clahe_enhancer(test_img, demo=0):
    """T1:163-202 without the plots: a [0,1] float slice in, the CLAHE-enhanced uint8 slice out."""
    return clahe_u8(to_u8(test_img), 3.0, (8, 8))
