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


# ---------------------------------------------------------------------------------------------------------------------------------
# cv2.resize on uint8 single-channel images: INTER_AREA (T1:236-238, 355-367: lung crops -> 125 x 250) and INTER_LINEAR (T1:486-488:
# the fused 250 x 250 image -> new_dim x new_dim).  PARITY UNPINNED like CLAHE (cv2 absent): restates opencv/modules/imgproc/src/
# resize.cpp (hal::resize, 8-bit path) as remembered:
#   * inv_scale = dsize / ssize (double), scale = 1 / inv_scale (NOT ssize / dsize: the two can differ in the last bit);
#   * INTER_AREA with both scales >= 1: integer scales -> box sum * (1.f / area) rounded half-even (the generic resizeAreaFast_ body;
#     OpenCV's SIMD 2x2 special case rounds half-up instead -- build dependent, the generic form is what is restated);
#     otherwise the DecimateAlpha tables of computeResizeAreaTab and float32 accumulation in table order (x first, then rows);
#   * INTER_AREA with an up-scaling axis falls into the bilinear code with the "area" coefficients
#     (sx = floor(dx * scale), fx = (dx + 1) - (sx + 1) * inv_scale, clamped / wrapped as in the source);
#   * bilinear on 8-bit data is fixed point: coefficients saturate_cast<short>(c * 2048), horizontal pass exact in int32, vertical
#     pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
INTER_LINEAR, INTER_AREA = 1, 3            # the cv2 constants
_DBL_EPS = np.finfo(np.float64).eps


def _area_tab(ssize, dsize, scale):
    """computeResizeAreaTab: list of (di, si, alpha float32) in table order."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, F((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, F(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, F(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _linear_tab(ssize, dsize, scale, inv_scale, area_mode, clamp):
    """(ofs, c0, c1): source index and the two 11-bit coefficients per destination index.  `clamp` is the x-axis behaviour (index and
    fraction clamped at both ends); the y axis keeps the raw index and the rows are clipped when they are read."""
    ofs, c0, c1 = np.zeros(dsize, np.int64), np.zeros(dsize, np.int64), np.zeros(dsize, np.int64)
    for d in range(dsize):
        if not area_mode:
            f = F((d + 0.5) * scale - 0.5)
            s = int(np.floor(f))
            f = F(f - F(s))
        else:
            s = int(np.floor(d * scale))
            f = F((d + 1) - (s + 1) * inv_scale)
            f = F(0) if f <= 0 else F(f - F(np.floor(f)))
        if clamp:
            if s < 0:
                f, s = F(0), 0
            if s >= ssize - 1:
                f, s = F(0), ssize - 1
        ofs[d] = s
        c0[d] = int(np.clip(np.rint(F(F(1) - f) * F(2048)), -32768, 32767))
        c1[d] = int(np.clip(np.rint(f * F(2048)), -32768, 32767))
    return ofs, c0, c1


def resize_u8(img, dsize, interpolation=INTER_LINEAR):
    """cv2.resize(img, dsize=(width, height), interpolation=...) for a 2-D uint8 image."""
    src = np.ascontiguousarray(img, np.uint8)
    assert src.ndim == 2 and src.size > 0
    sh, sw = src.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    inv_x, inv_y = dw / sw, dh / sh
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    isx, isy = int(np.rint(scale_x)), int(np.rint(scale_y))
    fast = abs(scale_x - isx) < _DBL_EPS and abs(scale_y - isy) < _DBL_EPS
    if interpolation == INTER_LINEAR and fast and isx == 2 and isy == 2:
        interpolation = INTER_AREA
    if interpolation == INTER_AREA and scale_x >= 1 and scale_y >= 1:
        if fast:
            s = np.zeros((dh, dw), np.int64)
            for ky in range(isy):
                for kx in range(isx):
                    s += src[ky:ky + dh * isy:isy, kx:kx + dw * isx:isx]
            v = s.astype(F) * F(F(1) / F(isx * isy))                      # int * float scale, then saturate_cast<uchar> = round half even
            return np.clip(np.rint(v), 0, 255).astype(np.uint8)
        xtab, ytab = _area_tab(sw, dw, scale_x), _area_tab(sh, dh, scale_y)
        buf = np.zeros((sh, dw), F)
        s32 = src.astype(F)
        for di, si, a in xtab:                                             # buf[dxn] += S[si] * alpha, in table order, every source row at once
            buf[:, di] = buf[:, di] + s32[:, si] * a
        out = np.zeros((dh, dw), np.uint8)
        acc, prev = None, -1
        for di, si, b in ytab:
            if di != prev:
                if prev >= 0:
                    out[prev] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
                acc, prev = b * buf[si], di
            else:
                acc = acc + b * buf[si]
        out[prev] = np.clip(np.rint(acc), 0, 255).astype(np.uint8)
        return out
    if interpolation not in (INTER_LINEAR, INTER_AREA):
        raise ValueError("only INTER_LINEAR and INTER_AREA are restated")
    area_mode = interpolation == INTER_AREA
    xo, a0, a1 = _linear_tab(sw, dw, scale_x, inv_x, area_mode, True)
    yo, b0, b1 = _linear_tab(sh, dh, scale_y, inv_y, area_mode, False)
    s64 = src.astype(np.int64)
    x1 = np.minimum(xo + 1, sw - 1)
    rows = s64[:, xo] * a0[None, :] + s64[:, x1] * a1[None, :]            # horizontal pass: exact integers (pixel * 2048 scale)
    r0 = np.clip(yo, 0, sh - 1)
    r1 = np.clip(yo + 1, 0, sh - 1)
    v = ((b0[:, None] * (rows[r0] >> 4)) >> 16) + ((b1[:, None] * (rows[r1] >> 4)) >> 16)
    return np.clip((v + 2) >> 2, 0, 255).astype(np.uint8)


def crop_resize_fuse(img_u8, rect1, rect2, interpolation=INTER_AREA):
    """T1:354-358 (cts) / T1:364-368 (infections): the two lung rectangles (x, y, w, h) of a slice, each resized to 125 x 250 and put side
    by side -> 250 x 250."""
    a, b, c, d = rect1
    e, f, g, h = rect2
    i1 = resize_u8(img_u8[b:b + d, a:a + c], (125, 250), interpolation)
    i2 = resize_u8(img_u8[f:f + h, e:e + g], (125, 250), interpolation)
    return np.concatenate((i1, i2), axis=1)


# =======================================================================================================================
# cropper: cv2.findContours(img, RETR_TREE, CHAIN_APPROX_SIMPLE) -> contourArea -> two largest -> boundingRect   (T1:211-233, T3:213-236)
#
# PARITY UNPINNED (cv2 absent, see the module docstring).  Restated from the published algorithm OpenCV implements -- Suzuki & Abe,
# "Topological structural analysis of digitized binary images by border following", CVGIP 30 (1985), Algorithm 1 -- with the
# conventions of opencv/modules/imgproc/src/contours.cpp as remembered:
#   * non-zero pixels are foreground; the image is embedded in a 1-pixel zero frame (OpenCV >= 3.2 copies with copyMakeBorder, so
#     components touching the image edge keep their border pixels; coordinates are reported without the frame);
#   * raster scan; at a 0 -> 1 transition with an UNVISITED pixel an outer border starts, at a (positive label) -> 0 transition a hole
#     border starts from the last foreground pixel; 8-connected following: first neighbour searched clockwise from the background side
#     (direction codes 0 = +x, counter-clockwise in image coordinates: 1 = (+1,-1), 2 = (0,-1) ...), next neighbour counter-clockwise
#     from the one after the previous point; a pixel whose right neighbour was examined and found empty is marked with the negative label;
#   * CHAIN_APPROX_SIMPLE keeps a point only where the step direction changes;
#   * hierarchy (Suzuki's LNBD table): the parent of a new border follows from the last border met on the row; cv2 returns the tree
#     flattened in pre-order with siblings in REVERSE order of discovery (cvInsertNodeIntoTree pushes a new contour at the head of its
#     parent's child list) -- the order only matters to np.argsort's tie-breaking in `cropper`;
#   * contourArea = |shoelace| / 2 over the (integer) points in double; boundingRect = (min x, min y, max x - min x + 1, max y - min y + 1).
# =======================================================================================================================
_DX = (1, 1, 0, -1, -1, -1, 0, 1)
_DY = (0, -1, -1, -1, 0, 1, 1, 1)


def _follow_border(f, x, y, is_hole, nbd):
    """icvFetchContourEx: follow one border of label image f (framed, int32) from (x, y); marks f; returns the CHAIN_APPROX_SIMPLE points, the full chain's
    shoelace sum (2 x signed area) and the bounding box (framed coordinates)."""
    pts = []
    s_end = s = 0 if is_hole else 4
    while True:
        s = (s - 1) & 7
        if f[y + _DY[s], x + _DX[s]] != 0:
            break
        if s == s_end:
            break
    minx = maxx = x; miny = maxy = y
    if f[y + _DY[s], x + _DX[s]] == 0:                     # single-pixel component
        f[y, x] = -nbd
        return [(x, y)], 0, (x, y, x, y)
    x1, y1 = x + _DX[s], y + _DY[s]                         # first neighbour (i1)
    x3, y3 = x, y
    prev_s = s ^ 4
    shoelace = 0
    while True:
        s_end = s
        while True:
            s += 1
            x4, y4 = x3 + _DX[s & 7], y3 + _DY[s & 7]
            if f[y4, x4] != 0:
                break
        s &= 7
        if ((s - 1) & 0xFFFFFFFF) < s_end:                  # the right neighbour was examined and is empty
            f[y3, x3] = -nbd
        elif f[y3, x3] == 1:
            f[y3, x3] = nbd
        if s != prev_s:
            pts.append((x3, y3)); prev_s = s
        shoelace += x3 * y4 - y3 * x4
        minx = min(minx, x4); maxx = max(maxx, x4); miny = min(miny, y4); maxy = max(maxy, y4)
        if x4 == x and y4 == y and x3 == x1 and y3 == y1:
            break
        x3, y3 = x4, y4
        s = (s + 4) & 7
    return pts, shoelace, (minx, miny, maxx, maxy)


def find_contours(img):
    """cv2.findContours(img, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE) for a 2-D uint8 image.
    -> list of dicts in cv2's output order: points int32 [k, 2] (x, y), area (cv2.contourArea), rect (cv2.boundingRect), is_hole, parent (index into the
    same list or -1)."""
    img = np.asarray(img)
    h, w = img.shape
    f = np.zeros((h + 2, w + 2), np.int32)
    f[1:-1, 1:-1] = (img != 0)
    found = []                                              # discovery order: [pts, shoelace, bbox, is_hole, parent label]
    info = {1: (True, 1)}                                   # label -> (is_hole, parent label); the frame (label 1) counts as a hole border
    nbd = 1
    for y in range(1, h + 1):
        row = f[y]
        lnbd = 1
        x = 1
        while x <= w + 1:
            nz = np.flatnonzero(row[x:w + 2] != row[x - 1:w + 1])
            if nz.size == 0:
                break
            x += int(nz[0])
            p, prev = int(row[x]), int(row[x - 1])
            start = None
            if prev == 0 and p == 1:
                start = (x, False)
            elif p == 0 and prev >= 1:
                if prev > 1:
                    lnbd = prev
                start = (x - 1, True)
            if start is not None:
                sx, is_hole = start
                nbd += 1
                b_hole, b_parent = info[lnbd]
                parent = (b_parent if b_hole == is_hole else lnbd)          # Suzuki's table 1
                info[nbd] = (is_hole, parent)
                pts, sh, bb = _follow_border(f, sx, y, is_hole, nbd)
                found.append((pts, sh, bb, is_hole, parent, nbd))
            v = int(row[x])                                 # (after marking)
            if v not in (0, 1):
                lnbd = abs(v)
            x += 1
    # flatten: pre-order, siblings in reverse order of discovery
    children = {}
    for i, c in enumerate(found):
        children.setdefault(c[4], []).append(i)
    order = []

    def visit(label):
        for i in reversed(children.get(label, [])):
            order.append(i); visit(found[i][5])
    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 20000))
    visit(1)
    pos = {found[i][5]: k for k, i in enumerate(order)}
    out = []
    for i in order:
        pts, sh, (x0, y0, x1, y1), is_hole, parent, label = found[i]
        out.append({"points": np.array(pts, np.int32).reshape(-1, 2) - 1, "area": abs(float(sh)) * 0.5, "rect": (x0 - 1, y0 - 1, x1 - x0 + 1, y1 - y0 + 1),
                    "is_hole": is_hole, "parent": pos.get(parent, -1)})
    return out


def contour_area(points):
    """cv2.contourArea(contour): |sum(x_prev * y - y_prev * x)| / 2 in double over the closed polygon."""
    p = np.asarray(points, np.float64).reshape(-1, 2)
    if len(p) == 0:
        return 0.0
    q = np.roll(p, 1, axis=0)
    return abs(float((q[:, 0] * p[:, 1] - q[:, 1] * p[:, 0]).sum())) * 0.5


def bounding_rect(points):
    p = np.asarray(points).reshape(-1, 2)
    x0, y0 = p.min(0); x1, y1 = p.max(0)
    return int(x0), int(y0), int(x1 - x0 + 1), int(y1 - y0 + 1)


def lung_rects(img_u8):
    """The rectangle part of `cropper` (T1:219-233): contours -> areas -> np.argsort -> the largest and the second largest -> boundingRect each.
    -> (points_lung1 [x, y, w, h], points_lung2 [p, q, r, s])."""
    cs = find_contours(img_u8)
    areas = [c["area"] for c in cs]
    x = np.argsort(areas)
    if x.size == 0:
        raise IndexError("cropper: no contour in the image (the reference indexes x[x.size - 1] of an empty argsort)")
    c1, c2 = cs[int(x[x.size - 1])], cs[int(x[x.size - 2])]              # (one contour: x[-1] twice, as numpy's negative index does in the reference)
    return list(c1["rect"]), list(c2["rect"])


def cropper(test_img, demo=0):
    """cropper(test_img, demo) T1:211-273: test_img in [0, 1] (a lung mask) -> (fused uint8 [250, 250], points_lung1, points_lung2)."""
    u8 = to_u8(test_img)
    r1, r2 = lung_rects(u8)
    return crop_resize_fuse(u8, r1, r2, INTER_AREA), r1, r2
