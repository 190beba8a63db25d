"""GPU versions of the two cleanly defined image steps the reference runs on every CT slice before training
(/root/reference/Scripts/task1_preprocessing_plus_unet_with_comments.py, `T1`):

    img = (img - xmin)/(xmax - xmin)                         T1:336-337   -> min_max_normalize / min_max_to_u8
    clahe_enhancer(test_img, demo)                           T1:163-202   -> clahe_enhancer (same name and arguments; `demo` plots nothing here)
    cts = cts / 255                                          T1:520       -> u8_to_unit
    cv2.resize(img[b:b+d, a:a+c], (125,250), INTER_AREA) x2 + np.concatenate   T1:354-358, 364-368 -> crop_resize_fuse
    cv2.resize(cts[i], (new_dim,new_dim), INTER_LINEAR)      T1:486-488   -> resize

    cropper(test_img, demo)                                  T1:211-273   -> cropper (same name, arguments and return triple), lung_rects

They call the C ABI (`unet_pre_*`, csrc/kernels_pre.hip); like the rest of the product there is no CPU fallback.  The contour search of
`cropper` (cv2.findContours / contourArea / boundingRect, T1:219-233) is serial border following: it runs in native host code of the same
library (csrc/host_contours.hip, `unet_pre_contours_u8`, threaded over slices); its output -- two (x, y, w, h) rectangles per slice -- feeds
crop_resize_fuse on the GPU.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


def _ctx():
    torch = _torch()
    if not torch.cuda.is_available():
        raise _lib.UNetHipError("preprocess: no GPU visible to torch; the kernels have no CPU fallback")
    return _lib.load(), _lib.Context.get(torch.cuda.current_device())


def _as_batch(a):
    a = np.asarray(a)
    if a.ndim == 2:
        return a[None], True
    if a.ndim == 3:
        return a, False
    raise ValueError("expected a [H,W] slice or a [N,H,W] stack")


def min_max_to_u8(img):
    """np.uint8((img - img.min())/(img.max() - img.min()) * 255) per slice (T1:336-337 + T1:165-166), uint8 numpy out."""
    torch = _torch(); lib, ctx = _ctx()
    a, single = _as_batch(img)
    x = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    n, h, w = x.shape
    out = torch.empty((n, h, w), dtype=torch.uint8, device="cuda")
    ws = torch.empty(max(lib.unet_pre_minmax_ws_bytes(n), 16), dtype=torch.uint8, device="cuda")
    ctx.check(lib.unet_pre_minmax_to_u8(ctx.handle, x.data_ptr(), out.data_ptr(), n, h * w, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream), "pre_minmax_to_u8")
    r = out.cpu().numpy()
    return r[0] if single else r


def clahe_u8(img_u8, clip_limit=3.0, tile_grid_size=(8, 8)):
    """cv2.createCLAHE(clipLimit, tileGridSize).apply(img) for uint8 slice(s)."""
    torch = _torch(); lib, ctx = _ctx()
    if isinstance(img_u8, torch.Tensor):                          # already on the device (clahe_enhancer): [n, h, w] uint8
        x, single = img_u8, False
    else:
        a, single = _as_batch(img_u8)
        x = torch.from_numpy(np.ascontiguousarray(a, np.uint8)).cuda()
    n, h, w = x.shape
    tx, ty = int(tile_grid_size[0]), int(tile_grid_size[1])
    out = torch.empty_like(x)
    ws = torch.empty(max(lib.unet_pre_clahe_ws_bytes(n, tx, ty), 16), dtype=torch.uint8, device="cuda")
    ctx.check(lib.unet_pre_clahe_u8(ctx.handle, x.data_ptr(), out.data_ptr(), n, h, w, float(clip_limit), tx, ty, ws.data_ptr(), ws.numel(),
                                    torch.cuda.current_stream().cuda_stream), "pre_clahe_u8")
    r = out.cpu().numpy()
    return r[0] if single else r


def clahe_enhancer(test_img, demo=0):
    """T1:163-202: a [0,1] slice (or stack) -> np.uint8(test_img*255) -> CLAHE(clipLimit 3.0, 8x8 tiles); returns the uint8 image(s)."""
    torch = _torch(); lib, ctx = _ctx()
    a, single = _as_batch(test_img)
    x = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    u8 = torch.empty(x.shape, dtype=torch.uint8, device="cuda")
    ctx.check(lib.unet_pre_unit_to_u8(ctx.handle, x.data_ptr(), u8.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream), "pre_unit_to_u8")
    r = clahe_u8(u8, 3.0, (8, 8))                                 # (the uint8 image stays on the device between the two kernels)
    return r[0] if single else r


def u8_to_unit(img_u8):
    """uint8 / 255 (T1:520) as float32."""
    torch = _torch(); lib, ctx = _ctx()
    x = torch.from_numpy(np.ascontiguousarray(img_u8, np.uint8)).cuda()
    out = torch.empty(x.shape, dtype=torch.float32, device="cuda")
    ctx.check(lib.unet_pre_u8_to_unit(ctx.handle, x.data_ptr(), out.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream), "pre_u8_to_unit")
    return out.cpu().numpy()


INTER_LINEAR, INTER_AREA = 1, 3           # cv2's values


def _resize_into(lib, ctx, x, rects, out, x0, dw, interpolation):
    torch = _torch()
    n, sh, sw = x.shape
    r = None
    if rects is not None:
        r = np.ascontiguousarray(np.asarray(rects, np.int64).reshape(n, 4).astype(np.int32))
    ctx.check(lib.unet_pre_resize_u8(ctx.handle, x.data_ptr(), n, sh, sw, r.ctypes.data if r is not None else None, out.data_ptr(), out.shape[1], dw, out.shape[2], x0,
                                     int(interpolation), torch.cuda.current_stream().cuda_stream), "pre_resize_u8")


def resize(img_u8, dsize, interpolation=INTER_LINEAR):
    """cv2.resize(img, dsize=(width, height), interpolation=cv2.INTER_LINEAR | cv2.INTER_AREA) for uint8 slice(s) (T1:486-488)."""
    torch = _torch(); lib, ctx = _ctx()
    a, single = _as_batch(img_u8)
    x = torch.from_numpy(np.ascontiguousarray(a, np.uint8)).cuda()
    dw, dh = int(dsize[0]), int(dsize[1])
    out = torch.empty((x.shape[0], dh, dw), dtype=torch.uint8, device="cuda")
    _resize_into(lib, ctx, x, None, out, 0, dw, interpolation)
    r = out.cpu().numpy()
    return r[0] if single else r


def crop_resize_fuse(img_u8, rects1, rects2, size=(125, 250), interpolation=INTER_AREA):
    """T1:354-358 / 364-368: per slice, the two lung rectangles (x, y, w, h) -- `all_points1[i]`, `all_points2[i]` -- cut out, each resized
    to `size` = (width 125, height 250) with INTER_AREA and put side by side: [N,H,W] uint8 -> [N,250,250] uint8 (no concatenate pass: both
    crops are written into their half of the output)."""
    torch = _torch(); lib, ctx = _ctx()
    a, single = _as_batch(img_u8)
    x = torch.from_numpy(np.ascontiguousarray(a, np.uint8)).cuda()
    dw, dh = int(size[0]), int(size[1])
    out = torch.empty((x.shape[0], dh, 2 * dw), dtype=torch.uint8, device="cuda")
    _resize_into(lib, ctx, x, rects1, out, 0, dw, interpolation)
    _resize_into(lib, ctx, x, rects2, out, dw, dw, interpolation)
    r = out.cpu().numpy()
    return r[0] if single else r


def prepare_cts(raw_slices, rects1=None, rects2=None, new_dim=None):
    """The per-slice chain of read_nii(..., 'cts') (T1:336-337 -> 348 -> 354-358 -> 486 -> 520): min-max normalise, clahe_enhancer, then -- when
    the lung rectangles are given -- crop + INTER_AREA resize + fuse and the INTER_LINEAR resize to new_dim (224 in the reference), /255.
    [N,H,W] raw (e.g. Hounsfield) slices in, [N,h,w,1] float32 in [0,1] out -- the shape the runners take."""
    u8 = clahe_u8(min_max_to_u8(raw_slices), 3.0, (8, 8))
    if rects1 is not None:
        u8 = crop_resize_fuse(u8, rects1, rects2)
    if new_dim is not None:
        u8 = resize(u8, (int(new_dim), int(new_dim)), INTER_LINEAR)
    a = u8_to_unit(u8)
    return a[..., None] if a.ndim == 3 else a[None, ..., None]


def prepare_infections(raw_masks, rects1=None, rects2=None, new_dim=None):
    """read_nii(..., 'infections') (T1:336-337, 360-368, 488, 521): min-max, np.uint8(img*255), crop + resize + fuse, resize, /255 -- the soft
    labels the loss is trained on."""
    u8 = min_max_to_u8(raw_masks)
    if rects1 is not None:
        u8 = crop_resize_fuse(u8, rects1, rects2)
    if new_dim is not None:
        u8 = resize(u8, (int(new_dim), int(new_dim)), INTER_LINEAR)
    a = u8_to_unit(u8)
    return a[..., None] if a.ndim == 3 else a[None, ..., None]


def contours(img_u8, max_contours=4096, threads=0):
    """Per uint8 slice: (areas float64 [k], rects int32 [k, 4]) of EVERY contour cv2.findContours(img, RETR_TREE, CHAIN_APPROX_SIMPLE) returns, in cv2's order
    (`[cv2.contourArea(c) for c in contours]`, `cv2.boundingRect(c)`; T1:219-220, 232-233).  Host code of the library; needs no GPU."""
    import ctypes as C
    lib = _lib.load()
    a, single = _as_batch(img_u8)
    a = np.ascontiguousarray(a, np.uint8)
    n, h, w = a.shape
    while True:
        areas = np.zeros((n, max_contours), np.float64); rects = np.zeros((n, max_contours, 4), np.int32); counts = np.zeros(n, np.int32)
        rc = lib.unet_pre_contours_u8(None, a.ctypes.data, n, h, w, max_contours, areas.ctypes.data, rects.ctypes.data, counts.ctypes.data, int(threads))
        if rc != 0:
            raise _lib.UNetHipError(f"unet_pre_contours_u8 failed with status {rc}")
        if counts.max(initial=0) <= max_contours:
            break
        max_contours = int(counts.max())                     # a very noisy slice: ask again with room for all of its contours
    out = [(areas[i, :counts[i]].copy(), rects[i, :counts[i]].copy()) for i in range(n)]
    return out[0] if single else out


def lung_rects(img_u8, threads=0):
    """The rectangle part of `cropper` (T1:219-233) for slice(s): `x = np.argsort(areas)`; the largest (`x[x.size - 1]`) and second largest
    (`x[x.size - 2]`) contour; `cv2.boundingRect` of each.  -> (rects1, rects2) int32 [N, 4] (or two 4-lists for one slice): `all_points1`,
    `all_points2` of T1:340-345, the arguments crop_resize_fuse takes.  A slice without any contour raises (the reference indexes an empty
    argsort there; T1:333 skips uniform masks before calling)."""
    a, single = _as_batch(img_u8)
    r1, r2 = np.zeros((len(a), 4), np.int32), np.zeros((len(a), 4), np.int32)
    for i, (areas, rects) in enumerate(contours(a, threads=threads)):
        x = np.argsort(areas)
        if x.size == 0:
            raise IndexError(f"cropper: slice {i} has no contour")
        r1[i], r2[i] = rects[x[x.size - 1]], rects[x[x.size - 2]]          # (one contour: numpy's x[-1] twice, as in the reference)
    return (r1[0].tolist(), r2[0].tolist()) if single else (r1, r2)


def cropper(test_img, demo=0):
    """cropper(test_img, demo) T1:211-273: a [0, 1] lung mask -> (fused uint8 [250, 250], points_lung1 [x, y, w, h], points_lung2 [p, q, r, s]).
    `np.uint8(test_img * 255)` and the crops + INTER_AREA resizes + fuse run on the GPU, the contour search in the library's host code; `demo` plots nothing."""
    torch = _torch(); lib, ctx = _ctx()
    a = np.asarray(test_img)
    x = torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
    u8 = torch.empty(x.shape, dtype=torch.uint8, device="cuda")
    ctx.check(lib.unet_pre_unit_to_u8(ctx.handle, x.data_ptr(), u8.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream), "pre_unit_to_u8")
    u8 = u8.cpu().numpy()
    r1, r2 = lung_rects(u8)
    return crop_resize_fuse(u8, np.asarray([r1], np.int32), np.asarray([r2], np.int32)), r1, r2
