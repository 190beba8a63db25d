"""GPU versions of the two cleanly defined image steps the reference runs on every CT slice before training
(/root/reference/Scripts/task1_preprocessing_plus_unet_with_comments.py, `T1`):

    img = (img - xmin)/(xmax - xmin)                         T1:336-337   -> min_max_normalize / min_max_to_u8
    clahe_enhancer(test_img, demo)                           T1:163-202   -> clahe_enhancer (same name and arguments; `demo` plots nothing here)
    cts = cts / 255                                          T1:520       -> u8_to_unit

They call the C ABI (`unet_pre_*`, csrc/kernels_pre.hip); like the rest of the product there is no CPU fallback.  Resizing and the
contour-based lung cropper (cv2.resize / cv2.findContours, T1:211-270, 335) stay on the host side of a user's pipeline: out of scope.
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
    r = clahe_u8(u8.cpu().numpy(), 3.0, (8, 8))
    return r[0] if single else r


def u8_to_unit(img_u8):
    """uint8 / 255 (T1:520) as float32."""
    torch = _torch(); lib, ctx = _ctx()
    x = torch.from_numpy(np.ascontiguousarray(img_u8, np.uint8)).cuda()
    out = torch.empty(x.shape, dtype=torch.float32, device="cuda")
    ctx.check(lib.unet_pre_u8_to_unit(ctx.handle, x.data_ptr(), out.data_ptr(), x.numel(), torch.cuda.current_stream().cuda_stream), "pre_u8_to_unit")
    return out.cpu().numpy()


def prepare_cts(raw_slices):
    """The per-slice chain of read_nii(..., 'cts') without the resize / crop steps (T1:336-337 -> 348 -> 520): min-max normalise,
    clahe_enhancer, /255.  [N,H,W] raw (e.g. Hounsfield) slices in, [N,H,W,1] float32 in [0,1] out -- the shape the runners take."""
    u8 = clahe_u8(min_max_to_u8(raw_slices), 3.0, (8, 8))
    a = u8_to_unit(u8)
    return a[..., None] if a.ndim == 3 else a[None, ..., None]
