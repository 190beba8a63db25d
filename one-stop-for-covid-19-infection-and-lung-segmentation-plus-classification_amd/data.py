"""Injectable data for the runners: synthetic CT slices + masks and the hold-out split.

The reference acquires data with pip/Kaggle/Drive (T1:8-136) and pre-processes NIfTI volumes
(T1:163-686) into ``cts, infections : float [N,224,224,1] in [0,1]`` (uint8/255, T1:520, 678,
soft bilinear-resized labels T1:488).  None of that is on the hot path; here the arrays are
injected (``.npy``/arrays) or synthesised with the same value structure.
"""
from __future__ import annotations

import math

import numpy as np


def synthetic_ct(n: int, size: int = 512, seed: int = 0):
    """x,y float32 [n,size,size,1].  Image = 0.1 background + 4 random 2-D Gaussians + N(0,.02)
    noise, clipped, quantised to k/255.  Mask = union of 1-3 random ellipses (1-10 % area each),
    Gaussian-blurred (sigma 1 px) and quantised to k/255 (soft labels like T1:488)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    xs = np.empty((n, size, size, 1), np.float32)
    ys = np.empty((n, size, size, 1), np.float32)
    r = np.arange(-3, 4, dtype=np.float32)
    g1 = np.exp(-0.5 * r * r); g1 /= g1.sum()
    for i in range(n):
        img = np.full((size, size), 0.1, np.float32)
        for _ in range(4):
            cx, cy = rng.uniform(0.1, 0.9, 2); s = rng.uniform(0.05, 0.3); a = rng.uniform(0.2, 0.8)
            img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
        img += rng.normal(0, 0.02, img.shape).astype(np.float32)
        xs[i, :, :, 0] = np.round(np.clip(img, 0, 1) * 255) / 255
        m = np.zeros((size, size), np.float32)
        for _ in range(rng.integers(1, 4)):
            cx, cy = rng.uniform(0.2, 0.8, 2)
            area = rng.uniform(0.01, 0.10); ratio = rng.uniform(0.5, 2.0)
            ra = math.sqrt(area * ratio / math.pi); rb = math.sqrt(area / ratio / math.pi)
            th = rng.uniform(0, math.pi)
            u = (xx - cx) * math.cos(th) + (yy - cy) * math.sin(th)
            v = -(xx - cx) * math.sin(th) + (yy - cy) * math.cos(th)
            m = np.maximum(m, ((u / ra) ** 2 + (v / rb) ** 2 <= 1).astype(np.float32))
        mp = np.pad(m, 3, mode="edge")                              # separable 7-tap Gaussian blur
        m = sum(g1[k] * mp[k:k + size, 3:3 + size] for k in range(7))
        mp = np.pad(m, 3, mode="edge")
        m = sum(g1[k] * mp[3:3 + size, k:k + size] for k in range(7))
        ys[i, :, :, 0] = np.round(np.clip(m, 0, 1) * 255) / 255
    return xs, ys


def synthetic_classification(n: int, size: int = 224, seed: int = 0, channels: int = 1):
    """x float32 [n,size,size,channels], y int [n]: the synthetic CT slices above (channels > 1 -- BASELINE.json configs[4] names 224 x 224 x 3 -- : channel k is the
    slice under its own intensity window, k / 255-quantised like the first); label 1 = the slice has an infection mask, and then the
    lesion is also painted into the image (+0.25 inside the soft mask) so the classes are separable.  Mirrors how the reference
    derives its labels: y = 1 iff the infection mask of the slice is not uniform (task2_covid19_classifcation.py:413-418)."""
    xs, ms = synthetic_ct(n, size, seed)
    rng = np.random.default_rng(seed + 7919)
    y = (rng.random(n) < 0.6).astype(np.int64)
    if n >= 4:
        y[:2] = (0, 1); y[2:4] = (0, 1)                           # both classes present at least twice (stratified split needs it)
    xs = np.clip(xs + 0.25 * ms * y[:, None, None, None], 0, 1).astype(np.float32)
    if channels > 1:
        xs = np.concatenate([np.clip((xs - 0.08 * k) * (1.0 + 0.15 * k), 0, 1) for k in range(channels)], axis=3).astype(np.float32)
    return np.round(xs * 255).astype(np.float32) / 255, y


def train_test_split(x, y, test_size: float = 0.3, random_state: int = 42):
    """sklearn.model_selection.train_test_split(x, y, test_size=0.3, random_state=42) as called
    at T1:762 -- restated (ShuffleSplit): n_test = ceil(test_size*n); perm = RandomState(seed)
    .permutation(n); test = perm[:n_test]; train = perm[n_test:].  Returns x_tr, x_va, y_tr, y_va."""
    n = len(x)
    n_test = int(math.ceil(test_size * n))
    n_train = n - n_test
    if n_train < 1 or n_test < 1:
        raise ValueError(f"train_test_split: n={n} too small for test_size={test_size}")
    perm = np.random.RandomState(random_state).permutation(n)
    te, tr = perm[:n_test], perm[n_test:n_test + n_train]
    return x[tr], x[te], y[tr], y[te]


def kfold_indices(n: int, n_splits: int, random_state: int = 42):
    """sklearn.model_selection.KFold(n_splits, random_state=42, shuffle=True).split(range(n)) restated
    (task1_crossval_4folds_unet.py:1047): indices = arange(n) shuffled by RandomState(seed); fold i takes the next
    n//k (+1 for the first n%k folds) shuffled indices as TEST; train/test index arrays are returned sorted ascending
    (sklearn builds them from a boolean mask)."""
    if n_splits < 2 or n_splits > n:
        raise ValueError(f"kfold: n_splits={n_splits} invalid for n={n}")
    idx = np.arange(n)
    np.random.RandomState(random_state).shuffle(idx)
    sizes = np.full(n_splits, n // n_splits, dtype=int); sizes[: n % n_splits] += 1
    out, cur = [], 0
    for sz in sizes:
        mask = np.zeros(n, bool); mask[idx[cur:cur + sz]] = True
        out.append((np.flatnonzero(~mask), np.flatnonzero(mask)))
        cur += sz
    return out
