"""CPU oracle for the U-Net segmentation hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (torch-CPU / NumPy, fp32 or fp64) of the arithmetic the
reference executes on its hot path.  It is the *checker* for the HIP engine.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product package never does (it fails loudly without the HIP library).

Reference citations (paths relative to /root/reference/Scripts/):
  T1 = task1_preprocessing_plus_unet_with_comments.py, T3 = task3_lung_segmentation_unet.py

  graph ............ T1:853-916  (== T3:850-913)
  dice_coeff ....... T1:784-790      dice_loss T1:792-794     bce_dice_loss T1:797-799
  BCE arithmetic ... keras.losses.binary_crossentropy (imported T1:60); the identical
                     logit-form expression is spelled out in the reference's own
                     weighted_bce_loss T1:819-825
  compile / fit .... T1:1053, T1:1059-1061 (Adam(lr=5e-4), bs 32, epochs 80)
  evaluate ......... T1:1101;  threshold sweeps T1:1196-1221, 1250-1281, 1304-1330

PARITY STATUS.  The loss / metric closures (dice_coeff, dice_loss, weighted_bce_loss) are
PINNED: ``tests/golden/make_loss_goldens.py`` AST-extracts them from the reference source,
executes them against a NumPy ``K`` shim in the build container and commits input/output
vectors (tests/golden/loss_goldens.npz) that this oracle must reproduce.
Everything else -- Conv2D / BatchNormalization / MaxPooling2D / Dropout / Conv2DTranspose /
Adam / fit / evaluate / segmentation_models metrics -- lives in third-party packages that
are neither vendored under /root/reference nor installable here (keras ~2.3.x on
tensorflow 2.2.0, segmentation_models 1.0.x; versions un-pinned by the reference).  For
those the oracle restates the libraries' documented semantics:
**parity unpinned** w.r.t. the un-versioned dependencies; cross-checked here against
``torch.nn.functional`` and analytic known answers only.

Layouts follow Keras ``channels_last``: activations [N,H,W,C]; Conv2D kernel [kh,kw,Cin,Cout];
Conv2DTranspose kernel [kh,kw,Cout,Cin]; BN weights gamma, beta, moving_mean, moving_variance.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3          # keras BatchNormalization default epsilon
BN_MOMENTUM = 0.99     # keras BatchNormalization default momentum
DROPOUT_RATE = 0.25    # T1:863
ADAM_LR, ADAM_B1, ADAM_B2, ADAM_EPS = 5e-4, 0.9, 0.999, 1e-7   # T1:1053 + keras defaults
BCE_EPS = 1e-7         # keras.backend.epsilon()
SM_SMOOTH = 1e-5       # segmentation_models metric smooth default

# ---------------------------------------------------------------------------------------
# Graph description, T1:853-916.  Names: cKa/cKb 3x3 convs of block K, bnK, uK = ConvT.
# Order == Keras layer-creation order == model.get_weights() order.
# ---------------------------------------------------------------------------------------
ENC = [32, 64, 128, 256]


def layer_table(in_ch: int = 1):
    """[(name, kind, cin, cout)] in Keras creation order (T1:859-913)."""
    t = []
    c_prev = in_ch
    for k, c in enumerate(ENC, start=1):                       # T1:859-881
        t += [(f"c{k}a", "conv3", c_prev, c), (f"c{k}b", "conv3", c, c), (f"bn{k}", "bn", c, c)]
        c_prev = c
    t += [("c5a", "conv3", 256, 512), ("c5b", "conv3", 512, 512)]   # T1:883-884
    c_prev = 512
    for k, c in zip([6, 7, 8, 9], [256, 128, 64, 32]):          # T1:886-911
        t += [(f"u{k}", "convT", c_prev, c), (f"bn{k}", "bn", 2 * c, 2 * c),
              (f"c{k}a", "conv3", 2 * c, c), (f"c{k}b", "conv3", c, c)]
        c_prev = c
    t += [("out", "conv1", 32, 1)]                              # T1:913
    return t


def weight_shapes(in_ch: int = 1):
    """OrderedDict name -> shape, Keras layouts; 'X/kernel','X/bias','bnK/gamma|beta|mean|var'."""
    d = OrderedDict()
    for name, kind, cin, cout in layer_table(in_ch):
        if kind == "conv3":
            d[name + "/kernel"] = (3, 3, cin, cout); d[name + "/bias"] = (cout,)
        elif kind == "conv1":
            d[name + "/kernel"] = (1, 1, cin, cout); d[name + "/bias"] = (cout,)
        elif kind == "convT":
            d[name + "/kernel"] = (2, 2, cout, cin); d[name + "/bias"] = (cout,)
        else:
            for p in ("gamma", "beta", "mean", "var"):
                d[f"{name}/{p}"] = (cout,)
    return d


def trainable_names(in_ch: int = 1):
    return [k for k in weight_shapes(in_ch) if not (k.endswith("/mean") or k.endswith("/var"))]


def count_params(in_ch: int = 1):
    total = sum(int(np.prod(s)) for s in weight_shapes(in_ch).values())
    train = sum(int(np.prod(weight_shapes(in_ch)[k])) for k in trainable_names(in_ch))
    return total, train


def init_weights(seed: int = 0, in_ch: int = 1, dtype=np.float32):
    """Keras initialisers: he_normal (truncated normal, +-2 sigma,
    sigma = sqrt(2/fan_in)/0.87962566) for the 3x3 convs (T1:859...), glorot_uniform for ConvT
    and the 1x1 head (Keras default), zero biases, BN gamma=1 beta=0 mean=0 var=1."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, kind, cin, cout in layer_table(in_ch):
        if kind == "conv3":
            fan_in = 9 * cin
            std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
            k = rng.standard_normal((3, 3, cin, cout))
            bad = np.abs(k) > 2.0
            while bad.any():                                      # resample the tails
                k[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(k) > 2.0
            w[name + "/kernel"] = (k * std).astype(dtype)
            w[name + "/bias"] = np.zeros(cout, dtype)
        elif kind == "conv1":
            lim = math.sqrt(6.0 / (cin + cout))
            w[name + "/kernel"] = rng.uniform(-lim, lim, (1, 1, cin, cout)).astype(dtype)
            w[name + "/bias"] = np.zeros(cout, dtype)
        elif kind == "convT":
            # keras fan computation for a [kh,kw,Cout,Cin] kernel: receptive=4, fan_in=4*Cout, fan_out=4*Cin
            lim = math.sqrt(6.0 / (4 * cout + 4 * cin))
            w[name + "/kernel"] = rng.uniform(-lim, lim, (2, 2, cout, cin)).astype(dtype)
            w[name + "/bias"] = np.zeros(cout, dtype)
        else:
            w[name + "/gamma"] = np.ones(cout, dtype); w[name + "/beta"] = np.zeros(cout, dtype)
            w[name + "/mean"] = np.zeros(cout, dtype); w[name + "/var"] = np.ones(cout, dtype)
    return w


# ---------------------------------------------------------------------------------------
# Single ops (NHWC numpy/torch in, NHWC torch out).  Internally NCHW for torch functional.
# ---------------------------------------------------------------------------------------
def _t(a, dtype):
    return a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a), dtype=dtype)


def conv3x3_bias_relu(x, kernel, bias, relu=True, relu_mask=None):
    """Conv2D(C,(3,3),'relu','same') T1:859: cross-correlation, zero pad 1, +bias, ReLU.
    relu_mask ({0,1} array of the output shape): use z * mask instead of max(z, 0) -- the same function wherever sign(z) agrees with
    the mask; the GPU tests pass the ENGINE's own sign pattern so that a pre-activation which rounds to the other side of 0 in fp32
    (a discontinuity of the gradient, not an arithmetic error) does not enter the gradient comparison."""
    y = F.conv2d(x.permute(0, 3, 1, 2), kernel.permute(3, 2, 0, 1), bias, padding=1)
    y = y.permute(0, 2, 3, 1)
    if relu and relu_mask is not None:
        return y * relu_mask
    return torch.relu(y) if relu else y


def conv1x1_sigmoid(x, kernel, bias):
    """Conv2D(1,(1,1),'sigmoid') T1:913."""
    z = torch.tensordot(x, kernel[0, 0], dims=([3], [0])) + bias
    return torch.sigmoid(z)


def batchnorm(x, gamma, beta, mean, var, training):
    """BatchNormalization() T1:861: training -> biased batch statistics over (N,H,W);
    returns (y, batch_mean, batch_var_biased).  Inference -> moving statistics."""
    if training:
        mu = x.mean(dim=(0, 1, 2))
        va = ((x - mu) ** 2).mean(dim=(0, 1, 2))
    else:
        mu, va = mean, var
    y = (x - mu) / torch.sqrt(va + BN_EPS) * gamma + beta
    return y, mu, va


def bn_moving_update(mean, var, mu, va, n):
    """keras: moving = moving*momentum + batch*(1-momentum); variance uses n/(n-1) (TF fused BN)."""
    new_mean = mean * BN_MOMENTUM + mu * (1 - BN_MOMENTUM)
    new_var = var * BN_MOMENTUM + va * (n / max(n - 1.0, 1.0)) * (1 - BN_MOMENTUM)
    return new_mean, new_var


def maxpool2x2(x, sel=None):
    """MaxPooling2D((2,2)) T1:862: stride 2, 'valid'; ties -> first in row-major (di,dj).
    sel ({0,1} array of x's shape with exactly one 1 per 2 x 2 window): take THAT element of every window instead of the maximum -- the same
    function wherever it is the maximum; the GPU tests pass the engine's own choices (pool_selection) so that a window whose two largest
    entries differ by less than the fp32 round-off (the gradient is routed to the other element: a discontinuity, not an arithmetic error) does
    not enter the gradient comparison."""
    if sel is not None:
        n, h, w, c = x.shape
        return (x * sel)[:, :h // 2 * 2, :w // 2 * 2].reshape(n, h // 2, 2, w // 2, 2, c).sum(dim=(2, 4))
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)


def pool_selection(x):
    """{0,1} float64 array marking, per 2 x 2 window of x [N,H,W,C] (numpy), the element MaxPooling2D picks (first maximum in row-major order)."""
    x = np.asarray(x)
    n, h, w, c = x.shape
    h2, w2 = h // 2, w // 2
    win = x[:, :h2 * 2, :w2 * 2].reshape(n, h2, 2, w2, 2, c).transpose(0, 1, 3, 5, 2, 4).reshape(n, h2, w2, c, 4)
    one = np.eye(4)[win.argmax(-1)]                                      # first maximum on ties
    sel = np.zeros((n, h, w, c))
    sel[:, :h2 * 2, :w2 * 2] = one.reshape(n, h2, w2, c, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(n, h2 * 2, w2 * 2, c)
    return sel


def dropout(x, keep_mask, rate=DROPOUT_RATE):
    """Dropout(0.25) T1:863, inverted: y = x*mask/(1-rate).  keep_mask None -> identity."""
    if keep_mask is None:
        return x
    return x * keep_mask * (1.0 / (1.0 - rate))


def convT2x2s2_bias(x, kernel, bias):
    """Conv2DTranspose(C,(2,2),strides=(2,2),'same') T1:886:
    u[n,2i+a,2j+b,o] = bias[o] + sum_c x[n,i,j,c]*K[a,b,o,c]."""
    y = F.conv_transpose2d(x.permute(0, 3, 1, 2), kernel.permute(3, 2, 0, 1), bias, stride=2)
    return y.permute(0, 2, 3, 1)


class _StoreBF16(torch.autograd.Function):
    """Emulates the build's bf16-STORAGE mode (BASELINE.json configs[3]/[4] name bf16; the reference itself is fp32): a tensor that
    the engine writes to HBM as bf16 is rounded to bf16 (nearest even) on the way forward, and so is its gradient on the way
    back; all arithmetic in between stays in the oracle's precision.  Not part of the reference's algorithm: test
    infrastructure for the mixed-precision path only."""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def store_bf16(t):
    return _StoreBF16.apply(t)


def forward(weights, x, training=False, keep_masks=None, dtype=torch.float32, want_acts=False, store=None, ckpt=False, relu_masks=None, pool_sel=None):
    """Whole graph T1:853-916.  weights: dict name->array/tensor.  x: [N,H,W,Cin].
    keep_masks: None (dropout off / inference) or dict 'p1'..'p4' -> {0,1} arrays of the
    pooled shapes.  store: None, or store_bf16 to emulate bf16 storage of every activation the engine materialises
    (conv / BN / pool / ConvT outputs and the concat buffers; the probabilities stay full precision).
    relu_masks: None or dict conv name ('c1a' ... 'c9b') -> {0,1} array: the ReLU sign pattern to use (see conv3x3_bias_relu);
    pool_sel: None or dict 'p1'..'p4' -> {0,1} array: the element every 2 x 2 pooling window takes (see maxpool2x2).
    ckpt: recompute each block in backward (torch.utils.checkpoint) instead of keeping its intermediates: same
    arithmetic, a third of the memory -- what lets the fp64 golden of the 512x512 batch-16 step fit in this
    container (tests/golden/make_fullsize_goldens.py); no activations are returned then.
    Returns (p, acts, bn_batch_stats)."""
    st = store if store is not None else (lambda t: t)
    W = {k: _t(v, dtype) for k, v in weights.items()}
    a = OrderedDict()
    stats = OrderedDict()
    keep = want_acts and not ckpt
    rm = (lambda n: _t(relu_masks[n], dtype)) if relu_masks is not None else (lambda n: None)

    def run(fn, *args):
        if ckpt:
            from torch.utils.checkpoint import checkpoint
            return checkpoint(fn, *args, use_reentrant=False)
        return fn(*args)

    def enc(k, h):                                                           # T1:859-881
        h = st(conv3x3_bias_relu(h, W[f"c{k}a/kernel"], W[f"c{k}a/bias"], relu_mask=rm(f"c{k}a")))
        if keep: a[f"c{k}a"] = h
        h = st(conv3x3_bias_relu(h, W[f"c{k}b/kernel"], W[f"c{k}b/bias"], relu_mask=rm(f"c{k}b")))
        if keep: a[f"c{k}b"] = h
        h, mu, va = batchnorm(h, W[f"bn{k}/gamma"], W[f"bn{k}/beta"], W[f"bn{k}/mean"], W[f"bn{k}/var"], training)
        h = st(h); stats[f"bn{k}"] = (mu.detach(), va.detach(), h.shape[0] * h.shape[1] * h.shape[2])
        if keep: a[f"bn{k}"] = h
        skip = h
        h = st(maxpool2x2(h, _t(pool_sel[f"p{k}"], dtype) if pool_sel is not None else None))
        if training and keep_masks is not None:
            h = st(dropout(h, _t(keep_masks[f"p{k}"], dtype)))
        if keep: a[f"p{k}"] = h
        return skip, h

    def mid(h):                                                              # T1:883-884
        h = st(conv3x3_bias_relu(h, W["c5a/kernel"], W["c5a/bias"], relu_mask=rm("c5a")))
        if keep: a["c5a"] = h
        h = st(conv3x3_bias_relu(h, W["c5b/kernel"], W["c5b/bias"], relu_mask=rm("c5b")))
        if keep: a["c5b"] = h
        return h

    def dec(k, h, skip):                                                     # T1:886-911
        u = st(convT2x2s2_bias(h, W[f"u{k}/kernel"], W[f"u{k}/bias"]))
        if keep: a[f"u{k}"] = u
        h = st(torch.cat([u, skip], dim=3))                                  # [up, skip]
        h, mu, va = batchnorm(h, W[f"bn{k}/gamma"], W[f"bn{k}/beta"], W[f"bn{k}/mean"], W[f"bn{k}/var"], training)
        h = st(h); stats[f"bn{k}"] = (mu.detach(), va.detach(), h.shape[0] * h.shape[1] * h.shape[2])
        if keep: a[f"bn{k}"] = h
        h = st(conv3x3_bias_relu(h, W[f"c{k}a/kernel"], W[f"c{k}a/bias"], relu_mask=rm(f"c{k}a")))
        if keep: a[f"c{k}a"] = h
        h = st(conv3x3_bias_relu(h, W[f"c{k}b/kernel"], W[f"c{k}b/bias"], relu_mask=rm(f"c{k}b")))
        if keep: a[f"c{k}b"] = h
        return h

    h = _t(x, dtype)
    skips = {}
    for k in (1, 2, 3, 4):
        skips[k], h = run(lambda t, k=k: enc(k, t), h)
    h = run(mid, h)
    for k, sk in zip((6, 7, 8, 9), (4, 3, 2, 1)):
        h = run(lambda t, s, k=k: dec(k, t, s), h, skips[sk])
    p = conv1x1_sigmoid(h, W["out/kernel"], W["out/bias"])                   # T1:913
    if keep: a["out"] = p
    return (p, a, stats) if keep else (p, None, stats)


# ---------------------------------------------------------------------------------------
# Loss / metrics
# ---------------------------------------------------------------------------------------
def dice_coeff(y_true, y_pred):
    """T1:784-790: (2*sum(t*p)+1)/(sum(t)+sum(p)+1) over ALL elements of the batch."""
    inter = (y_true * y_pred).sum()
    return (2.0 * inter + 1.0) / (y_true.sum() + y_pred.sum() + 1.0)


def binary_crossentropy_mean(y_true, y_pred):
    """keras binary_crossentropy on probabilities, then Keras' mean over everything.
    clip -> logit -> max(z,0) - z*t + log(1+exp(-|z|))  (same algebra as T1:819-825).
    VERSION NOTE (parity unpinned, see the header): this is the form of standalone Keras 2.3 on TF 1.x (`K.binary_crossentropy` with
    from_logits=False clips to [eps, 1 - eps] with eps = 1e-7 and converts back to logits) and of the reference's own weighted_bce_loss.
    tf.keras / TF 2.2 -- the other stack the reference's notebooks could have run on -- back-tracks a `Sigmoid` op to its logits and applies
    sigmoid_cross_entropy_with_logits WITHOUT the clip.  The two differ by <= 1e-7 per element except where |logit| > 16.1 (p within 1e-7 of 0 or 1),
    where the clipped form saturates at 16.1 * |t - p|; far below the 1e-3 Dice bar and invisible to the tests, but a property of this restatement."""
    p = torch.clamp(y_pred, BCE_EPS, 1.0 - BCE_EPS)
    z = torch.log(p / (1.0 - p))
    l = torch.clamp(z, min=0) - z * y_true + torch.log1p(torch.exp(-torch.abs(z)))
    return l.mean()


def bce_dice_loss(y_true, y_pred):
    """T1:797-799: 0.5*BCE + 0.5*(1-dice)."""
    return 0.5 * binary_crossentropy_mean(y_true, y_pred) + 0.5 * (1.0 - dice_coeff(y_true, y_pred))


def threshold_sums(y_true, y_pred, thresholds):
    """segmentation_models building blocks: pr=(p>t); tp=sum(gt*pr); sum(pr); sum(gt) (gt soft)."""
    gt = np.asarray(y_true, np.float64).ravel()
    p = np.asarray(y_pred).ravel()
    out = np.zeros((len(thresholds), 3), np.float64)
    for i, t in enumerate(thresholds):
        pr = (p > np.float32(t)).astype(np.float64)
        out[i] = ((gt * pr).sum(), pr.sum(), gt.sum())
    return out


def sm_scores(tp, spr, sgt, smooth=SM_SMOOTH):
    """segmentation_models 1.0 FScore(beta=1) / IOUScore / Precision / Recall from the sums."""
    fp, fn = spr - tp, sgt - tp
    return {
        "dice": (2 * tp + smooth) / (2 * tp + fn + fp + smooth),
        "iou": (tp + smooth) / (sgt + spr - tp + smooth),
        "precision": (tp + smooth) / (tp + fp + smooth),
        "recall": (tp + smooth) / (tp + fn + smooth),
    }


# ---------------------------------------------------------------------------------------
# Training step (fwd -> loss -> autograd bwd -> Keras-form Adam) and evaluation
# ---------------------------------------------------------------------------------------
def loss_and_grads(weights, x, y, keep_masks=None, dtype=torch.float32, want_acts=False, store=None, ckpt=False, relu_masks=None, pool_sel=None):
    """One training-mode fwd + bwd.  Returns dict(loss, dice, grads{name}, bn_stats, p[, acts, act_grads])."""
    names = trainable_names(np.asarray(x).shape[-1])
    W = {k: _t(v, dtype).clone() for k, v in weights.items()}
    for k in names:
        W[k].requires_grad_(True)
    p, acts, stats = forward(W, x, training=True, keep_masks=keep_masks, dtype=dtype, want_acts=want_acts, store=store, ckpt=ckpt, relu_masks=relu_masks, pool_sel=pool_sel)
    t = _t(y, dtype)
    loss = bce_dice_loss(t, p)
    dice = dice_coeff(t, p)
    if want_acts:
        for v in acts.values():
            v.retain_grad()
    loss.backward()
    out = dict(loss=float(loss.detach()), dice=float(dice.detach()), p=p.detach().numpy(),
               grads={k: W[k].grad.numpy() for k in names},
               bn_stats={k: (m.detach().numpy(), v.detach().numpy(), n) for k, (m, v, n) in stats.items()})
    if want_acts:
        out["acts"] = {k: v.detach().numpy() for k, v in acts.items()}
        out["act_grads"] = {k: (v.grad.numpy() if v.grad is not None else None) for k, v in acts.items()}
    return out


def adam_keras(params, grads, m, v, t, lr=ADAM_LR, b1=ADAM_B1, b2=ADAM_B2, eps=ADAM_EPS):
    """Keras-2.3 Adam (T1:1053): t is the 1-based step AFTER increment.
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; p -= lr_t*m/(sqrt(v)+eps).  In place."""
    lr_t = lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
    for k in grads:
        g = grads[k].astype(params[k].dtype)
        m[k] = b1 * m[k] + (1 - b1) * g
        v[k] = b2 * v[k] + (1 - b2) * g * g
        params[k] = (params[k] - lr_t * m[k] / (np.sqrt(v[k]) + eps)).astype(params[k].dtype)


class OracleTrainer:
    """Stateful fwd/bwd/Adam on CPU: the reference's model.compile + train_on_batch."""

    def __init__(self, weights, dtype=torch.float32, arch="unet"):
        self.w = OrderedDict((k, np.array(v)) for k, v in weights.items())
        self.dtype = dtype
        self.arch = arch
        self.in_ch = self.w["c1a/kernel"].shape[2]
        names = trainable_names(self.in_ch) if arch == "unet" else pp_trainable_names(self.in_ch)
        self.m = {k: np.zeros_like(self.w[k]) for k in names}
        self.v = {k: np.zeros_like(self.w[k]) for k in names}
        self.t = 0
        self._fwd = forward if arch == "unet" else pp_forward

    def train_step(self, x, y, keep_masks=None):
        r = (loss_and_grads if self.arch == "unet" else pp_loss_and_grads)(self.w, x, y, keep_masks, self.dtype)
        for k, (mu, va, n) in r["bn_stats"].items():
            nm, nv = bn_moving_update(self.w[k + "/mean"], self.w[k + "/var"], mu, va, n)
            self.w[k + "/mean"] = nm.astype(self.w[k + "/mean"].dtype)
            self.w[k + "/var"] = nv.astype(self.w[k + "/var"].dtype)
        self.t += 1
        adam_keras(self.w, r["grads"], self.m, self.v, self.t)
        return r["loss"], r["dice"]

    def predict(self, x, batch_size=32):
        outs = []
        with torch.no_grad():
            for i in range(0, len(x), batch_size):
                outs.append(self._fwd(self.w, x[i:i + batch_size], training=False, dtype=self.dtype)[0].numpy())
        return np.concatenate(outs, 0)

    def evaluate(self, x, y, batch_size=32, thresholds=()):
        """model.evaluate (T1:1101): loss sample-weighted over batches; metrics = mean of the
        per-batch values (Keras stateful-metric Mean).  Also per-threshold sm metrics."""
        losses, dices, ns, per_t = [], [], [], []
        with torch.no_grad():
            for i in range(0, len(x), batch_size):
                xb, yb = x[i:i + batch_size], y[i:i + batch_size]
                p = self._fwd(self.w, xb, training=False, dtype=self.dtype)[0]
                t = _t(yb, self.dtype)
                losses.append(float(bce_dice_loss(t, p))); dices.append(float(dice_coeff(t, p))); ns.append(len(xb))
                if len(thresholds):
                    s = threshold_sums(yb, p.numpy(), thresholds)
                    per_t.append({k: v for k, v in sm_scores(s[:, 0], s[:, 1], s[:, 2]).items()})
        out = dict(loss=float(np.average(losses, weights=ns)), dice_coeff=float(np.mean(dices)))
        if len(thresholds):
            for k in ("dice", "iou", "precision", "recall"):
                out[k] = np.mean([b[k] for b in per_t], axis=0)
        return out


# =======================================================================================
# U-Net++ (nested skips), /root/reference/Scripts/task1_unet_plus_plus.py:858-950  (UPP)
#   encoder k=1..4 : Conv(C,elu) -> Dropout(.2) -> Conv(C,elu) -> BN -> [c_k] -> MaxPool      UPP:876-914
#   conv_block(x,C): Conv(C,elu) -> Dropout(.4) -> BN -> Conv(C,elu) -> Dropout(.4) -> BN     UPP:860-868
#   nested decoder : x1_2=[up(c2),c1] x2_2=[up(c3),c2] x1_3=[up(x2_2),c1,x1_2] x3_2=[up(c4),c3]
#                    x2_3=[up(x3_2),c2,x2_2] x1_4=[up(x2_3),c1,x1_2,x1_3]                      UPP:888-924
#   head           : Conv(1,1x1,sigmoid, he_normal) on conv1_4                                UPP:946-947
#   (the c5 / depth-5 path is commented out in the reference, UPP:926-944; p4 is computed but unused)
# Same PARITY STATUS as above: third-party Keras semantics restated, "parity unpinned".
# =======================================================================================
PP_ENC_DROP, PP_BLOCK_DROP = 0.2, 0.4
PP_NODES = [  # (node, width, ConvT input, concat sources after the up-sampled tensor)
    ("x1_2", 32, "c2", ["c1"]), ("x2_2", 64, "c3", ["c2"]), ("x1_3", 32, "x2_2", ["c1", "x1_2"]),
    ("x3_2", 128, "c4", ["c3"]), ("x2_3", 64, "x3_2", ["c2", "x2_2"]), ("x1_4", 32, "x2_3", ["c1", "x1_2", "x1_3"])]
PP_WIDTH = {"c1": 32, "c2": 64, "c3": 128, "c4": 256, "x1_2": 32, "x2_2": 64, "x1_3": 32, "x3_2": 128, "x2_3": 64, "x1_4": 32}


def pp_layer_table(in_ch: int = 1):
    """[(name, kind, cin, cout)] in Keras creation order (UPP:876-947)."""
    t = []

    def enc(k, cin, c):
        return [(f"c{k}a", "conv3", cin, c), (f"c{k}b", "conv3", c, c), (f"bn{k}", "bn", c, c)]

    def node(name, c, src, skips):
        cin = c + sum(PP_WIDTH[s] for s in skips)
        return [(f"u{name[1:]}", "convT", PP_WIDTH[src], c), (f"{name}a", "conv3", cin, c), (f"{name}abn", "bn", c, c),
                (f"{name}b", "conv3", c, c), (f"{name}bbn", "bn", c, c)]

    t += enc(1, in_ch, 32) + enc(2, 32, 64) + node(*PP_NODES[0]) + enc(3, 64, 128) + node(*PP_NODES[1]) + node(*PP_NODES[2])
    t += enc(4, 128, 256) + node(*PP_NODES[3]) + node(*PP_NODES[4]) + node(*PP_NODES[5])
    t += [("out", "conv1", 32, 1)]
    return t


def pp_weight_shapes(in_ch: int = 1):
    d = OrderedDict()
    for name, kind, cin, cout in pp_layer_table(in_ch):
        if kind == "conv3":
            d[name + "/kernel"] = (3, 3, cin, cout); d[name + "/bias"] = (cout,)
        elif kind == "conv1":
            d[name + "/kernel"] = (1, 1, cin, cout); d[name + "/bias"] = (cout,)
        elif kind == "convT":
            d[name + "/kernel"] = (2, 2, cout, cin); d[name + "/bias"] = (cout,)
        else:
            for p in ("gamma", "beta", "mean", "var"):
                d[f"{name}/{p}"] = (cout,)
    return d


def pp_trainable_names(in_ch: int = 1):
    return [k for k in pp_weight_shapes(in_ch) if not (k.endswith("/mean") or k.endswith("/var"))]


def pp_init_weights(seed: int = 0, in_ch: int = 1, dtype=np.float32):
    """he_normal for every Conv2D incl. the head (UPP:861-947), glorot_uniform for ConvT (Keras default)."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, kind, cin, cout in pp_layer_table(in_ch):
        if kind in ("conv3", "conv1"):
            kk = 3 if kind == "conv3" else 1
            std = math.sqrt(2.0 / (kk * kk * cin)) / 0.87962566103423978
            k = rng.standard_normal((kk, kk, cin, cout))
            bad = np.abs(k) > 2.0
            while bad.any():
                k[bad] = rng.standard_normal(int(bad.sum())); bad = np.abs(k) > 2.0
            w[name + "/kernel"] = (k * std).astype(dtype); w[name + "/bias"] = np.zeros(cout, dtype)
        elif kind == "convT":
            lim = math.sqrt(6.0 / (4 * cout + 4 * cin))
            w[name + "/kernel"] = rng.uniform(-lim, lim, (2, 2, cout, cin)).astype(dtype); w[name + "/bias"] = np.zeros(cout, dtype)
        else:
            w[name + "/gamma"] = np.ones(cout, dtype); w[name + "/beta"] = np.zeros(cout, dtype)
            w[name + "/mean"] = np.zeros(cout, dtype); w[name + "/var"] = np.ones(cout, dtype)
    return w


def pp_forward(weights, x, training=False, keep_masks=None, dtype=torch.float32, want_acts=False, ckpt=False):
    """U-Net++ graph.  keep_masks: None or dict conv-name -> {0,1} array of that conv's output shape (the
    Dropout that follows it); ckpt: recompute each encoder block / node in backward (see forward()); returns
    (p, acts, bn_batch_stats)."""
    W = {k: _t(v, dtype) for k, v in weights.items()}
    a, stats, T = OrderedDict(), OrderedDict(), {}
    keep = want_acts and not ckpt

    def run(fn, *args):
        if ckpt:
            from torch.utils.checkpoint import checkpoint
            return checkpoint(fn, *args, use_reentrant=False)
        return fn(*args)

    def conv(name, h, rate):
        z = conv3x3_bias_relu(h, W[name + "/kernel"], W[name + "/bias"], relu=False)
        y = F.elu(z)
        if training and keep_masks is not None and rate > 0:
            y = dropout(y, _t(keep_masks[name], dtype), rate)
        if keep: a[name] = y
        return y

    def bn(name, h):
        y, mu, va = batchnorm(h, W[name + "/gamma"], W[name + "/beta"], W[name + "/mean"], W[name + "/var"], training)
        stats[name] = (mu.detach(), va.detach(), h.shape[0] * h.shape[1] * h.shape[2])
        if keep: a[name] = y
        return y

    def enc(k, h):
        h = conv(f"c{k}a", h, PP_ENC_DROP)
        h = conv(f"c{k}b", h, 0.0)
        return bn(f"bn{k}", h)

    def node(nm, c, tsrc, *tskips):
        u = convT2x2s2_bias(tsrc, W[f"u{nm[1:]}/kernel"], W[f"u{nm[1:]}/bias"])
        if keep: a[f"u{nm[1:]}"] = u
        hh = torch.cat([u] + list(tskips), dim=3)
        hh = bn(nm + "abn", conv(nm + "a", hh, PP_BLOCK_DROP))
        return bn(nm + "bbn", conv(nm + "b", hh, PP_BLOCK_DROP))

    h = _t(x, dtype)
    todo = {1: [], 2: ["x1_2"], 3: ["x2_2", "x1_3"], 4: ["x3_2", "x2_3", "x1_4"]}
    nodes = {n[0]: n for n in PP_NODES}
    for k in (1, 2, 3, 4):
        T[f"c{k}"] = run(lambda t, k=k: enc(k, t), h)
        for nm in todo[k]:
            _, c, src, skips = nodes[nm]
            T[nm] = run(lambda ts, *sk, nm=nm, c=c: node(nm, c, ts, *sk), T[src], *[T[s] for s in skips])
        h = maxpool2x2(T[f"c{k}"])
        if keep: a[f"p{k}"] = h
    p = conv1x1_sigmoid(T["x1_4"], W["out/kernel"], W["out/bias"])
    if keep: a["out"] = p
    return (p, a, stats) if keep else (p, None, stats)


def pp_loss_and_grads(weights, x, y, keep_masks=None, dtype=torch.float32, want_acts=False, ckpt=False):
    names = pp_trainable_names(np.asarray(x).shape[-1])
    W = {k: _t(v, dtype).clone() for k, v in weights.items()}
    for k in names:
        W[k].requires_grad_(True)
    p, acts, stats = pp_forward(W, x, training=True, keep_masks=keep_masks, dtype=dtype, want_acts=want_acts, ckpt=ckpt)
    t = _t(y, dtype)
    loss = bce_dice_loss(t, p); dice = dice_coeff(t, p)
    if want_acts:
        for v in acts.values():
            v.retain_grad()
    loss.backward()
    out = dict(loss=float(loss.detach()), dice=float(dice.detach()), p=p.detach().numpy(),
               grads={k: W[k].grad.numpy() for k in names},
               bn_stats={k: (m.detach().numpy(), v.detach().numpy(), n) for k, (m, v, n) in stats.items()})
    if want_acts:
        out["acts"] = {k: v.detach().numpy() for k, v in acts.items()}
        out["act_grads"] = {k: (v.grad.numpy() if v.grad is not None else None) for k, v in acts.items()}
    return out


# =======================================================================================
# Slice classifier, /root/reference/Scripts/task2_covid19_classifcation.py:747-776  (T2)
#   block k (C=16,32,64): Conv(C,relu,he_normal) -> BN -> Conv(C,relu,he_normal) -> BN -> MaxPool      T2:748-764
#   Flatten -> Dense(32,relu) -> Dropout(.4) -> Dense(1,sigmoid)                                        T2:772-776
#   compile(loss='binary_crossentropy', Adam(5e-4), metrics=[f1])  T2:829; f1/precision/recall closures T2:688-703
#   fit(..., class_weight=weights) T2:833-835: Keras multiplies the per-sample loss by the weight of the sample's class
#   and averages over the batch (keras/engine/training_utils.py weighted_masked_objective).  NOTE: the reference passes the
#   ndarray returned by sklearn's compute_class_weight, which Keras 2.3 only honours when it is a dict -- the oracle exposes
#   the weights as an explicit argument, (1, 1) = what the reference effectively trains with.
# Same PARITY STATUS as above for the Keras layers: third-party semantics restated, "parity unpinned".  f1/precision/recall are the
# reference's own closures and ARE pinned: tests/golden/f1_goldens.npz holds the outputs of the reference's functions executed on
# seeded vectors (tests/golden/make_f1_goldens.py), tests/test_classifier_host.py checks cls_f1 against them.
# =======================================================================================
CLS_C = [16, 32, 64]
CLS_HIDDEN = 32
CLS_DROP = 0.4
K_EPS = 1e-7


def cls_layer_table(in_ch: int = 1, hw=(224, 224)):
    t, cp = [], in_ch
    for k, c in enumerate(CLS_C, 1):
        t += [(f"c{k}a", "conv3", cp, c), (f"bn{k}a", "bn", c, c), (f"c{k}b", "conv3", c, c), (f"bn{k}b", "bn", c, c)]
        cp = c
    t += [("fc1", "dense", (hw[0] // 8) * (hw[1] // 8) * CLS_C[-1], CLS_HIDDEN), ("fc2", "dense", CLS_HIDDEN, 1)]
    return t


def cls_weight_shapes(in_ch: int = 1, hw=(224, 224)):
    d = OrderedDict()
    for name, kind, cin, cout in cls_layer_table(in_ch, hw):
        if kind == "conv3":
            d[f"{name}/kernel"] = (3, 3, cin, cout); d[f"{name}/bias"] = (cout,)
        elif kind == "dense":
            d[f"{name}/kernel"] = (cin, cout); d[f"{name}/bias"] = (cout,)
        else:
            for p in ("gamma", "beta", "mean", "var"):
                d[f"{name}/{p}"] = (cout,)
    return d


def cls_trainable_names(in_ch: int = 1, hw=(224, 224)):
    return [k for k in cls_weight_shapes(in_ch, hw) if not (k.endswith("/mean") or k.endswith("/var"))]


def cls_init_weights(seed: int = 0, in_ch: int = 1, hw=(224, 224), dtype=np.float32):
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, shape in cls_weight_shapes(in_ch, hw).items():
        if name.endswith("/kernel") and len(shape) == 4:            # he_normal (T2:748...)
            std = math.sqrt(2.0 / (9 * shape[2])) / 0.87962566103423978
            k = rng.standard_normal(shape); bad = np.abs(k) > 2
            while bad.any():
                k[bad] = rng.standard_normal(int(bad.sum())); bad = np.abs(k) > 2
            w[name] = (k * std).astype(dtype)
        elif name.endswith("/kernel"):                              # Dense default glorot_uniform
            lim = math.sqrt(6.0 / (shape[0] + shape[1]))
            w[name] = rng.uniform(-lim, lim, shape).astype(dtype)
        elif name.endswith("/gamma") or name.endswith("/var"):
            w[name] = np.ones(shape, dtype)
        else:
            w[name] = np.zeros(shape, dtype)
    return w


def cls_forward(weights, x, training=False, keep_mask=None, dtype=torch.float32, want_acts=False, ckpt=False, relu_masks=None, pool_sel=None):
    """keep_mask: None or {0,1} array [n, 32] of the Dropout(0.4) after Dense(32).  ckpt: recompute each conv block in
    backward (see forward()).  Returns (p [n], acts, bn stats)."""
    W = {k: _t(v, dtype) for k, v in weights.items()}
    a, stats = OrderedDict(), OrderedDict()
    keep = want_acts and not ckpt

    def block(k, h):
        for ab in "ab":
            h = conv3x3_bias_relu(h, W[f"c{k}{ab}/kernel"], W[f"c{k}{ab}/bias"], relu_mask=(_t(relu_masks[f"c{k}{ab}"], dtype) if relu_masks is not None else None))
            if keep: a[f"c{k}{ab}"] = h
            nm = f"bn{k}{ab}"
            src = h
            h, mu, va = batchnorm(h, W[nm + "/gamma"], W[nm + "/beta"], W[nm + "/mean"], W[nm + "/var"], training)
            stats[nm] = (mu.detach(), va.detach(), src.shape[0] * src.shape[1] * src.shape[2])
            if keep: a[nm] = h
        h = maxpool2x2(h, _t(pool_sel[f"p{k}"], dtype) if pool_sel is not None else None)
        if keep: a[f"p{k}"] = h
        return h

    h = _t(x, dtype)
    for k in (1, 2, 3):
        if ckpt:
            from torch.utils.checkpoint import checkpoint
            h = checkpoint(lambda t, k=k: block(k, t), h, use_reentrant=False)
        else:
            h = block(k, h)
    flat = h.reshape(h.shape[0], -1)                                  # channels_last Flatten: (i*W + j)*C + c
    h1 = torch.relu(flat @ W["fc1/kernel"] + W["fc1/bias"])
    if training and keep_mask is not None:
        h1 = dropout(h1, _t(keep_mask, dtype), CLS_DROP)
    if keep: a["h1"] = h1
    p = torch.sigmoid(h1 @ W["fc2/kernel"] + W["fc2/bias"]).reshape(-1)
    if keep: a["out"] = p
    return (p, a, stats) if keep else (p, None, stats)


def cls_f1(y_true, y_pred):
    """f1 / precision / recall of T2:688-703 (K.round = round-half-even, K.epsilon() = 1e-7)."""
    tp = torch.sum(torch.round(torch.clamp(y_true * y_pred, 0, 1)))
    possible = torch.sum(torch.round(torch.clamp(y_true, 0, 1)))
    predicted = torch.sum(torch.round(torch.clamp(y_pred, 0, 1)))
    precision = tp / (predicted + K_EPS); recall = tp / (possible + K_EPS)
    return 2 * ((precision * recall) / (precision + recall + K_EPS))


def cls_loss(y_true, y_pred, class_weights=(1.0, 1.0)):
    p = torch.clamp(y_pred, BCE_EPS, 1.0 - BCE_EPS)
    z = torch.log(p / (1.0 - p))
    l = torch.clamp(z, min=0) - z * y_true + torch.log1p(torch.exp(-torch.abs(z)))
    w = torch.where(y_true >= 0.5, torch.as_tensor(class_weights[1], dtype=l.dtype), torch.as_tensor(class_weights[0], dtype=l.dtype))
    return (l * w).mean()


def cls_loss_and_grads(weights, x, y, keep_mask=None, class_weights=(1.0, 1.0), dtype=torch.float32, want_acts=False, ckpt=False, relu_masks=None, pool_sel=None):
    xs = np.asarray(x)
    names = cls_trainable_names(xs.shape[-1], xs.shape[1:3])
    W = {k: _t(v, dtype).clone() for k, v in weights.items()}
    for k in names:
        W[k].requires_grad_(True)
    p, acts, stats = cls_forward(W, x, training=True, keep_mask=keep_mask, dtype=dtype, want_acts=want_acts, ckpt=ckpt, relu_masks=relu_masks, pool_sel=pool_sel)
    t = _t(np.asarray(y, np.float64).reshape(-1), dtype)
    loss = cls_loss(t, p, class_weights); f1 = cls_f1(t, p)
    if want_acts:
        for v in acts.values():
            v.retain_grad()
    loss.backward()
    out = dict(loss=float(loss.detach()), f1=float(f1.detach()), p=p.detach().numpy(),
               grads={k: W[k].grad.numpy() for k in names},
               bn_stats={k: (m.detach().numpy(), v.detach().numpy(), n) for k, (m, v, n) in stats.items()})
    if want_acts:
        out["acts"] = {k: v.detach().numpy() for k, v in acts.items()}
        out["act_grads"] = {k: (v.grad.numpy() if v.grad is not None else None) for k, v in acts.items()}
    return out


class ClsOracleTrainer:
    """Stateful fwd/bwd/Adam of the classifier on CPU (model.compile + train_on_batch, T2:829-835)."""

    def __init__(self, weights, dtype=torch.float32, class_weights=(1.0, 1.0)):
        self.w = OrderedDict((k, np.array(v)) for k, v in weights.items())
        self.dtype, self.cw = dtype, class_weights
        names = [k for k in self.w if not (k.endswith("/mean") or k.endswith("/var"))]
        self.m = {k: np.zeros_like(self.w[k]) for k in names}
        self.v = {k: np.zeros_like(self.w[k]) for k in names}
        self.t = 0

    def train_step(self, x, y, keep_mask=None):
        r = cls_loss_and_grads(self.w, x, y, keep_mask, self.cw, self.dtype)
        for k, (mu, va, n) in r["bn_stats"].items():
            nm, nv = bn_moving_update(self.w[k + "/mean"], self.w[k + "/var"], mu, va, n)
            self.w[k + "/mean"] = nm.astype(self.w[k + "/mean"].dtype); self.w[k + "/var"] = nv.astype(self.w[k + "/var"].dtype)
        self.t += 1
        adam_keras(self.w, r["grads"], self.m, self.v, self.t)
        return r["loss"], r["f1"]

    def predict(self, x, batch_size=32):
        outs = []
        with torch.no_grad():
            for i in range(0, len(x), batch_size):
                outs.append(cls_forward(self.w, x[i:i + batch_size], training=False, dtype=self.dtype)[0].numpy())
        return np.concatenate(outs, 0)

    def evaluate(self, x, y, batch_size=32):
        """model.evaluate (T2:884): loss sample-weighted over batches (no class weights at evaluate), f1 = mean of batch values."""
        ls, fs, ns = [], [], []
        with torch.no_grad():
            for i in range(0, len(x), batch_size):
                p = cls_forward(self.w, x[i:i + batch_size], training=False, dtype=self.dtype)[0]
                t = _t(np.asarray(y[i:i + batch_size], np.float64).reshape(-1), self.dtype)
                ls.append(float(cls_loss(t, p))); fs.append(float(cls_f1(t, p))); ns.append(len(p))
        return dict(loss=float(np.average(ls, weights=ns)), f1=float(np.mean(fs)))
