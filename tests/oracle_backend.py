"""Test-only backend: plugs the CPU oracle into keras_like.UNetModel so the HOST logic
(fit / evaluate / checkpoints / runners) can be exercised without a GPU.  Never shipped."""
import numpy as np

from oracle import unet_oracle as O


class OracleBackend:
    def __init__(self, h, w, in_ch=1, dtype=None, arch="unet"):
        import torch
        self.h, self.w, self.in_ch, self.arch = h, w, in_ch, arch
        self.dtype = dtype or torch.float32
        self.lr = O.ADAM_LR
        self.tr = None

    def set_weights(self, w):
        if self.tr is None:
            self.tr = O.OracleTrainer(w, self.dtype, self.arch)
        else:
            for k in self.tr.w:
                self.tr.w[k] = np.array(w[k], dtype=self.tr.w[k].dtype)

    def get_weights(self):
        return {k: np.array(v) for k, v in self.tr.w.items()}

    def reset_optimizer(self):
        for k in self.tr.m:
            self.tr.m[k][...] = 0; self.tr.v[k][...] = 0
        self.tr.t = 0

    def get_optimizer_state(self):
        return {"step": int(self.tr.t), "lr": float(self.lr), "m": {k: np.array(v, np.float32) for k, v in self.tr.m.items()},
                "v": {k: np.array(v, np.float32) for k, v in self.tr.v.items()}}

    def set_optimizer_state(self, st):
        for k in self.tr.m:
            self.tr.m[k][...] = st["m"][k]; self.tr.v[k][...] = st["v"][k]
        self.tr.t = int(st["step"]); self.lr = float(st.get("lr", self.lr))

    def train_batch(self, x, y, training_dropout=True):
        assert abs(self.lr - O.ADAM_LR) < 1e-12
        return np.array(self.tr.train_step(x, y, None))

    def predict_batch(self, x, y=None):
        import torch
        with torch.no_grad():
            p, acts, _ = self.tr._fwd(self.tr.w, x, training=False, dtype=self.dtype, want_acts=True)
            self._last_acts = {k: v.numpy() for k, v in acts.items()}
            ld = None
            if y is not None:
                t = torch.as_tensor(np.asarray(y), dtype=self.dtype)
                ld = np.array([float(O.bce_dice_loss(t, p)), float(O.dice_coeff(t, p))])
        return p.numpy(), ld

    def threshold_sums(self, p, y, thresholds):
        return O.threshold_sums(y, p, thresholds)

    def tap(self, n, name):
        """activation `name` of the last predict_batch (inference mode)"""
        return self._last_acts[name]


class ClsOracleBackend:
    """Same, for classifier.ClassifierModel (arch "classifier")."""

    def __init__(self, h, w, in_ch=1, dtype=None):
        import torch
        self.h, self.w, self.in_ch, self.arch = h, w, in_ch, "classifier"
        self.dtype = dtype or torch.float32
        self.lr = O.ADAM_LR
        self.tr = None
        self.cw = (1.0, 1.0)

    def set_weights(self, w):
        if self.tr is None:
            self.tr = O.ClsOracleTrainer(w, self.dtype)
        else:
            for k in self.tr.w:
                self.tr.w[k] = np.array(w[k], dtype=self.tr.w[k].dtype)

    def get_weights(self):
        return {k: np.array(v) for k, v in self.tr.w.items()}

    def reset_optimizer(self):
        for k in self.tr.m:
            self.tr.m[k][...] = 0; self.tr.v[k][...] = 0
        self.tr.t = 0

    def set_class_weights(self, w0, w1):
        self.cw = (float(w0), float(w1))

    def train_batch(self, x, y, training_dropout=True):
        self.tr.cw = self.cw
        return np.array(self.tr.train_step(x, y, None))

    def predict_batch(self, x, y=None):
        import torch
        with torch.no_grad():
            p = O.cls_forward(self.tr.w, x, training=False, dtype=self.dtype)[0]
            ld = None
            if y is not None:
                t = torch.as_tensor(np.asarray(y, np.float64).reshape(-1), dtype=self.dtype)
                ld = np.array([float(O.cls_loss(t, p, self.cw)), float(O.cls_f1(t, p))])
        return p.numpy().reshape(-1, 1), ld
