"""HIP engine: owns the flat parameter / gradient / Adam / workspace buffers (torch-ROCm is
used ONLY as device allocator, stream owner and process-group bootstrap) and drives the op
programs of libunet_hip.so.  One process per GPU; gradients and the few batch-global
reductions (BatchNorm sums, Dice sums) go through torch.distributed (RCCL over xGMI).

Replaces the Keras model object the reference builds at
task1_preprocessing_plus_unet_with_comments.py:853-916 and trains at T1:1053-1061.
"""
from __future__ import annotations

import ctypes as C
import math
import sys
from collections import OrderedDict

import numpy as np

from . import _lib, dp
from .weights import weight_shapes

ADAM_LR, ADAM_B1, ADAM_B2, ADAM_EPS = 5e-4, 0.9, 0.999, 1e-7     # Adam(lr=0.0005), T1:1053


def _torch():
    import torch
    return torch


class HipUNet:
    """Backend used by keras_like.UNetModel.  All tensors NHWC fp32 on one MI355X."""

    def __init__(self, h: int, w: int, in_ch: int = 1, device: int | None = None, conv_algo: int = _lib.ALGO_AUTO,
                 process_group=None, sync_bn: bool = True, dropout_rate: float = 0.25, seed: int = 0, lr: float = ADAM_LR,
                 arch: str = "unet", dtype: str = "fp32", force_dp: bool = False, options: dict | None = None, small_allreduce: str = "device",
                 comm_timeout_ms: int = 60000, private_context: bool = False, grad_buckets: bool = True):
        torch = _torch()
        self.lib = _lib.load()
        # dtype "bf16": activations / activation gradients stored as bf16 in the workspace (BASELINE.json configs[3], [4]); image,
        # targets, probabilities, parameters, gradients and Adam state stay fp32
        self.dtype = dtype
        self._dtype_id = {"fp32": _lib.DTYPE_F32, "bf16": _lib.DTYPE_BF16}[dtype]
        if not torch.cuda.is_available():
            raise _lib.UNetHipError("HipUNet: no GPU visible to torch (torch.cuda.is_available() is False); "
                                    "the hot path has no CPU fallback")
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.dev = torch.device("cuda", self.device_index)
        # options: {"deterministic": 1, "bn_fold": 0, ...} (_lib.OPTIONS; include/unet_hip.h UNET_OPT_*) -> a private context carrying them
        # (engines with equal options share one context -- single thread, single stream; private_context=True: a context of this engine's own, _lib.Context)
        self.ctx = _lib.Context.get(self.device_index, options, private=private_context).retain()
        self.h, self.w, self.in_ch, self.algo = h, w, in_ch, conv_algo
        self.arch = arch                       # "unet" (T1:853-916) or "unetpp" (task1_unet_plus_plus.py:858-950; its dropout
        self._arch_id = {"unet": _lib.ARCH_UNET, "unetpp": _lib.ARCH_UNETPP, "classifier": _lib.ARCH_CLASSIFIER}[arch]   # rates fixed: >0 = on
        # (classifier: task2_covid19_classifcation.py:747-776; y / p are [n] vectors, the loss tensor is (bce, f1))
        self.class_weights = (1.0, 1.0)
        self.pg = process_group
        self.pg_grad = process_group
        self.world, self.rank = 1, 0
        # force_dp: walk the data-parallel program (sync points, side stream, second communicator) even at world size 1 -- every SUM all-reduce is
        # then the identity, so the step equals the plain one up to the order of the floating-point atomics (bit for bit under
        # options={"deterministic": 1}): how the RCCL code path is exercised on a 1-GPU box
        self._dp = False
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
            self.rank = dist.get_rank(process_group)
            self._dp = self.world > 1 or bool(force_dp)
            if self._dp:
                # gradient buckets get their OWN communicator: collectives of one process group are serialised on one
                # internal stream, so a bucket all-reduce would otherwise delay the small inline (BN / Dice) reductions
                self.pg_grad = dist.new_group(ranks=dist.get_process_group_ranks(process_group), backend=dist.get_backend(process_group))
        self.sync_bn = sync_bn
        self.dropout_rate, self.seed, self.lr = float(dropout_rate), int(seed), float(lr)
        self.step = 0
        self._drop_calls = 0                   # dropout stream position: advances with every training forward, NOT reset by compile() / reset_optimizer()
        self._plans = {}
        self._ws = None
        self._infer_ready = None               # the plan (by identity) whose inference-only preparation -- split weight images, moving-statistics scale / shift, folded tables -- is current
        self._pinned = {}                      # _to_dev: pinned staging rings by element count
        self._idx_pin, self._idx_ev, self._idx_i = [None] * 8, [None] * 8, 0
        self._loss_chunk, self._loss_i = None, 0
        # flat buffers sized from a probe plan
        probe = self._create_plan(1)
        self.n_params = self.lib.unet_model_param_count(probe)
        self.n_state = self.lib.unet_model_state_count(probe)
        self.params = torch.zeros(self.n_params, dtype=torch.float32, device=self.dev)
        self.grads = torch.zeros_like(self.params)
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        self.state = torch.zeros(self.n_state, dtype=torch.float32, device=self.dev)
        self._tinfo = OrderedDict()
        for name, shape in weight_shapes(in_ch, arch, (h, w)).items():
            st, off, cnt = C.c_int32(), C.c_int64(), C.c_int64()
            self.ctx.check(self.lib.unet_model_tensor_info(probe, name.encode(), C.byref(st), C.byref(off), C.byref(cnt)), "tensor_info")
            assert cnt.value == int(np.prod(shape)), (name, cnt.value, shape)
            self._tinfo[name] = (bool(st.value), off.value, cnt.value, shape)
        self.lib.unet_model_destroy(probe)
        self._comm_stream = torch.cuda.Stream(device=self.dev) if self._dp else None          # gradient buckets
        self._small_stream = torch.cuda.Stream(device=self.dev) if self._dp else None         # BatchNorm-backward sums whose reader is not the next op (dp.py)
        # small_allreduce: "device" = the BatchNorm / loss sums go through unet_comm_* (csrc/comm.hip: one kernel on the compute stream that pushes into the
        # peers' IPC-mapped receive areas; falls back to RCCL, loudly and on every rank together, if the areas cannot be mapped or a self-test fails);
        # "rccl" = torch.distributed all-reduces, inline or on the side stream.  Gradient buckets are RCCL either way.
        assert small_allreduce in ("device", "rccl"), small_allreduce
        # comm_timeout_ms: how long a device-side reduction polls for a peer before it latches the error word and returns NaN sums (keras_like.check_comm raises at
        # the next epoch / evaluate boundary).  A rank may legitimately lag (checkpoint writes, a new plan): raise it for slow filesystems, or take "rccl", which waits.
        self._comm_timeout_ms = int(comm_timeout_ms)
        self._comm_fallback = None             # why the device-side all-reduce was asked for but is not in use (None: in use, or never asked for)
        self._comm = self._make_comm() if (self._dp and small_allreduce == "device") else None
        # grad_buckets False (an A/B switch, bench.py --no-buckets): ONE all-reduce of the whole gradient buffer on the compute stream behind the backward program --
        # nothing overlaps it, so its whole duration is exposed (what the 5 overlapped buckets are measured against)
        self.grad_buckets = bool(grad_buckets)
        self._comm_prof = None                 # set_comm_profiling: event pairs of the gradient all-reduces and of the compute stream's wait in front of Adam

    # ------------------------------------------------------------------ plans / buffers
    def _create_plan(self, n, replicated=False):
        m = _lib.vp()
        self.ctx.check(self.lib.unet_model_create(self.ctx.handle, self._arch_id, self.in_ch, n, self.h, self.w,
                                                  self.world if (self.sync_bn and not replicated) else 1, self.algo, self._dtype_id, C.byref(m)), "model_create")
        return m

    def _plan(self, n: int, replicated: bool = False):
        """The op programs for a local batch of n.  replicated: a step that every rank runs on the SAME full batch with no cross-rank
        reduction (world size 1 baked in) -- how the host code handles a batch that does not divide by the world size."""
        torch = _torch()
        key = (n, True) if (replicated and self._dp) else n
        if key not in self._plans:
            m = self._create_plan(n, replicated)
            need = self.lib.unet_model_workspace_bytes(m, 1)
            self._plans[key] = {"m": m, "bytes": need, "bound_ws": None}
            if self.arch == "classifier":
                self.ctx.check(self.lib.unet_model_set_class_weights(m, *self.class_weights), "set_class_weights")
        p = self._plans[key]
        if self._ws is None or self._ws.numel() < p["bytes"]:
            self._ws = torch.empty(p["bytes"], dtype=torch.uint8, device=self.dev)
        if p["bound_ws"] != self._ws.data_ptr():
            self._infer_ready = None
            self.ctx.check(self.lib.unet_model_bind(p["m"], self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(),
                                                    self.adam_v.data_ptr(), self.state.data_ptr(), self._ws.data_ptr(),
                                                    self._ws.numel()), "model_bind")
            p["bound_ws"] = self._ws.data_ptr()
            p["sync"] = {}
            for prog in (0, 1, 2):
                cnt = self.lib.unet_model_sync_points(p["m"], prog, None, 0)
                arr = (_lib.SyncPoint * max(cnt, 1))()
                self.lib.unet_model_sync_points(p["m"], prog, arr, cnt)
                p["sync"][prog] = [(arr[i].after_op, arr[i].kind, arr[i].ptr, arr[i].count, arr[i].use_op) for i in range(cnt)]
        return p

    def _stream(self):
        return _torch().cuda.current_stream(self.dev).cuda_stream

    def _ws_view_f64(self, ptr, count):
        off = ptr - self._ws.data_ptr()
        return self._ws[off:off + 8 * count].view(_torch().float64)

    # ------------------------------------------------------------------ weights
    def set_weights(self, weights):
        torch = _torch()
        self._infer_ready = None
        for name, (is_state, off, cnt, shape) in self._tinfo.items():
            a = np.ascontiguousarray(np.asarray(weights[name], np.float32).reshape(-1))
            assert a.size == cnt, name
            (self.state if is_state else self.params)[off:off + cnt].copy_(torch.from_numpy(a))

    def get_weights(self):
        p, s = self.params.cpu().numpy(), self.state.cpu().numpy()
        return OrderedDict((name, (s if st else p)[off:off + cnt].reshape(shape).copy())
                           for name, (st, off, cnt, shape) in self._tinfo.items())

    def get_grads(self):
        g = self.grads.cpu().numpy()
        return OrderedDict((name, g[off:off + cnt].reshape(shape).copy())
                           for name, (st, off, cnt, shape) in self._tinfo.items() if not st)

    def set_class_weights(self, w0: float, w1: float):
        """Keras class_weight={0: w0, 1: w1} of model.fit (task2_covid19_classifcation.py:835); classifier only."""
        assert self.arch == "classifier"
        self.class_weights = (float(w0), float(w1))
        for p in self._plans.values():
            self.ctx.check(self.lib.unet_model_set_class_weights(p["m"], *self.class_weights), "set_class_weights")

    def _out_elems(self, n):
        return n if self.arch == "classifier" else n * self.h * self.w

    def reset_optimizer(self):
        self.adam_m.zero_(); self.adam_v.zero_(); self.step = 0

    def get_optimizer_state(self):
        """Adam's iteration count and moment slots keyed by tensor name (what Keras keeps in optimizer.weights, saved by model.save T1:1046-1047)"""
        m, v = self.adam_m.cpu().numpy(), self.adam_v.cpu().numpy()
        sl = [(name, off, cnt, shape) for name, (st, off, cnt, shape) in self._tinfo.items() if not st]
        return {"step": int(self.step), "lr": float(self.lr), "m": OrderedDict((n, m[o:o + c].reshape(s).copy()) for n, o, c, s in sl),
                "v": OrderedDict((n, v[o:o + c].reshape(s).copy()) for n, o, c, s in sl)}

    def set_optimizer_state(self, state):
        torch = _torch()
        for name, (st, off, cnt, shape) in self._tinfo.items():
            if st:
                continue
            for buf, src in ((self.adam_m, state["m"]), (self.adam_v, state["v"])):
                a = np.ascontiguousarray(np.asarray(src[name], np.float32).reshape(-1))
                assert a.size == cnt, name
                buf[off:off + cnt].copy_(torch.from_numpy(a))
        self.step = int(state["step"])
        if "lr" in state:
            self.lr = float(state["lr"])

    # ------------------------------------------------------------------ running programs
    def _to_dev(self, a):
        """Host array -> fp32 device tensor.  The copy goes through one of two PINNED staging buffers per size (the cast float64 -> float32 writes
        straight into it) and is asynchronous on the current stream: the host runs ahead to the next batch while this one computes, and the buffer is
        reused only after the copy that read it has finished (an event per buffer)."""
        torch = _torch()
        if isinstance(a, torch.Tensor):
            return a.to(self.dev, torch.float32).contiguous()
        a = np.asarray(a)
        ring = self._pinned.setdefault(a.size, {"buf": [], "ev": [], "i": 0})
        if len(ring["buf"]) < 2:
            ring["buf"].append(torch.empty(a.size, dtype=torch.float32).pin_memory()); ring["ev"].append(None)
        k = ring["i"] % len(ring["buf"]); ring["i"] += 1
        if ring["ev"][k] is not None:
            ring["ev"][k].synchronize()
        np.copyto(ring["buf"][k].numpy().reshape(a.shape), a, casting="same_kind")
        out = ring["buf"][k].to(self.dev, non_blocking=True).view(a.shape)
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(self.dev)); ring["ev"][k] = ev
        return out

    # ------------------------------------------------------------------ datasets resident in HBM (model.fit's x_train / y_train, T1:1059-1061)
    def resident(self, a, max_fraction: float = 0.4):
        """Upload a whole dataset [N, ...] ONCE as fp32 (chunks through the pinned ring) and return the device tensor -- or None when 4 * a.size
        exceeds max_fraction of the HBM that is free right now (the caller then feeds host batches through _to_dev).  The reference's sets are
        1130 x 224 x 224 (227 MB); 288 GB of HBM hold any CT set this path is run on."""
        torch = _torch()
        if isinstance(a, torch.Tensor):
            return a.to(self.dev, torch.float32).contiguous()
        a = np.asarray(a)
        free, _ = torch.cuda.mem_get_info(self.dev)
        if a.size == 0 or 4 * a.size > max_fraction * free:
            return None
        out = torch.empty(a.shape, dtype=torch.float32, device=self.dev)
        flat = out.view(a.shape[0], -1)
        per = max(1, (64 << 20) // max(4 * flat.shape[1], 1))                    # 64-MB chunks
        for i in range(0, a.shape[0], per):
            flat[i:i + per].copy_(self._to_dev(a[i:i + per]).view(-1, flat.shape[1]), non_blocking=True)
        return out

    def take(self, ds, idx):
        """ds[idx] on the device (unet_gather_samples): the mini-batch of a shuffled epoch out of a resident dataset; idx = host integers."""
        torch = _torch()
        idx = np.ascontiguousarray(idx, np.int64)
        if len(idx) and np.all(np.diff(idx) == 1):
            return ds[int(idx[0]):int(idx[0]) + len(idx)]                            # a contiguous range is a view
        sf = int(np.prod(ds.shape[1:]))
        if sf % 4:
            return ds[torch.from_numpy(idx).to(self.dev)]
        if len(idx) == 0:
            return torch.empty((0,) + tuple(ds.shape[1:]), dtype=torch.float32, device=self.dev)
        # eight pinned index buffers in rotation; a buffer is rewritten only after the H2D copy that read it has finished (an event per buffer, as in
        # _to_dev): fit() syncs the host once per epoch, so the host can queue many batches ahead of a GPU-bound step
        k = self._idx_i % 8; self._idx_i += 1
        if self._idx_ev[k] is not None:
            self._idx_ev[k].synchronize()
        if self._idx_pin[k] is None or self._idx_pin[k].numel() < len(idx):
            self._idx_pin[k] = torch.empty(max(len(idx), 256), dtype=torch.int64).pin_memory()
        self._idx_pin[k][:len(idx)].copy_(torch.from_numpy(idx))
        di = self._idx_pin[k][:len(idx)].to(self.dev, non_blocking=True)
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream(self.dev)); self._idx_ev[k] = ev
        out = torch.empty((len(idx),) + tuple(ds.shape[1:]), dtype=torch.float32, device=self.dev)
        self.ctx.check(self.lib.unet_gather_samples(self.ctx.handle, ds.data_ptr(), di.data_ptr(), out.data_ptr(), len(idx), sf, self._stream()), "gather_samples")
        return out

    def _make_comm(self):
        """unet_comm_* communicator over the ranks of self.pg (one node, <= 8 ranks), verified by one all-reduce of known values; None (= RCCL) unless
        EVERY rank got through."""
        import socket
        import warnings
        import torch.distributed as dist
        torch = _torch()
        lib, world, rank = self.lib, self.world, self.rank
        comm, why = _lib.vp(), ""
        handle = (C.c_ubyte * _lib.COMM_HANDLE_BYTES)()
        if world > _lib.COMM_MAX_WORLD:
            why = f"{world} ranks (the device-side all-reduce serves one node: <= {_lib.COMM_MAX_WORLD})"
        elif lib.unet_comm_create(self.ctx.handle, rank, world, C.byref(comm), handle) != 0:
            why, comm = self.ctx.last_error(), _lib.vp()
        mine = (socket.gethostname(), bytes(handle), why)
        everyone = [None] * world
        if world > 1:
            dist.all_gather_object(everyone, mine, group=self.pg)
        else:
            everyone = [mine]
        ok = not any(e[2] for e in everyone) and len({e[0] for e in everyone}) == 1
        if not ok and not why:
            why = next((f"rank {r}: {e[2]}" for r, e in enumerate(everyone) if e[2]), "ranks on different hosts")
        if ok and lib.unet_comm_connect(comm, b"".join(e[1] for e in everyone)) != 0:
            ok, why = False, self.ctx.last_error()
        if ok:                                             # self-test: sum over ranks of (rank + 1) * (i + 1), short timeout
            lib.unet_comm_set_timeout_ms(comm, 10000)
            t = (torch.arange(1, 301, dtype=torch.float64, device=self.dev) * (rank + 1))
            err = C.c_int32(0)
            if lib.unet_comm_allreduce_f64(comm, t.data_ptr(), t.numel(), self._stream()) != 0 or lib.unet_comm_status(comm, C.byref(err), self._stream()) != 0:
                ok, why = False, self.ctx.last_error()
            elif err.value != 0:
                ok, why = False, f"self-test: nothing from rank {err.value - 1} within 10 s"
            elif not torch.equal(t.cpu(), torch.arange(1, 301, dtype=torch.float64) * (world * (world + 1) // 2)):
                ok, why = False, "self-test: wrong sums"
            lib.unet_comm_set_timeout_ms(comm, self._comm_timeout_ms)
        if world > 1:                                      # all ranks or none
            flags = [None] * world
            dist.all_gather_object(flags, (ok, why), group=self.pg)
            bad = [(r, f[1]) for r, f in enumerate(flags) if not f[0]]
            if bad:
                ok, why = False, why or f"rank {bad[0][0]}: {bad[0][1]}"
        if not ok:
            if comm:
                lib.unet_comm_destroy(comm)
            warnings.warn(f"covidseg_amd: device-side small all-reduce unavailable ({why}); BatchNorm / loss sums go through torch.distributed instead", RuntimeWarning)
            self._comm_fallback = why or "unavailable"
            return None
        return comm

    def comm_selftest(self, calls=1000, doubles=1024):
        """First-contact diagnosis of the exchange paths of a data-parallel engine, without a training step (bench.py --comm-selftest): `calls` back-to-back small all-reduces
        of `doubles` float64 through the device-side communicator (csrc/comm.hip; exactness checked on every rank) and through torch.distributed, ten all-reduces of the whole
        gradient buffer through the bucket communicator -- microseconds per call by events on the stream each runs on."""
        import torch.distributed as dist
        torch = _torch()
        out = {"world": self.world, "rank": self.rank, "small_allreduce": "device (comm.hip)" if self._comm is not None else "torch.distributed", "fallback": self._comm_fallback}
        if not self._dp:
            return out
        ev = lambda: torch.cuda.Event(enable_timing=True)
        t = torch.ones(doubles, dtype=torch.float64, device=self.dev) * (self.rank + 1)
        if self._comm is not None:
            e0, e1 = ev(), ev()
            ok = True
            e0.record()
            for _ in range(calls):
                t.fill_(self.rank + 1.0)
                ok = ok and self.lib.unet_comm_allreduce_f64(self._comm, t.data_ptr(), t.numel(), self._stream()) == 0
            e1.record(); torch.cuda.synchronize(self.dev)
            out["device_us_per_call"] = round(e0.elapsed_time(e1) * 1e3 / calls, 2)          # (incl. the fill kernel in front of each call)
            out["device_exact"] = bool(ok and torch.all(t == self.world * (self.world + 1) / 2).item())
            out["device_status"] = self.comm_status()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(calls):
            t.fill_(self.rank + 1.0)
            dist.all_reduce(t, group=self.pg)
        e1.record(); torch.cuda.synchronize(self.dev)
        out["torch_distributed_us_per_call"] = round(e0.elapsed_time(e1) * 1e3 / calls, 2)
        out["torch_distributed_exact"] = bool(torch.all(t == self.world * (self.world + 1) / 2).item())
        g = torch.ones_like(self.grads)
        dist.all_reduce(g, group=self.pg_grad if getattr(self, "pg_grad", None) is not None else self.pg)
        torch.cuda.synchronize(self.dev)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(10):
            dist.all_reduce(g, group=self.pg_grad if getattr(self, "pg_grad", None) is not None else self.pg)
        e1.record(); torch.cuda.synchronize(self.dev)
        ms = e0.elapsed_time(e1) / 10
        out["gradient_buffer_mb"] = round(g.numel() * 4 / 1e6, 2); out["gradient_allreduce_ms"] = round(ms, 3)
        out["gradient_allreduce_busbw_gbs"] = round(2 * (self.world - 1) / self.world * g.numel() * 4 / (ms * 1e-3) / 1e9, 1) if self.world > 1 and ms > 0 else None
        return out

    def comm_status(self):
        """0, or 1 + the rank whose contribution to a device-side all-reduce did not arrive in time (sticky; synchronises the stream)"""
        if self._comm is None:
            return 0
        err = C.c_int32(0)
        self.ctx.check(self.lib.unet_comm_status(self._comm, C.byref(err), self._stream()), "comm_status")
        return err.value

    def set_comm_profiling(self, on: bool):
        """Data-parallel engines: bracket every gradient all-reduce (events on the side stream) and the compute stream's wait in front of Adam with timing events.
        comm_profile() turns what has been collected into per-step figures.  Off by default (a few event records per step)."""
        self._comm_prof = {"buckets": [], "waits": [], "small": 0, "steps0": self.step} if (on and self._dp) else None

    def comm_profile(self):
        """{"steps", "buckets": [{"mb", "ms"} in launch order], "allreduce_ms", "exposed_ms", "hidden_ms", "small_reductions_per_step"} averaged over the steps since
        set_comm_profiling(True): allreduce_ms = summed duration of a step's gradient all-reduces on their stream; exposed_ms = what the compute stream waited for them in
        front of the optimizer; hidden_ms = the rest (overlapped with backward).  Synchronises the device."""
        torch = _torch()
        pr = self._comm_prof
        if pr is None:
            return None
        torch.cuda.synchronize(self.dev)
        steps = max(1, self.step - pr["steps0"])
        per = max(1, len(pr["buckets"]) // steps)              # buckets per step, in launch order
        ms = [0.0] * per; mb = [0.0] * per
        for i, (nbytes, e0, e1) in enumerate(pr["buckets"][:per * steps]):
            ms[i % per] += e0.elapsed_time(e1) / steps; mb[i % per] = nbytes / 1e6
        exposed = sum(w0.elapsed_time(w1) for w0, w1 in pr["waits"]) / steps
        total = sum(ms)
        return {"steps": steps, "buckets": [{"mb": round(b, 2), "ms": round(m, 4)} for b, m in zip(mb, ms)], "allreduce_ms": round(total, 4), "exposed_ms": round(exposed, 4),
                "hidden_ms": round(max(total - exposed, 0.0), 4), "small_reductions_per_step": round(pr["small"] / steps, 1)}

    def close(self):
        if getattr(self, "_comm", None):
            self.lib.unet_comm_destroy(self._comm); self._comm = None
        for p in getattr(self, "_plans", {}).values():
            self.lib.unet_model_destroy(p["m"])
        self._plans = {}
        if getattr(self, "ctx", None) is not None:
            self.ctx.release(); self.ctx = None

    def __del__(self):
        if sys is None or sys.is_finalizing():                            # (interpreter teardown: the HIP runtime may be gone already -- the process's memory goes with it)
            return
        try:
            self.close()
        except Exception:
            pass

    def _all_reduce(self, t, group=None):
        """SUM all-reduce.  backend nccl (= RCCL) reduces device tensors in place; the gloo branch (used by the
        single-GPU multi-process tests) stages through the host."""
        import torch.distributed as dist
        group = group if group is not None else self.pg
        if dist.get_backend(group) == "gloo" and t.is_cuda:
            c = t.cpu(); dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group); t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    def _run(self, plan, prog, replicated=False):
        lib, m = self.lib, plan["m"]
        nops = lib.unet_model_num_ops(m, prog)
        run_range = lambda b, e: self.ctx.check(lib.unet_model_run(m, prog, b, e, self._stream()), "model_run")
        if not self._dp or replicated:
            # serving: a second predict on unchanged weights skips the ops that only re-derive per-weight data (weight images, inference BatchNorm scale / shift, folded
            # tables: ~20 launches, 0.15 of the 1.0 ms a batch-1 512 x 512 predict takes).  Anything that could have touched that data clears _infer_ready: set_weights,
            # an optimizer step, any other program or plan (all plans share one workspace), a workspace re-bind.
            if prog == _lib.PROG_FWD_INFER:
                if self._infer_ready is plan:
                    b = 0
                    for i in plan["infer_prep_ops"]:
                        if i > b:
                            run_range(b, i)
                        b = i + 1
                    if nops > b:
                        run_range(b, nops)
                    return
                run_range(0, nops)
                if "infer_prep_ops" not in plan:
                    names = [o[0] for o in self._op_names(m, prog)]
                    plan["infer_prep_ops"] = [i for i, nm in enumerate(names) if nm.startswith(("weight_images", "bn_finalize_infer", "bn_fold_prepare"))]
                self._infer_ready = plan
                return
            self._infer_ready = None
            run_range(0, nops)
            return
        self._infer_ready = None
        torch = _torch()
        cur = torch.cuda.current_stream(self.dev)

        def reduce_small(ptr, count):
            if self._comm is not None:
                self.ctx.check(lib.unet_comm_allreduce_f64(self._comm, ptr, count, self._stream()), "comm_allreduce")
            else:
                self._all_reduce(self._ws_view_f64(ptr, count))

        prof = self._comm_prof

        def reduce_bucket(ptr, count):
            off = (ptr - self.grads.data_ptr()) // 4
            ev = torch.cuda.Event(); ev.record(cur)
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                if prof is not None:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(self._comm_stream)
                self._all_reduce(self.grads[off:off + count], self.pg_grad)
                if prof is not None:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(self._comm_stream)
                    prof["buckets"].append((count * 4, e0, e1))

        def finish_buckets():
            # the compute stream meets the side stream in front of Adam: the time it stands here is the EXPOSED part of the gradient exchange
            if prof is not None:
                w0 = torch.cuda.Event(enable_timing=True); w0.record(cur)
            cur.wait_stream(self._comm_stream)
            if prof is not None:
                w1 = torch.cuda.Event(enable_timing=True); w1.record(cur)
                prof["waits"].append((w0, w1))

        def reduce_small_async(ptr, count):
            # the reduction runs on its own stream behind everything launched so far; the compute stream goes on with the independent op(s) the program
            # placed behind the producer and waits for `done` right before the reader (dp.run_program)
            ev = torch.cuda.Event(); ev.record(cur)
            with torch.cuda.stream(self._small_stream):
                self._small_stream.wait_event(ev)
                self._all_reduce(self._ws_view_f64(ptr, count))
                done = torch.cuda.Event(); done.record(self._small_stream)
            return done

        kinds = ((0, 1, 2) if self.sync_bn else ()) + ((3,) if self.grad_buckets else ())
        if prof is not None:
            prof["small"] += sum(1 for sp in plan["sync"][prog] if sp[1] in kinds and sp[1] != 3)
        # (device-side reductions cost one small kernel: they run inline even where the program leaves room for a side stream)
        dp.run_program(run_range, nops, plan["sync"][prog], reduce_small, reduce_bucket,
                       finish_buckets, kinds, None if self._comm is not None else reduce_small_async, lambda done: cur.wait_event(done))
        if prog == _lib.PROG_BWD and not self.grad_buckets:
            if prof is not None:
                e0 = torch.cuda.Event(enable_timing=True); e0.record(cur)
            self._all_reduce(self.grads, self.pg_grad)
            if prof is not None:
                e1 = torch.cuda.Event(enable_timing=True); e1.record(cur)
                prof["buckets"].append((self.grads.numel() * 4, e0, e1)); prof["waits"].append((e0, e1))

    def _loss_tensor(self, plan):
        torch = _torch()
        ptr = self.lib.unet_model_loss_ptr(plan["m"])
        off = ptr - self._ws.data_ptr()
        return self._ws[off:off + 8].view(torch.float32)

    def _loss_slot(self, plan):
        """A fresh [loss, metric] slot that the coming forward's loss_finalize writes besides the workspace pair (unet_model_set_loss_out): the per-step result a
        caller keeps (fit() reads one pair per step at the end of the epoch) with no copy kernel behind the step.  Slots are cut from 4096-pair chunks; a chunk
        lives as long as a slot of it does."""
        torch = _torch()
        if self._loss_chunk is None or self._loss_i >= self._loss_chunk.shape[0]:
            self._loss_chunk, self._loss_i = torch.zeros((4096, 2), dtype=torch.float32, device=self.dev), 0
        slot = self._loss_chunk[self._loss_i]; self._loss_i += 1
        self.ctx.check(self.lib.unet_model_set_loss_out(plan["m"], slot.data_ptr()), "set_loss_out")
        return slot

    def forward_backward(self, x, y, training_dropout=True, replicated=False):
        """fwd (training mode) + loss + bwd on one batch; gradients land in self.grads.
        Returns a device tensor [loss, dice_coeff] (no host sync).  Under data parallelism x, y are THIS rank's shard of the global batch
        (equal shard sizes on all ranks); replicated=True: x, y are the whole batch, identical on every rank, no reductions."""
        torch = _torch()
        xd, yd = self._to_dev(x), self._to_dev(y)
        n = xd.shape[0]
        plan = self._plan(n, replicated)
        if not hasattr(self, "_p_train") or self._p_train.numel() != self._out_elems(n):
            self._p_train = torch.empty(self._out_elems(n), dtype=torch.float32, device=self.dev)
        assert yd.numel() == self._out_elems(n), (tuple(yd.shape), self._out_elems(n))
        rate = self.dropout_rate if training_dropout else 0.0
        # one Philox key per (seed, rank, training forward): ranks draw different masks for their shards, and a re-compiled model (k-fold runner:
        # one compile() per fold, CV4:1062) continues the stream instead of replaying it
        self.ctx.check(self.lib.unet_model_set_dropout(plan["m"], rate, (self.seed * 1000003 + (0 if replicated else self.rank) * 2654435761 + self._drop_calls) & 0xFFFFFFFFFFFFFFFF), "set_dropout")
        self._drop_calls += 1
        self.ctx.check(self.lib.unet_model_set_io(plan["m"], xd.data_ptr(), yd.data_ptr(), self._p_train.data_ptr()), "set_io")
        self._keep = (xd, yd)
        slot = self._loss_slot(plan)
        self._run(plan, _lib.PROG_FWD_TRAIN, replicated)
        self._run(plan, _lib.PROG_BWD, replicated)
        return slot

    def adam_step(self, replicated=False):
        self.step += 1
        self._infer_ready = None
        t = self.step
        lr_t = self.lr * math.sqrt(1.0 - ADAM_B2 ** t) / (1.0 - ADAM_B1 ** t)
        self.ctx.check(self.lib.unet_adam_keras(self.ctx.handle, self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(),
                                                self.adam_v.data_ptr(), self.n_params, lr_t, ADAM_B1, ADAM_B2, ADAM_EPS,
                                                1.0 if (self.world == 1 or self.sync_bn or replicated) else 1.0 / self.world, self._stream()), "adam")

    def train_batch(self, x, y, training_dropout=True, replicated=False):
        """One optimizer step (model.fit inner loop, T1:1059).  Returns device tensor [loss, dice]."""
        out = self.forward_backward(x, y, training_dropout, replicated)          # (its own slot, written by the forward's loss_finalize: no copy behind the step)
        self.adam_step(replicated)
        return out

    def predict_batch(self, x, y=None, replicated=False):
        """Inference forward (moving BN stats, no dropout).  Returns (p [n,h,w,1] device tensor,
        [loss, dice] device tensor or None).  Under data parallelism (and not `replicated`) x, y are this rank's shard: p is the shard's,
        [loss, dice] the global batch's."""
        torch = _torch()
        xd = self._to_dev(x)
        n = xd.shape[0]
        plan = self._plan(n, replicated)
        p = torch.empty((n, 1) if self.arch == "classifier" else (n, self.h, self.w, 1), dtype=torch.float32, device=self.dev)
        yd = self._to_dev(y) if y is not None else None
        self.ctx.check(self.lib.unet_model_set_io(plan["m"], xd.data_ptr(), yd.data_ptr() if yd is not None else None, p.data_ptr()), "set_io")
        self._keep = (xd, yd)
        slot = self._loss_slot(plan) if yd is not None else None
        if yd is None:
            self.ctx.check(self.lib.unet_model_set_loss_out(plan["m"], None), "set_loss_out")
        self._run(plan, _lib.PROG_FWD_INFER, replicated)
        return p, slot

    def threshold_sums(self, p, y, thresholds, replicated=False):
        """[T,3] float64 (sum gt*pr, sum pr, sum gt) with pr = p > t  (sm.metrics, T1:1206-1207); summed over the ranks' shards under data parallelism."""
        torch = _torch()
        yd = self._to_dev(y)
        th = torch.tensor(np.asarray(thresholds, np.float32), device=self.dev)
        out = torch.zeros((len(thresholds), 3), dtype=torch.float64, device=self.dev)
        self.ctx.check(self.lib.unet_seg_metrics_sweep(self.ctx.handle, p.data_ptr(), yd.data_ptr(), th.data_ptr(), len(thresholds),
                                                       out.data_ptr(), p.numel(), self._stream()), "metrics_sweep")
        if self._dp and not replicated:
            self._all_reduce(out)
        return out

    def barrier(self):
        if self._dp:
            import torch.distributed as dist
            _torch().cuda.synchronize(self.dev)
            dist.barrier(group=self.pg)

    def tap(self, n, name, grad=False, replicated=False):
        """Copy of an intermediate activation / gradient of the last run (tests)."""
        return self.tap_device(n, name, grad, replicated).float().cpu().numpy()

    def tap_device(self, n, name, grad=False, replicated=False):
        """The same tensor as a strided VIEW of the workspace on the device [n, h, w, c] (valid until the next run overwrites it)."""
        torch = _torch()
        plan = self._plan(n, replicated)
        ptr, ld, nn, hh, ww, cc = _lib.vp(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self.ctx.check(self.lib.unet_model_tap(plan["m"], name.encode(), int(grad), C.byref(ptr), C.byref(ld), C.byref(nn), C.byref(hh),
                                               C.byref(ww), C.byref(cc)), f"tap({name})")
        off = ptr.value - self._ws.data_ptr()
        pix = nn.value * hh.value * ww.value
        esz = self.lib.unet_model_tap_elem_bytes(plan["m"], name.encode(), int(grad))
        tdt = torch.bfloat16 if esz == 2 else torch.float32
        flat = self._ws[off:off + esz * ((pix - 1) * ld.value + cc.value)].view(tdt)
        return torch.as_strided(flat, (nn.value, hh.value, ww.value, cc.value), (hh.value * ww.value * ld.value, ww.value * ld.value, ld.value, 1))

    def _op_names(self, m, prog):
        out = []
        for i in range(self.lib.unet_model_num_ops(m, prog)):
            nm, fl, by, ms, calls = C.c_char_p(), C.c_double(), C.c_double(), C.c_double(), C.c_int64()
            self.lib.unet_model_op_info(m, prog, i, C.byref(nm), C.byref(fl), C.byref(by), C.byref(ms), C.byref(calls))
            out.append((nm.value.decode(),))
        return out

    def op_profile(self, n, prog):
        """[(name, flops, bytes, ms, calls)] accumulated while ctx profiling was on."""
        plan = self._plan(n)
        out = []
        for i in range(self.lib.unet_model_num_ops(plan["m"], prog)):
            nm, fl, by, ms, calls = C.c_char_p(), C.c_double(), C.c_double(), C.c_double(), C.c_int64()
            self.lib.unet_model_op_info(plan["m"], prog, i, C.byref(nm), C.byref(fl), C.byref(by), C.byref(ms), C.byref(calls))
            out.append((nm.value.decode(), fl.value, by.value, ms.value, calls.value))
        return out

    def set_profiling(self, on: bool, n: int | None = None):
        self.ctx.check(self.lib.unet_ctx_set_profiling(self.ctx.handle, int(on)), "set_profiling")
        if n is not None:
            self.lib.unet_model_reset_timers(self._plan(n)["m"])
