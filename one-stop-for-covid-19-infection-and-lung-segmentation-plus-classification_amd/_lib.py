"""ctypes binding of the C ABI declared in include/unet_hip.h (libunet_hip.so, gfx950).

There is no CPU fallback: if the shared library is missing or no MI355X is visible the
product raises -- it never routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# COVIDSEG_AMD_LIB: another build of the SAME library (the sanitizer build of `make asan`, an A/B build); the default -- and the product -- is the in-tree file
LIB_PATH = os.environ.get("COVIDSEG_AMD_LIB") or os.path.join(_HERE, "libunet_hip.so")
ABI_VERSION = 16
COMM_HANDLE_BYTES, COMM_MAX_WORLD, COMM_MAX_DOUBLES = 64, 8, 2048          # include/unet_hip.h: UNET_COMM_*

ALGO_AUTO, ALGO_NAIVE, ALGO_MFMA = 0, 1, 2          # fp16-split h2 kernels where the shape allows / VALU kernels / strict fp32 MFMA kernels
# unet_ctx_set_option (include/unet_hip.h UNET_OPT_*)
OPTIONS = {"relu_bits": 1, "bn_fold": 2, "enc_bn_fused": 3, "bn_concat_analytic": 4, "bn_fuse_stats": 5, "deterministic": 6, "head_fused": 7, "skip_raw": 8, "pool_sums_fused": 9, "head_bwd_fused": 10, "conv_pp": 13}
ARCH_UNET, ARCH_UNETPP, ARCH_CLASSIFIER = 0, 1, 2
DTYPE_F32, DTYPE_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_ELU = 0, 1, 2
MASK_NONE, MASK_RELU, MASK_ELU, MASK_ELU_DROP = 0, 1, 2, 3
PROG_FWD_TRAIN, PROG_BWD, PROG_FWD_INFER = 0, 1, 2
SYNC_BN_FWD, SYNC_LOSS, SYNC_BN_BWD, SYNC_GRAD_BUCKET = 0, 1, 2, 3

f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
vp = C.c_void_p
i32, i64, u64, f32, f64, sz = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_size_t


class SyncPoint(C.Structure):
    _fields_ = [("after_op", i32), ("kind", i32), ("ptr", vp), ("count", i64), ("use_op", i32), ("reserved", i32)]


# name -> (restype, argtypes); pointers to device memory are passed as void* integers
_PROTOS = {
    "unet_abi_version": (i32, []),
    "unet_ctx_create": (i32, [i32, C.POINTER(vp)]),
    "unet_ctx_destroy": (None, [vp]),
    "unet_last_error": (C.c_char_p, [vp]),
    "unet_ctx_set_profiling": (i32, [vp, i32]),
    "unet_ctx_set_option": (i32, [vp, i32, i32]),
    "unet_ctx_get_option": (i32, [vp, i32]),
    "unet_conv3x3_w_ws_floats": (sz, [i32, i32]),
    "unet_conv3x3_pick_algo": (i32, [i32, i32, i32, i32]),
    "unet_conv3x3_exec_ratio": (f64, [i32, i32, i32, i32, i32]),
    "unet_conv3x3_wgrad_exec_ratio": (f64, [i32, i32, i32, i32, i32]),
    "unet_request_bn_stats": (i32, [vp, i32]),
    "unet_relu_bits_supported": (i32, [i32, i32, i32, i32, i32]),
    "unet_relu_bits_bytes": (sz, [i32, i32, i32, i32]),
    "unet_request_relu_bits": (i32, [vp, vp]),
    "unet_allow_k_slices": (i32, [vp]),
    "unet_ctx_max_kernel_scratch_bytes": (i32, [vp]),
    "unet_conv3x3_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, u64, i32, vp, vp]),
    "unet_conv3x3_bwd_data": (i32, [vp, vp, vp, vp, i32, f32, u64, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "unet_conv3x3_bwd_weights_ws_bytes": (sz, [i32, i32, i32, i32, i32]),
    "unet_conv3x3_bwd_weights": (i32, [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, vp]),
    "unet_convT2x2_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "unet_convT2x2_bwd_data": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "unet_convT2x2_bwd_weights_ws_bytes": (sz, [i32, i32, i32, i32, i32]),
    "unet_convT2x2_bwd_weights": (i32, [vp, vp, vp, i32, vp, vp, vp, sz, i32, i32, i32, i32, i32, i32, vp]),
    # bf16-storage variants (ABI v5): same argument lists minus the algorithm selector
    "unet_cast_f32_to_bf16": (i32, [vp, vp, vp, i64, vp]),
    "unet_cast_bf16_to_f32": (i32, [vp, vp, vp, i64, vp]),
    "unet_conv3x3_fwd_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, u64, vp, vp]),
    "unet_conv3x3_first_fwd_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, u64, vp]),
    "unet_conv3x3_bwd_data_bf16": (i32, [vp, vp, vp, vp, i32, f32, u64, vp, vp, i32, i32, i32, i32, i32, vp]),
    "unet_conv3x3_bwd_weights_ws_bytes_bf16": (sz, [i32, i32, i32, i32, i32]),
    "unet_conv3x3_bwd_weights_bf16": (i32, [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, vp]),
    "unet_conv3x3_first_bwd_weights_bf16": (i32, [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, vp]),
    "unet_convT2x2_fwd_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "unet_convT2x2_bwd_data_bf16": (i32, [vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "unet_convT2x2_bwd_weights_ws_bytes_bf16": (sz, [i32, i32, i32, i32, i32]),
    "unet_convT2x2_bwd_weights_bf16": (i32, [vp, vp, vp, i32, vp, vp, vp, sz, i32, i32, i32, i32, i32, vp]),
    "unet_bn_stats": (i32, [vp, vp, i32, vp, i64, i32, vp]),
    "unet_bn_stats_bf16": (i32, [vp, vp, i32, vp, i64, i32, vp]),
    "unet_conv3x3_bnfold_supported": (i32, [i32, i32, i32, i32, i32]),
    "unet_conv3x3_bnfold_ws_floats": (sz, [i32, i32, i32]),
    "unet_conv3x3_bnfold_fwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]),
    "unet_conv3x3_bnfold_bwd_weights": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp, i32, i32, i32, i32, i32, i32, vp]),
    "unet_bn_stats_concat": (i32, [vp, vp, i32, vp, f64, vp, vp, vp, i64, i32, i32, vp]),
    "unet_bn_stats_concat_bf16": (i32, [vp, vp, i32, vp, f64, vp, vp, vp, i64, i32, i32, vp]),
    "unet_bn_finalize_train": (i32, [vp, vp, f64, vp, vp, vp, vp, vp, i32, vp]),
    "unet_bn_finalize_infer": (i32, [vp, vp, vp, vp, vp, vp, i32, vp]),
    "unet_bn_apply": (i32, [vp, vp, i32, vp, vp, i32, i64, i32, vp]),
    "unet_bn_apply_bf16": (i32, [vp, vp, i32, vp, vp, i32, i64, i32, vp]),
    "unet_bn_bwd_stats": (i32, [vp, vp, i32, vp, i32, vp, vp, i64, i32, vp]),
    "unet_bn_bwd_stats_bf16": (i32, [vp, vp, i32, vp, i32, vp, vp, i64, i32, vp]),
    "unet_bn_bwd_param_grads": (i32, [vp, vp, vp, vp, i32, vp]),
    "unet_bn_bwd_apply": (i32, [vp, vp, i32, vp, i32, vp, vp, f64, i32, f32, u64, vp, i32, i64, i32, vp]),
    "unet_bn_bwd_apply_bf16": (i32, [vp, vp, i32, vp, i32, vp, vp, f64, i32, f32, u64, vp, i32, i64, i32, vp]),
    "unet_maxpool2x2_dropout_fwd": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, f32, u64, vp]),
    "unet_maxpool2x2_dropout_fwd_bf16": (i32, [vp, vp, i32, vp, i32, i32, i32, i32, f32, u64, vp]),
    "unet_maxpool2x2_dropout_bwd": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32, u64, i32, vp]),
    "unet_maxpool2x2_dropout_bwd_bf16": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32, u64, i32, vp]),
    "unet_bn_apply_maxpool_dropout_fwd": (i32, [vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, f32, u64, vp]),
    "unet_bn_apply_maxpool_dropout_fwd_bf16": (i32, [vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, f32, u64, vp]),
    "unet_maxpool2x2_dropout_bwd_bnstats": (i32, [vp, vp, i32, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, f32, u64, vp]),
    "unet_maxpool2x2_dropout_bwd_sums": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, u64, vp]),
    "unet_maxpool2x2_dropout_bwd_sums_bf16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, u64, vp]),
    "unet_bn_maxpool_bwd_apply_bf16": (i32, [vp, vp, i32, vp, vp, f64, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32, u64, vp]),
    "unet_bn_bwd_skip_term": (i32, [vp, vp, vp, vp, vp, vp, i32, f64, vp]),
    "unet_bn_maxpool_bwd_apply": (i32, [vp, vp, i32, vp, vp, f64, vp, i32, vp, vp, i32, i32, i32, i32, i32, f32, u64, vp]),
    "unet_maxpool2x2_dropout_bwd_bnstats_bf16": (i32, [vp, vp, i32, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, f32, u64, vp]),
    "unet_head_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]),
    "unet_head_fwd_bf16": (i32, [vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]),
    "unet_loss_finalize": (i32, [vp, vp, f64, vp, vp]),
    "unet_head_bwd": (i32, [vp, vp, vp, vp, vp, vp, f64, vp, vp, vp, i64, i32, i32, vp]),
    "unet_head_bwd_bf16": (i32, [vp, vp, vp, vp, vp, vp, f64, vp, vp, vp, i64, i32, i32, vp]),
    "unet_adam_keras": (i32, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp]),
    "unet_seg_metrics_sweep": (i32, [vp, vp, vp, vp, i32, vp, i64, vp]),
    "unet_gather_samples": (i32, [vp, vp, vp, i64, i64, vp]),
    "unet_conv3x3_head_supported": (i32, [vp, i32, i32, i32, i32]),
    "unet_conv3x3_head_fwd": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "unet_conv3x3_bwd_data_pool_sums_supported": (i32, [vp, i32, i32, i32, i32]),
    "unet_conv3x3_bwd_data_pool_sums": (i32, [vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "unet_head_dy": (i32, [vp, vp, vp, vp, f64, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "unet_head_bwd_stream_supported": (i32, [vp, i32, i32, i32]),
    "unet_head_dzm": (i32, [vp, vp, vp, vp, f64, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "unet_conv3x3_bwd_data_dzm": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "unet_conv3x3_bwd_weights_dzm": (i32, [vp, vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, vp]),
    "unet_zero": (i32, [vp, vp, sz, vp]),
    "unet_comm_create": (i32, [vp, i32, i32, vp, vp]),
    "unet_comm_connect": (i32, [vp, vp]),
    "unet_comm_set_timeout_ms": (i32, [vp, i32]),
    "unet_comm_allreduce_f64": (i32, [vp, vp, i32, vp]),
    "unet_comm_status": (i32, [vp, vp, vp]),
    "unet_comm_destroy": (None, [vp]),
    "unet_copy_slice": (i32, [vp, vp, i32, vp, i32, i64, i32, vp]),
    "unet_copy_slice_bf16": (i32, [vp, vp, i32, vp, i32, i64, i32, vp]),
    "unet_accum_slices": (i32, [vp, C.POINTER(vp), C.POINTER(i32), i32, vp, i32, i64, i32, i32, vp]),
    "unet_accum_slices_bf16": (i32, [vp, C.POINTER(vp), C.POINTER(i32), i32, vp, i32, i64, i32, i32, vp]),
    "unet_dense_ws_bytes": (sz, [i32, i32, i32]),
    "unet_dense_fwd": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, u64, vp, sz, vp]),
    "unet_dense_fwd_bf16": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, u64, vp, sz, vp]),
    "unet_dense_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "unet_dense_bwd_bf16": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "unet_cls_head_fwd": (i32, [vp, vp, vp, vp, vp, vp, f32, f32, vp, i32, i32, vp]),
    "unet_cls_loss_finalize": (i32, [vp, vp, f64, vp, vp]),
    "unet_cls_head_bwd": (i32, [vp, vp, vp, vp, vp, f32, f32, f64, f32, vp, vp, vp, vp, i32, i32, vp]),
    # image steps in front of the path (kernels_pre.hip)
    "unet_pre_minmax_ws_bytes": (sz, [i32]),
    "unet_pre_minmax_to_u8": (i32, [vp, vp, vp, i32, i64, vp, sz, vp]),
    "unet_pre_unit_to_u8": (i32, [vp, vp, vp, i64, vp]),
    "unet_pre_u8_to_unit": (i32, [vp, vp, vp, i64, vp]),
    "unet_pre_clahe_ws_bytes": (sz, [i32, i32, i32]),
    "unet_pre_clahe_u8": (i32, [vp, vp, vp, i32, i32, i32, f32, i32, i32, vp, sz, vp]),
    "unet_pre_resize_u8": (i32, [vp, vp, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "unet_pre_contours_u8": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, i32]),        # host buffers (ABI v7)
    "unet_model_create": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(vp)]),
    "unet_model_dtype": (i32, [vp]),
    "unet_model_tap_elem_bytes": (i32, [vp, C.c_char_p, i32]),
    "unet_model_destroy": (None, [vp]),
    "unet_model_set_loss_out": (C.c_int32, [vp, vp]),
    "unet_model_param_count": (i64, [vp]),
    "unet_model_state_count": (i64, [vp]),
    "unet_model_workspace_bytes": (sz, [vp, i32]),
    "unet_model_tensor_info": (i32, [vp, C.c_char_p, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64)]),
    "unet_model_bind": (i32, [vp, vp, vp, vp, vp, vp, vp, sz]),
    "unet_model_set_io": (i32, [vp, vp, vp, vp]),
    "unet_model_set_dropout": (i32, [vp, f32, u64]),
    "unet_model_set_class_weights": (i32, [vp, f32, f32]),
    "unet_model_num_ops": (i32, [vp, i32]),
    "unet_model_sync_points": (i32, [vp, i32, C.POINTER(SyncPoint), i32]),
    "unet_model_run": (i32, [vp, i32, i32, i32, vp]),
    "unet_model_loss_ptr": (vp, [vp]),
    "unet_model_tap": (i32, [vp, C.c_char_p, i32, C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                             C.POINTER(i32), C.POINTER(i32)]),
    "unet_model_op_info": (i32, [vp, i32, i32, C.POINTER(C.c_char_p), f64p, f64p, f64p, C.POINTER(i64)]),
    "unet_model_reset_timers": (i32, [vp]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)

_lib = None


class UNetHipError(RuntimeError):
    pass


def load():
    """dlopen libunet_hip.so and attach prototypes.  Raises UNetHipError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UNetHipError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C <package>/csrc`). There is no CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.unet_abi_version() != ABI_VERSION:
        raise UNetHipError(f"ABI mismatch: library {lib.unet_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


class Context:
    """One unet_ctx per (device, option set), shared by the engines built with them.  A unet_ctx carries mutable host state (slot copies that must be zero between
    launches, the statistics bookkeeping, the profiling flag, one ConvT image): engines that share a context must be driven from ONE thread on ONE stream at a time
    -- what every runner of this package does.  An engine that is used from its own thread / stream takes a context of its own (HipUNet(private_context=True))."""
    _cache = {}
    users = 0                                                  # engines holding this context (retain / release)

    def __init__(self, device: int):
        self.lib = load()
        h = vp()
        rc = self.lib.unet_ctx_create(device, C.byref(h))
        if rc != 0:
            raise UNetHipError(
                f"unet_ctx_create(device={device}) failed with status {rc}: no gfx950 (MI355X) device visible. "
                "This engine has no CPU fallback.")
        self.handle = h
        self.device = device

    @classmethod
    def get(cls, device: int, options: dict | None = None, private: bool = False) -> "Context":
        """The shared default-option context of a device, or -- with options -- a private context carrying them
        (unet_ctx_set_option: a model reads the options when it is created, the op-level entry points when they launch)."""
        key = (device, frozenset((k, int(v)) for k, v in options.items())) if options else device
        if private:                                            # not cached, not shared: released (and destroyed) by the engine that asked for it
            ctx = cls(device)
            for k, v in (options or {}).items():
                ctx.check(ctx.lib.unet_ctx_set_option(ctx.handle, OPTIONS[k], int(v)), f"set_option({k})")
            return ctx
        if key not in cls._cache:                              # one context per (device, option set): engines with the same options share it (a context owns 24 MB of device scratch)
            ctx = cls(device)
            for k, v in (options or {}).items():
                ctx.check(ctx.lib.unet_ctx_set_option(ctx.handle, OPTIONS[k], int(v)), f"set_option({k})")
            cls._cache[key] = ctx
        return cls._cache[key]

    def retain(self):
        self.users += 1
        return self

    def release(self):
        """An engine lets go; a private context is destroyed with its last user, a cached one stays for the next engine (close() frees it explicitly)."""
        self.users = max(0, self.users - 1)
        if self.users == 0 and self.handle is not None and not any(v is self for v in type(self)._cache.values()):
            self.close()

    def close(self):
        """unet_ctx_destroy: frees the context's device scratch (slot copies, ConvT image); the object must not be used afterwards.  Refused while engines other
        than the caller's still hold the context."""
        if self.users > 1:
            raise UNetHipError(f"Context.close(): {self.users} engines still use this context (close them first, or build engines with private_context=True)")
        if self.handle is not None:
            for k in [k for k, v in type(self)._cache.items() if v is self]:
                del type(self)._cache[k]
            self.lib.unet_ctx_destroy(self.handle)
            self.handle = None

    def last_error(self) -> str:
        msg = self.lib.unet_last_error(self.handle)
        return msg.decode() if msg else "?"

    def check(self, rc: int, what: str = ""):
        if rc != 0:
            msg = self.lib.unet_last_error(self.handle)
            raise UNetHipError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
