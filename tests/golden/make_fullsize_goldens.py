"""Known answers for ONE training-mode forward + backward at the sizes BASELINE.json states:

    unet_512_bs16    configs[1]  U-Net (T1:853-916) 512x512x1, batch 16
    unetpp_256_bs32  configs[3]  U-Net++ (UPP:858-950) 256x256x1, batch 32
    cls_224_bs256    configs[4]  slice classifier (T2:747-776) 224x224x1, batch 256

computed by the CPU oracle in FLOAT64 (dropout off, training-mode BatchNorm; each block recomputed in backward so the step
fits in this container's 62 GB: `ckpt=True`, same arithmetic).  Stored per case: loss, dice_coeff / f1, the norm and the sum
of EVERY parameter gradient, seven gradients in full, every 1009th probability (classifier: all), the batch statistics of
every BatchNorm, and checksums of the seeded inputs.  tests/test_gpu_fullsize.py runs the same step on the HIP engine.

    python tests/golden/make_fullsize_goldens.py [case ...]       (about 6 + 5 + 3 minutes on 8 cores, 40 GB peak)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, HERE)
from oracle import unet_oracle as O          # noqa: E402
import fullsize_cases as FC                  # noqa: E402

P_STRIDE = 1009


def bf16_round(a):
    """round-to-nearest-even to bf16, returned as float32 (what the engine's bf16-storage mode does to the conv / ConvT kernels when it lays them out)"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def main_bf16store(names):
    """--store bf16: the same step with bf16 STORAGE emulated in the float64 oracle (oracle.store_bf16 at every activation / activation gradient the engine materialises,
    conv / ConvT kernels rounded to bf16 as the engine lays them out) -> fullsize_<case>_bf16store.npz.  U-Net cases only (the oracle emulates the storage points of that
    graph).  tests/test_gpu_fullsize.py measures the engine's bf16-storage step against BOTH this fixture and the plain float64 one."""
    for name in names:
        arch, w, x, y = FC.build(name)
        assert arch == "unet", "storage emulation exists for the U-Net graph"
        for k in w:
            if k.endswith("/kernel") and k not in ("c1a/kernel", "out/kernel"):
                w[k] = bf16_round(w[k])
        t0 = time.time()
        r = O.loss_and_grads(w, x, y, dtype=torch.float64, ckpt=True, store=O.store_bf16)
        arrs = dict(loss=np.float64(r["loss"]), metric=np.float64(r["dice"]), p_sample=r["p"].reshape(-1)[::P_STRIDE].astype(np.float32), p_mean=np.float64(r["p"].mean()))
        for k, g in r["grads"].items():
            arrs["gnorm/" + k] = np.float64(np.linalg.norm(g)); arrs["gsum/" + k] = np.float64(g.sum())
        for k in FC.FULL_GRADS[arch]:
            arrs["grad/" + k] = r["grads"][k].astype(np.float32)
        out = os.path.join(HERE, f"fullsize_{name}_bf16store.npz")
        np.savez_compressed(out, **arrs)
        print(f"{name} (bf16 storage emulated): {time.time() - t0:.0f} s, loss {r['loss']:.9f}, dice {r['dice']:.9f}, wrote {out} ({os.path.getsize(out)} bytes)", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--store" and sys.argv[2] == "bf16":
        return main_bf16store(sys.argv[3:] or ["unet_512_bs16"])
    for name in (sys.argv[1:] or list(FC.CASES)):
        arch, w, x, y = FC.build(name)
        t0 = time.time()
        fn = {"unet": O.loss_and_grads, "unetpp": O.pp_loss_and_grads, "classifier": O.cls_loss_and_grads}[arch]
        r = fn(w, x, y, dtype=torch.float64, ckpt=True)
        ws, xs, ys = FC.checksums(w, x, y)
        arrs = dict(loss=np.float64(r["loss"]), metric=np.float64(r["f1"] if arch == "classifier" else r["dice"]),
                    w_checksum=np.float64(ws), x_sum=np.float64(xs), y_sum=np.float64(ys),
                    p_sample=(r["p"].reshape(-1) if arch == "classifier" else r["p"].reshape(-1)[::P_STRIDE]).astype(np.float32),
                    p_mean=np.float64(r["p"].mean()))
        for k, g in r["grads"].items():
            arrs["gnorm/" + k] = np.float64(np.linalg.norm(g)); arrs["gsum/" + k] = np.float64(g.sum())
        for k in FC.FULL_GRADS[arch]:
            arrs["grad/" + k] = r["grads"][k].astype(np.float32)
        for k, (mu, va, n) in r["bn_stats"].items():
            arrs["bn_mean/" + k] = mu.astype(np.float32); arrs["bn_var/" + k] = va.astype(np.float32)
        # calibration: how far the SAME step in fp32 on the CPU (torch / oneDNN -- the stand-in for the reference's fp32 Keras run) lands from
        # the fp64 answer.  At these pixel counts ~1e-6 of the ReLU pre-activations round to the other side of 0 in fp32; each flip is a
        # discontinuity of the gradient and the first layers' weight gradients move by ~sqrt(flip fraction).  The GPU test bounds the engine's
        # error by max(3e-4, 4 x this) per tensor.
        r32 = fn(w, x, y, dtype=torch.float32, ckpt=True)
        for k, g in r["grads"].items():
            e = float(np.linalg.norm(r32["grads"][k].astype(np.float64) - g) / (np.linalg.norm(g) + 1e-30))
            arrs["fp32ref_relerr/" + k] = np.float64(e)
        arrs["fp32ref_loss_err"] = np.float64(abs(r32["loss"] - r["loss"]))
        out = os.path.join(HERE, f"fullsize_{name}.npz")
        np.savez_compressed(out, **arrs)
        print(f"{name}: {time.time() - t0:.0f} s, loss {r['loss']:.9f}, metric {float(arrs['metric']):.9f}, wrote {out} ({os.path.getsize(out)} bytes)", flush=True)


if __name__ == "__main__":
    main()
