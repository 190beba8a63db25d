"""Seeded inputs of the full-size known-answer cases (BASELINE.json configs[1], [3], [4] at their stated sizes).
Shared by the generator (make_fullsize_goldens.py: fp64 CPU oracle, run once in the build container) and by
tests/test_gpu_fullsize.py (HIP engine on the MI355X): both rebuild the same weights and batches from seeds
(numpy Generator streams are stable across versions) and verify the checksums stored in the fixture."""
import numpy as np

CASES = {
    # name: (arch, size, batch, data seed, weight seed)
    "unet_512_bs16": ("unet", 512, 16, 11, 21),          # configs[1]: U-Net infection seg 512x512x1 bs16
    "unet_512_bs8": ("unet", 512, 8, 14, 24),            # configs[2]: U-Net lung seg (T3:850-913, the same graph) 512x512x1, the per-rank batch 8 of global 64 on 8 GPUs
    "unetpp_256_bs32": ("unetpp", 256, 32, 12, 22),      # configs[3]: U-Net++ 256x256 bs32
    "cls_224_bs256": ("classifier", 224, 256, 13, 23),   # configs[4]: slice classifier 224x224 bs256 (1 channel, as the reference feeds it)
    "cls_224x3_bs256": ("classifier", 224, 256, 15, 25), # configs[4] AS WRITTEN in BASELINE.json: 224x224x3 bs256 (a 3-channel image: first conv 3 -> 16)
}
IN_CH = {"cls_224x3_bs256": 3}                           # every other case: 1 channel
# the per-tensor gradient bound of tests/test_gpu_fullsize.py is max(3e-4, 4 x min(E_k, cap)), E_k = distance of the fp32-CPU evaluation of the same step from float64.
# cap = 2.5e-3 by default; the 3-channel classifier is the most flip-sensitive case (its own fp32-CPU run lands 1e-3 ... 8e-3 from float64 in the first two blocks)
EK_CAP = {"cls_224x3_bs256": 1e-2}
FULL_GRADS = {"unet": ("c1a/kernel", "c9b/kernel", "out/kernel", "bn1/gamma", "bn9/beta", "u9/bias", "c5b/bias"),
              "unetpp": ("c1a/kernel", "x1_4b/kernel", "out/kernel", "bn1/gamma", "x1_4bbn/beta", "u1_4/bias", "c4b/bias"),
              "classifier": ("c1a/kernel", "c1b/kernel", "fc2/kernel", "bn1a/gamma", "bn3b/beta", "fc1/bias", "c3b/bias")}


def build(name):
    """-> (arch, weights dict, x, y).  Biases / BN betas are made non-zero and gammas non-unit so no term of the graph is trivially absent."""
    from oracle import unet_oracle as O
    from covidseg_amd.data import synthetic_classification, synthetic_ct
    arch, size, n, dseed, wseed = CASES[name]
    if arch == "classifier":
        x, y = synthetic_classification(n, size, seed=dseed, channels=IN_CH.get(name, 1))
        y = y.astype(np.float32)
        w = O.cls_init_weights(wseed, IN_CH.get(name, 1), (size, size))
    else:
        x, y = synthetic_ct(n, size, seed=dseed)
        w = O.init_weights(seed=wseed) if arch == "unet" else O.pp_init_weights(seed=wseed)
    rng = np.random.default_rng(1000 + wseed)
    for k in w:
        if k.endswith("/bias") or k.endswith("/beta"):
            w[k] = (rng.standard_normal(w[k].shape) * 0.1).astype(np.float32)
        elif k.endswith("/gamma"):
            w[k] = rng.uniform(0.5, 1.5, w[k].shape).astype(np.float32)
    return arch, w, x, y


def checksums(w, x, y):
    return (float(sum(float(np.abs(v.astype(np.float64)).sum()) for v in w.values())), float(x.astype(np.float64).sum()),
            float(np.asarray(y, np.float64).sum()))
