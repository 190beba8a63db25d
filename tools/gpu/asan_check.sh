#!/bin/bash
# Host-side sanitizer run: the UBSan + libstdc++-assertions build of libunet_hip.so (`make -C <package>/csrc ubsan`, host objects only; default) or,
# with `asan` as the first argument, the ASan + UBSan build with the sanitizer runtime preloaded (which the HIP runtime of this image does not survive: its HSA allocation
# interceptor fails at the first device allocation -- profiles/r05_sanitizer.txt); builds the three graphs (plans of two batch sizes, all option sets of the options test), runs training steps, inference, taps, the DP program
# at world 1 and tears everything down.  Any ASan / UBSan report goes to gpurun_out/asan/report.* and fails the script.
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/asan; mkdir -p $OUT; rm -f $OUT/report.*
MODE=${1:-ubsan}
RT=""
if [ "$MODE" = asan ]; then RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1); fi
LIB=$PWD/build/$MODE/libunet_hip_$MODE.so
[ -f "$LIB" ] || make -C one-stop-*/csrc $MODE -j8 >/dev/null || exit 1
export COVIDSEG_AMD_LIB=$LIB
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:log_path=$PWD/$OUT/report:abort_on_error=0
export UBSAN_OPTIONS=print_stacktrace=1:log_path=$PWD/$OUT/report
LD_PRELOAD=$RT timeout 900 python - <<'PY' > $OUT/run.log 2>&1
import numpy as np, torch, sys
sys.path.insert(0, ".")
from covidseg_amd.data import synthetic_ct, synthetic_classification
from covidseg_amd.engine import HipUNet
from covidseg_amd import weights as W, _lib
print("library:", _lib.LIB_PATH)
x, y = synthetic_ct(3, 64, seed=1)
first = {}
for arch, opts in (("unet", None), ("unet", {"deterministic": 1}), ("unet", {"bn_fold": 0}), ("unet", {"head_fused": 0, "skip_raw": 0}), ("unet", {"relu_bits": 0, "pool_sums_fused": 0}),
                   ("unetpp", None), ("classifier", None)):
    eng = HipUNet(64, 64, 1, device=0, arch=arch, options=opts, dropout_rate=0.25 if arch == "unet" else 0.2, private_context=True)
    eng.set_weights(W.init_weights(0, 1, arch, (64, 64)))
    if arch == "classifier":
        xc, yc = synthetic_classification(6, 64, seed=2); yc = yc.astype(np.float32)
        for n in (6, 4):
            print(arch, n, eng.train_batch(xc[:n], yc[:n]).cpu().numpy())
        eng.predict_batch(xc[:2], yc[:2])
    else:
        for n in (3, 2):
            ld = eng.train_batch(x[:n], y[:n]).cpu().numpy()
            print(arch, opts, n, ld)
            # the instrumented library must compute what the product computes: a finite BCE + Dice loss of an untrained net on 11 % foreground, a Dice coefficient in (0, 1)
            assert np.isfinite(ld).all() and 0.3 < ld[0] < 2.0 and 0.0 < ld[1] < 0.9, (arch, opts, n, ld)
            if arch == "unet": first.setdefault(n, ld); assert np.abs(ld - first[n]).max() < 2e-2, (opts, n, ld, first[n])          # every U-Net graph form starts from the same weights
        p, ld = eng.predict_batch(x[:1], y[:1]); eng.predict_batch(x[:1])
        eng.tap(1, "c9b" if arch == "unet" else "x1_4b"); eng.tap(1, "bn1")
        eng.get_grads(); eng.get_weights()
    eng.close()
# the data-parallel program at world 1 (sync points, comm.hip, buckets)
import torch.distributed as dist, tempfile, os
dist.init_process_group("gloo", rank=0, world_size=1, init_method="file://" + os.path.join(tempfile.mkdtemp(), "s"))
eng = HipUNet(64, 64, 1, device=0, process_group=dist.group.WORLD, force_dp=True, private_context=True)
eng.set_weights(W.init_weights(0, 1, "unet", (64, 64)))
print("dp", eng.train_batch(x, y).cpu().numpy(), eng.comm_status())
eng.close(); dist.destroy_process_group()
print("ASAN_RUN_DONE")
PY
rc=$?
tail -5 $OUT/run.log
n=$(ls $OUT/report.* 2>/dev/null | wc -l)
echo "exit $rc, sanitizer report files: $n"
if [ $rc -ne 0 ] || [ $n -ne 0 ] || ! grep -q ASAN_RUN_DONE $OUT/run.log; then head -60 $OUT/report.* 2>/dev/null; exit 1; fi
echo "sanitizer check ($MODE): clean"
