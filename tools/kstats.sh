#!/bin/bash
# per-kernel time per step of the default bench under an environment switch: bash tools/kstats.sh <tag> [VAR=val ...]
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/ks_$tag
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag/trace -- python bench.py --steps 10 --warmup 15 --no-cpu-baseline > gpurun_out/ks_$tag/trace.log 2>&1
f=$(find gpurun_out/ks_$tag/trace -name "*kernel_stats.csv" | head -1)
python - "$f" "$tag" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = None
for r in rows:
    if "adam_kernel" in r["Name"]: steps = int(r["Calls"])
tot = 0.0
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / steps; tot += ms
    if ms > 0.02: print(sys.argv[2], r["Name"][:64].ljust(64), int(r["Calls"]) // steps if int(r["Calls"]) >= steps else r["Calls"], "%.3f ms/step" % ms, "%.1f us" % (float(r["AverageNs"]) / 1e3))
print(sys.argv[2], "TOTAL kernel ms/step %.3f" % tot)
PY
