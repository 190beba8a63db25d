#!/bin/bash
# Same-box A/B of two builds of libunet_hip.so with the per-op profile (box-to-box and run-to-run spread is 3-5 %: never compare across gpurun calls).
#   tools/ab_ops.sh build/exp/libunet_a.so build/exp/libunet_b.so [rounds] [profile_ops args...]
# Writes gpurun_out/ab/{a,b}_<round>.txt and prints the per-op medians side by side.
A=$1; B=$2; R=${3:-3}; shift 3
PK=one-stop-for-covid-19-infection-and-lung-segmentation-plus-classification_amd
mkdir -p gpurun_out/ab; cp $PK/libunet_hip.so /tmp/libunet_keep.so
for r in $(seq 1 $R); do
  cp $A $PK/libunet_hip.so; python tools/profile_ops.py "$@" > gpurun_out/ab/a_$r.txt 2>&1
  cp $B $PK/libunet_hip.so; python tools/profile_ops.py "$@" > gpurun_out/ab/b_$r.txt 2>&1
done
cp /tmp/libunet_keep.so $PK/libunet_hip.so
python - "$R" <<'PY'
import sys, statistics as st
R = int(sys.argv[1])
def load(tag):
    rows = {}
    order = []
    for r in range(1, R + 1):
        for ln in open(f"gpurun_out/ab/{tag}_{r}.txt"):
            f = ln.split()
            if len(f) == 6 and f[0] != "op":
                try: ms = float(f[1])
                except ValueError: continue
                if f[0] not in rows: order.append(f[0])
                rows.setdefault(f[0], []).append(ms)
    return order, {k: st.median(v) for k, v in rows.items()}
order, a = load("a"); _, b = load("b")
ta = tb = 0.0
for k in order:
    if k not in b: continue
    ta += a[k]; tb += b[k]
    d = (b[k] - a[k]) / a[k] * 100 if a[k] > 0 else 0
    if abs(d) >= 2.0 and max(a[k], b[k]) > 0.02: print(f"{k:28s} {a[k]:7.3f} {b[k]:7.3f} {d:+6.1f}%")
print(f"{'sum':28s} {ta:7.3f} {tb:7.3f} {(tb - ta) / ta * 100:+6.1f}%")
PY
