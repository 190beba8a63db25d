#!/bin/bash
# Same-box A/B of two builds of libunet_hip.so (the pool's boxes differ by ~5 %, so only back-to-back runs on ONE box compare):
#   tools/ab_bench.sh tmp_ab/old.so tmp_ab/new.so [rounds] [extra bench.py args...]     (run it on the GPU box: gpurun -- 'bash tools/ab_bench.sh ...')
# Alternates the two libraries and prints images/s and ms/step of every run, then the medians.
set -u
A="$1"; B="$2"; R="${3:-3}"; shift 3 2>/dev/null || shift $#
LIB="$(ls -d one-stop-*_amd)/libunet_hip.so"
cp "$LIB" /tmp/_keep.so
out="${AB_OUT:-gpurun_out/ab.txt}"; mkdir -p "$(dirname "$out")"; : > "$out"
for r in $(seq "$R"); do
  for v in A B; do
    src="$A"; [ "$v" = B ] && src="$B"
    cp "$src" "$LIB"
    python bench.py --no-cpu-baseline --no-strict-leg --no-traffic-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])" >> "$out"
  done
done
cp /tmp/_keep.so "$LIB"
python - "$out" <<'PY'
import sys, statistics as st
rows=[l.split() for l in open(sys.argv[1])]
for v in "AB":
    ms=[float(r[2]) for r in rows if r[0]==v]
    print(v, "median ms", st.median(ms), "runs", ms)
PY
