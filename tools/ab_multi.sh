#!/bin/bash
# Same-box A/B/C... of several builds of libunet_hip.so, alternated round-robin on ONE box (boxes differ by ~3-5 %):
#   tools/ab_multi.sh "<lib1> <lib2> ..." [rounds] [extra bench.py args...]        (run on the GPU box through gpurun)
# Prints images/s and ms/step of every run, then the median per build.
set -u
LIBS="$1"; R="${2:-3}"; shift 2 2>/dev/null || shift $#
LIB="$(ls -d one-stop-*_amd)/libunet_hip.so"
cp "$LIB" /tmp/_keep.so
out="${AB_OUT:-gpurun_out/ab_multi.txt}"; mkdir -p "$(dirname "$out")"; : > "$out"
for r in $(seq "$R"); do
  for src in $LIBS; do
    cp "$src" "$LIB"
    python bench.py --no-cpu-baseline --no-strict-leg --no-fit-leg --no-traffic-leg "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$src', d['value'], d['ms_per_step'], d.get('predict_batch1_ms'))" >> "$out"
  done
done
cp /tmp/_keep.so "$LIB"
python - "$out" <<'PY'
import sys, statistics as st
rows=[l.split() for l in open(sys.argv[1])]
for v in dict.fromkeys(r[0] for r in rows):
    ms=[float(r[2]) for r in rows if r[0]==v]
    print(f"{v:28s} median {st.median(ms):8.3f} ms  runs {ms}")
PY
