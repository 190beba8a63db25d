# usage: bash tools/gpu/comm_ab.sh <out dir> [rounds]   -- what the data-parallel program costs at ONE rank (every all-reduce is the identity):
# the plain step against the process-group step with the BatchNorm / loss sums through csrc/comm.hip ("device") and through RCCL ("rccl"), alternated on one box
O=$1; R=${2:-3}
mkdir -p $O
: > $O/comm_ab.txt
for r in $(seq 1 $R); do
  for m in plain device rccl buckets; do
    if [ $m = plain ]; then
      python bench.py --steps 30 --warmup 5 --no-fit-leg --no-strict-leg --no-cpu-baseline 2>/dev/null | tail -1 > $O/line.json
    elif [ $m = buckets ]; then          # gradient buckets only (no BatchNorm / loss reductions): what the program and the RCCL side stream cost by themselves
      UNET_BENCH_FORCE_PG=1 python bench.py --steps 30 --warmup 5 --no-fit-leg --no-strict-leg --no-cpu-baseline --no-sync-bn 2>/dev/null | tail -1 > $O/line.json
    else
      UNET_BENCH_FORCE_PG=1 python bench.py --steps 30 --warmup 5 --no-fit-leg --no-strict-leg --no-cpu-baseline --small-allreduce $m 2>/dev/null | tail -1 > $O/line.json
    fi
    python - $m $O/line.json >> $O/comm_ab.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], d["ms_per_step"], d["value"], d["config"].get("small_allreduce"))
PY
  done
done
cat $O/comm_ab.txt
python - $O/comm_ab.txt <<'PY'
import sys, statistics as st
rows = [l.split() for l in open(sys.argv[1])]
for m in ("plain", "device", "rccl", "buckets"):
    v = [float(r[1]) for r in rows if r[0] == m]
    print(m, "median ms", st.median(v))
PY
