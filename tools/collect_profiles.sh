#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench command          -> <out>/kernel_stats.csv
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE; counters only)      -> <out>/pmc_traffic.json  (HBM bytes per launch of the dominant kernel)
# usage: bash tools/collect_profiles.sh gpurun_out/prof
set -u
OUT=${1:-gpurun_out/prof}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 10 --warmup 15 --no-cpu-baseline > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python tools/profile_ops.py --reps 1 --warm 3 > $OUT/pmc_$c.log 2>&1
done
python tools/summarize_profiles.py $OUT
