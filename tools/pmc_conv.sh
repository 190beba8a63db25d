#!/bin/bash
# PMC passes over the per-layer conv benchmark (tools/conv_wino_bench.py): wave-cycle breakdown of the conv kernels.
# usage (on the GPU box, from the repo root): bash tools/pmc_conv.sh <outdir>
set -e
OUT=${1:-gpurun_out/pmc_conv}
mkdir -p $OUT
export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python tools/conv_wino_bench.py --layers c3b,c1b --reps 3 > $OUT/$tag.log 2>&1 || true
done
python - <<'PY'
import csv, glob, collections, sys, os
out = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("OUT", "gpurun_out/pmc_conv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_" not in k: continue
        key = (k[:60], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in sorted(agg.items()):
    print(key)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
