#!/bin/bash
# PMC wave-cycle breakdown of every kernel of a training step (tools/profile_ops.py, 4 steps).  usage: bash tools/pmc_model.sh <outdir> [kernel substring]
OUT=${1:-gpurun_out/pmc_model}; PAT=${2:-wgrad_wino}
mkdir -p $OUT; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python tools/profile_ops.py --reps 1 --warm 3 > $OUT/$tag.log 2>&1 || true
done
PAT=$PAT OUT=$OUT python - <<'PY'
import csv, glob, collections, os
out, pat = os.environ["OUT"], os.environ["PAT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat not in k: continue
        key = k.replace("void ", "").replace("(anonymous namespace)::", "")[:50]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
for key, d in sorted(agg.items()):
    print(key)
    wc = d.get("SQ_WAVE_CYCLES", 1)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:18.0f}  {v / wc:8.3f} of WAVE_CYCLES")
PY
