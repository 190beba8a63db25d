#!/bin/bash
# PMC wave-cycle breakdown of the weight-gradient kernels inside the training step (tools/profile_ops.py, 1 profiled step).
# usage (GPU box, repo root): bash tools/pmc_wgrad.sh <outdir>
OUT=${1:-gpurun_out/pmc_wgrad}
mkdir -p $OUT
export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python tools/profile_ops.py --reps 1 --warm 2 > $OUT/$tag.log 2>&1 || true
done
python - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wgrad_wino" not in k and "conv_wino2d_kernel<32" not in k: continue
        agg[(k[:50], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in sorted(agg.items()):
    print(key)
    for c, v in sorted(d.items()):
        print(f"   {c:30s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
