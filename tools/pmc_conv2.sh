#!/bin/bash
# second PMC set for the Winograd kernels inside the training step: MFMA / VALU co-execution, VMEM and LDS queue pressure
OUT=${1:-gpurun_out/pmc_conv2}
mkdir -p $OUT
export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
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
        if "wgrad_wino2d" not in k and "conv_wino2d_kernel<32" not in k and "conv_wino2d4" not in k: continue
        agg[(k[:50], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in sorted(agg.items()):
    print(key)
    for c, v in sorted(d.items()):
        print(f"   {c:30s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
