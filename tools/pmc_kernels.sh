#!/bin/bash
# PMC counters of selected kernels inside one profiled training step (tools/profile_ops.py).  Counter sets are separate rocprofv3 passes (counters only).
# usage (GPU box, repo root): bash tools/pmc_kernels.sh <outdir> "<kernel-name substrings, | separated>" "<counter set 1>" ["<counter set 2>" ...]
OUT=$1; FILT=$2; shift 2
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/set$i -- python tools/profile_ops.py --reps 1 --warm 2 > $OUT/set$i.log 2>&1 || true
done
python - $OUT "$FILT" <<'PY'
import csv, glob, collections, sys, json, re
out, filt = sys.argv[1], sys.argv[2].split("|")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(s in k for s in filt): continue
        name = re.sub(r"\(anonymous namespace\)::|void ", "", k).split("(")[0][:70]
        agg[name + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for key, d in sorted(agg.items()):
    res[key] = {c: sum(v) / len(v) for c, v in d.items()}
    res[key]["launches_seen"] = max(len(v) for v in d.values())
json.dump(res, open(out + "/pmc_summary.json", "w"), indent=1)
for key, d in res.items():
    print(key)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {v:18.0f}")
PY
