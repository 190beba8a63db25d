O=gpurun_out/r5pp; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/tools/gpu/predict_trace.py 200 1 > $R/$O/predict.txt 2>&1
cd $R
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/predict_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/predict_kernel_trace.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r5pp/predict_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
tail=rows[-40:]
t0=int(tail[0]['Start_Timestamp'])
for r in tail:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  grid {r.get('Grid_Size_X','?'):>8s} wg {r.get('Workgroup_Size_X','?'):>4s}  {r['Kernel_Name'][:70]}")
PY
rm -rf $O/prof $O/predict_kernel_trace.csv
