# usage: bash tools/gpu/opt_ab.sh <out dir> '<options JSON A>' '<options JSON B>' [rounds]   -- same-box A/B of two option sets (bench.py --options), alternated
O=$1; A=$2; B=$3; R=${4:-3}
mkdir -p $O
: > $O/opt_ab.txt
for r in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then J="$A"; else J="$B"; fi
    python bench.py --steps 30 --warmup 5 --no-fit-leg --no-strict-leg --no-cpu-baseline --options "$J" 2>/dev/null | tail -1 > $O/line_$v.json
    python -c "import json,sys; d=json.load(open('$O/line_$v.json')); print('$v', d['ms_per_step'], d['value'])" >> $O/opt_ab.txt
  done
done
cat $O/opt_ab.txt
python - $O/opt_ab.txt <<'PY'
import sys, statistics as st
rows = [l.split() for l in open(sys.argv[1])]
for m in "AB":
    v = [float(r[1]) for r in rows if r[0] == m]
    print(m, "median ms", st.median(v))
PY
