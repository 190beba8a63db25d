# usage: bash tools/gpu/ab.sh <out dir> <lib A> <lib B> ["pytest args"]
O=$1; A=$2; B=$3; T=${4:-}
mkdir -p $O
if [ -n "$T" ]; then timeout 1800 python -m pytest $T -q -m gpu > $O/tests.txt 2>&1; tail -4 $O/tests.txt; fi
AB_OUT=$O/ab.txt bash tools/ab_bench.sh $A $B 3 --steps 30 --warmup 5 --no-fit-leg | tee $O/ab_summary.txt
