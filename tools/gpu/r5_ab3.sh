# usage: r5_ab3.sh <out> <libA> <libB> [pytest -k expr]
O=gpurun_out/$1; mkdir -p $O; A=$2; B=$3
if [ -n "$4" ]; then timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "$4" > $O/tests.txt 2>&1; tail -2 $O/tests.txt; fi
bash tools/ab_ops.sh $A $B 3 > $O/ab_ops.txt 2>&1; grep -E "conv3x3_(fwd|dgrad)|convT|sum" $O/ab_ops.txt | head -70
AB_OUT=$O/ab.txt bash tools/ab_bench.sh $A $B 3 --steps 30 --warmup 5 --no-fit-leg | tee $O/ab_summary.txt
