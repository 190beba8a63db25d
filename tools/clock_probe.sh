#!/bin/bash
# Sample shader clock / power while the bench runs (evidence for the DVFS-limited regime).
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|Average Graphics" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clocks.txt &
SAMP=$!
python bench.py --steps 60 --warmup 5 --no-cpu-baseline > gpurun_out/bench_clk.json 2>/dev/null
wait $SAMP
sort gpurun_out/clocks.txt | uniq -c | sort -rn | head -12
cat gpurun_out/bench_clk.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
