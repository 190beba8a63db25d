#!/bin/bash
# Sample power / clocks while the bench runs (evidence for the power-limited regime).
python bench.py --steps 150 --warmup 15 --no-cpu-baseline > gpurun_out/bench_clk.json 2>/dev/null &
BP=$!
sleep 6
for i in $(seq 1 8); do amd-smi metric -g 0 --power --clock 2>/dev/null | grep -E "SOCKET_POWER|GFX_0|CLK:|MIN_CLK|MAX_CLK|CLK_LOCKED|DEEP_SLEEP|THROTTLE" | head -12 | tr '\n' ' ' | sed 's/  */ /g'; echo; sleep 0.5; done > gpurun_out/clocks.txt
wait $BP
cat gpurun_out/clocks.txt
amd-smi static -g 0 --limit 2>/dev/null | grep -iE "power|cap" | head -8
python -c "import json; d=json.load(open('gpurun_out/bench_clk.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
