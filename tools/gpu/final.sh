# the round's last GPU call: the two headline bench lines on a fresh box first, then the whole -m gpu suite and smoke()
O=gpurun_out/final; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json | head -c 300; echo
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_flags.json 2>> $O/bench.err
timeout 1800 python -m pytest tests/ -q -m gpu > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
