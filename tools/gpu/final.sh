# the round's last GPU call: the two headline bench lines on a fresh box first, then the whole -m gpu suite and smoke()
mkdir -p gpurun_out/r4q
python bench.py > gpurun_out/r4q/bench.json 2> gpurun_out/r4q/bench.err; tail -c 400 gpurun_out/r4q/bench.json | head -c 300; echo
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4q/bench_driver_flags.json 2>> gpurun_out/r4q/bench.err
timeout 1800 python -m pytest tests/ -q -m gpu > gpurun_out/r4q/gpu_tests.txt 2>&1; tail -4 gpurun_out/r4q/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
