# round-5 baseline on one box: bench lines first (fresh box), per-op table, then the GPU suite + smoke
O=gpurun_out/r5base; mkdir -p $O
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_flags.json 2> $O/bench.err; tail -c 600 $O/bench_driver_flags.json | head -c 500; echo
python bench.py --config 2 --no-cpu-baseline --no-strict-leg --no-fit-leg > $O/bench_cfg2.json 2>> $O/bench.err
python tools/profile_ops.py > $O/ops_table.txt 2>&1
timeout 1500 python -m pytest tests/ -q -m gpu -x > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
