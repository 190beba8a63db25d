O=gpurun_out/r5tests; mkdir -p $O
timeout 1700 python -m pytest tests/ -q -m gpu -x > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
