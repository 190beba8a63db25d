mkdir -p gpurun_out/r4i
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r4i/gpu_tests.txt 2>&1; tail -8 gpurun_out/r4i/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4i/smoke.txt 2>&1; tail -2 gpurun_out/r4i/smoke.txt
