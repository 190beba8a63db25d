mkdir -p gpurun_out/r4m
timeout 1800 python -m pytest tests/ -q -m gpu > gpurun_out/r4m/gpu_tests.txt 2>&1; tail -8 gpurun_out/r4m/gpu_tests.txt
