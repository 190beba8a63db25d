set -x
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_fit_path.py tests/test_gpu_bench_cli.py -x -q -m gpu > gpurun_out/r4c/tests.txt 2>&1; tail -15 gpurun_out/r4c/tests.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err; tail -c 1500 gpurun_out/r4c/bench.json
