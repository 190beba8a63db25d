set -x
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv3x3 or convT" > gpurun_out/r4b/ops_tests.txt 2>&1; tail -3 gpurun_out/r4b/ops_tests.txt
AB_OUT=gpurun_out/r4b/ab.txt bash tools/ab_bench.sh build/ab/base.so build/ab/lds.so 3 --steps 30 --warmup 5 | tee gpurun_out/r4b/ab_summary.txt
python tools/profile_ops.py > gpurun_out/r4b/ops.txt 2>&1
