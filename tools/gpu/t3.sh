set -x
mkdir -p gpurun_out/r4d
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "head" > gpurun_out/r4d/t_ops.txt 2>&1; tail -15 gpurun_out/r4d/t_ops.txt
timeout 1800 python -m pytest tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r4d/t_model.txt 2>&1; tail -15 gpurun_out/r4d/t_model.txt
python tools/profile_ops.py > gpurun_out/r4d/ops.txt 2>&1; grep -E "c9b|head|sum of" gpurun_out/r4d/ops.txt
AB_OUT=gpurun_out/r4d/ab.txt bash tools/ab_bench.sh build/ab/lds.so build/ab/head.so 3 --steps 30 --warmup 5 --no-fit-leg | tee gpurun_out/r4d/ab_summary.txt
