mkdir -p gpurun_out/r4o
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "convT" > gpurun_out/r4o/t_ops.txt 2>&1; tail -3 gpurun_out/r4o/t_ops.txt
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "fp32" > gpurun_out/r4o/t_full.txt 2>&1; tail -3 gpurun_out/r4o/t_full.txt
AB_OUT=gpurun_out/r4o/ab.txt bash tools/ab_bench.sh build/ab/prev.so build/ab/bw2.so 3 --steps 30 --warmup 5 --no-fit-leg | tee gpurun_out/r4o/ab_summary.txt
python tools/profile_ops.py > gpurun_out/r4o/ops.txt 2>&1; grep -E "convT_wgrad|sum of" gpurun_out/r4o/ops.txt
