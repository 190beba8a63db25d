mkdir -p gpurun_out/r4p
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "convT" > gpurun_out/r4p/t_ops.txt 2>&1; tail -3 gpurun_out/r4p/t_ops.txt
AB_OUT=gpurun_out/r4p/ab.txt bash tools/ab_bench.sh build/ab/bw2.so build/ab/par.so 3 --steps 30 --warmup 5 --no-fit-leg | tee gpurun_out/r4p/ab_summary.txt
python tools/profile_ops.py > gpurun_out/r4p/ops.txt 2>&1; grep -E "convT_wgrad|sum of" gpurun_out/r4p/ops.txt
