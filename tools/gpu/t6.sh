set -x
mkdir -p gpurun_out/r4g
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "head" > gpurun_out/r4g/t_ops.txt 2>&1; tail -3 gpurun_out/r4g/t_ops.txt
timeout 1800 python -m pytest tests/test_gpu_bf16_model.py tests/test_gpu_dp.py -x -q -m gpu > gpurun_out/r4g/t_model.txt 2>&1; tail -5 gpurun_out/r4g/t_model.txt
timeout 1800 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r4g/t_full.txt 2>&1; tail -5 gpurun_out/r4g/t_full.txt
