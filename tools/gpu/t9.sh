mkdir -p gpurun_out/r4l
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py tests/test_gpu_fit_path.py -q -m gpu > gpurun_out/r4l/t_model.txt 2>&1; tail -12 gpurun_out/r4l/t_model.txt
