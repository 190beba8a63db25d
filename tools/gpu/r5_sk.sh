O=gpurun_out/r5sk; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv3x3" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python tools/gpu/predict_trace.py 200 1 2>&1 | tail -1
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_fit_path.py tests/test_gpu_dp.py -q -m gpu -x > $O/tests_model.txt 2>&1; tail -3 $O/tests_model.txt
bash tools/gpu/r5_predict_prof.sh > $O/predict_trace.txt 2>&1; tail -30 $O/predict_trace.txt
