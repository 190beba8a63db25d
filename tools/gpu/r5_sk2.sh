O=gpurun_out/r5sk; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "k_slices" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python tools/gpu/predict_trace.py 200 1 2>&1 | tail -1
python tools/gpu/predict_trace.py 200 1 2>&1 | tail -1
