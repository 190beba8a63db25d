bash tools/gpu/r5_upp.sh cur
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_unetpp.py tests/test_gpu_classifier.py -q -m gpu -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-strict-leg --no-fit-leg --steps 30 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('predict_batch1_ms'), d.get('predict_batch1_cold_ms'))"
