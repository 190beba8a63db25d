mkdir -p gpurun_out/r4n
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "pooled_sums or maxpool or encoder_tail" > gpurun_out/r4n/t_ops.txt 2>&1; tail -5 gpurun_out/r4n/t_ops.txt
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -q -m gpu > gpurun_out/r4n/t_model.txt 2>&1; tail -6 gpurun_out/r4n/t_model.txt
for r in 1 2 3; do
for o in '{"pool_sums_fused":0}' '{"pool_sums_fused":1}'; do
python bench.py --no-cpu-baseline --no-strict-leg --no-fit-leg --steps 30 --warmup 5 --options "$o" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$o', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r4n/ab.txt
done; done
python tools/profile_ops.py > gpurun_out/r4n/ops.txt 2>&1; grep -E "pool|dgrad:c[2345]a|sum of" gpurun_out/r4n/ops.txt
