mkdir -p gpurun_out/r4k
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py -x -q -m gpu > gpurun_out/r4k/t_model.txt 2>&1; tail -8 gpurun_out/r4k/t_model.txt
for r in 1 2 3; do
for o in '{"skip_raw":0}' '{"skip_raw":1}'; do
python bench.py --no-cpu-baseline --no-strict-leg --no-fit-leg --steps 30 --warmup 5 --options "$o" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$o', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r4k/ab.txt
done; done
python tools/profile_ops.py > gpurun_out/r4k/ops.txt 2>&1; grep -E "bn_apply_pool|bn_fold_prepare|sum of" gpurun_out/r4k/ops.txt
