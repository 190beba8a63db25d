set -x
mkdir -p gpurun_out/r4f
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "head" > gpurun_out/r4f/t_ops.txt 2>&1; tail -3 gpurun_out/r4f/t_ops.txt
timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16_model.py -x -q -m gpu > gpurun_out/r4f/t_model.txt 2>&1; tail -5 gpurun_out/r4f/t_model.txt
for r in 1 2 3; do
for o in '{"side_prep":0}' '{"side_prep":1}'; do
python bench.py --no-cpu-baseline --no-strict-leg --no-fit-leg --steps 30 --warmup 5 --options "$o" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$o', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r4f/ab.txt
done; done
python tools/profile_ops.py > gpurun_out/r4f/ops.txt 2>&1; grep -E "c9b|head|sum of" gpurun_out/r4f/ops.txt
timeout 1500 python tools/fullsize_report.py --flips --json gpurun_out/r4f/fullsize_flips.json cls_224_bs256 unet_512_bs16 > gpurun_out/r4f/fullsize_flips.txt 2>&1; tail -40 gpurun_out/r4f/fullsize_flips.txt
