set -x
mkdir -p gpurun_out/r4e
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "head" > gpurun_out/r4e/t_ops.txt 2>&1; tail -5 gpurun_out/r4e/t_ops.txt
for r in 1 2 3; do
for o in '{"head_fused":0}' '{"head_fused":1}'; do
python bench.py --no-cpu-baseline --no-strict-leg --no-fit-leg --steps 30 --warmup 5 --options "$o" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$o', d['value'], d['ms_per_step'])" | tee -a gpurun_out/r4e/ab.txt
done; done
python tools/profile_ops.py > gpurun_out/r4e/ops.txt 2>&1; grep -E "c9b|head|sum of" gpurun_out/r4e/ops.txt
