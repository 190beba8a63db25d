# same-box, same-process-options A/B of the training step with / without the persistent two-half conv schedule (context option CONV_PP): per-op profile + step time
for r in 1 2 3; do for v in 0 1; do
  echo "--- round $r conv_pp=$v"
  python tools/profile_ops.py --options "{\"conv_pp\": $v}" 2>&1 | grep -E "conv3x3_fwd:c1b|conv3x3_dgrad:c1b|conv3x3_dgrad:c9b|conv3x3_fwd_head:c9b|sum of op"
  python bench.py --steps 20 --warmup 8 --no-traffic-leg --options "{\"conv_pp\": $v}" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench ms_per_step', d['ms_per_step'], 'img/s', d['value'])"
done; done
