L=$(ls -d one-stop-*_amd)/libunet_hip.so
cp $L /tmp/keep.so
for v in "$@"; do
  cp build/exp/libunet_$v.so $L
  python bench.py --arch unetpp --size 256 --batch 32 --no-cpu-baseline --no-strict-leg --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], {k:v for k,v in list(d['roofline']['op_ms_per_step'].items())[:4]})"
done
cp /tmp/keep.so $L
