O=gpurun_out/r5ab1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "conv3x3" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
bash tools/ab_ops.sh build/exp/libunet_nop.so build/exp/libunet_p1.so 2 > $O/ab_ops.txt 2>&1; grep -E "conv3x3_(fwd|dgrad)" $O/ab_ops.txt | head -40; tail -3 $O/ab_ops.txt
