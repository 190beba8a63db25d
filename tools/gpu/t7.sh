mkdir -p gpurun_out/r4h
L=build/exp/libunet_exp5.so
python tools/h2_timeline.py $L 16 32 32 512 256 convT > gpurun_out/r4h/tl_u6_fwd.txt 2>&1
python tools/h2_timeline.py $L 16 64 64 256 128 convT > gpurun_out/r4h/tl_u7_fwd.txt 2>&1
python tools/h2_timeline.py $L 16 128 128 128 64 convT > gpurun_out/r4h/tl_u8_fwd.txt 2>&1
python tools/h2_timeline.py $L 16 32 32 512 256 convT_dgrad > gpurun_out/r4h/tl_u6_dgrad.txt 2>&1
python tools/h2_timeline.py $L 16 32 32 512 512 > gpurun_out/r4h/tl_c5b_fwd.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r4h/t_full.txt 2>&1; tail -3 gpurun_out/r4h/t_full.txt
cat gpurun_out/r4h/tl_*.txt | grep -v amdgpu
