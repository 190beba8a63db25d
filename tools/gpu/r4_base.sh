set -x
mkdir -p gpurun_out/r4a
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
python tools/profile_ops.py > gpurun_out/r4a/ops.txt 2>&1
L=build/exp/libunet_exp5.so
python tools/h2_timeline.py $L 16 512 512 32 32 > gpurun_out/r4a/tl_32_32_fwd.txt 2>&1
python tools/h2_timeline.py $L 16 512 512 32 32 dgrad > gpurun_out/r4a/tl_32_32_dgrad.txt 2>&1
python tools/h2_timeline.py $L 16 512 512 64 32 > gpurun_out/r4a/tl_64_32_fwd.txt 2>&1
python tools/h2_timeline.py $L 16 512 512 64 32 dgrad > gpurun_out/r4a/tl_64_32_dgrad.txt 2>&1
python tools/h2_timeline.py $L 16 256 256 64 64 > gpurun_out/r4a/tl_256_64_64_fwd.txt 2>&1
