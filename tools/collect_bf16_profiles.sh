#!/bin/bash
# bf16-storage evidence for profiles/: kernel trace + stats of the bf16 U-Net bench command, HBM traffic of its conv fwd+dgrad launches (two PMC
# passes, as tools/collect_profiles.sh), and the three bench lines.
# usage (GPU box, repo root): bash tools/collect_bf16_profiles.sh gpurun_out/prof_bf16
set -u
OUT=${1:-gpurun_out/prof_bf16}
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --dtype bf16 --steps 10 --warmup 15 --no-cpu-baseline > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python tools/profile_ops.py --dtype bf16 --reps 1 --warm 3 > $OUT/pmc_$c.log 2>&1
done
python tools/summarize_profiles.py $OUT > /dev/null 2>&1
python bench.py --dtype bf16 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_unet_512_bs16_bf16.json
python bench.py --dtype bf16 --arch unetpp --size 256 --batch 32 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_unetpp_256_bs32_bf16.json
python bench.py --dtype bf16 --arch classifier --size 224 --batch 256 --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_classifier_224_bs256_bf16.json
head -14 $OUT/kernel_stats.csv | cut -c1-70,150-200
