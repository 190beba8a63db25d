#!/bin/bash
# tools/publish_profiles.sh r05 gpurun_out/prof_r05: the summaries of a collect_rNN_profiles.sh run -> profiles/rNN_* (tracked); raw traces stay in gpurun_out/
R=$1; O=$2
cp $O/bench.json profiles/${R}_bench.json
cp $O/bench_driver_flags.json profiles/${R}_bench_driver_flags.json
for k in cfg2 strict_fp32 unetpp cls cls_fold16 bf16; do cp $O/bench_$k.json profiles/${R}_bench_$k.json; done
cp $O/bench_deterministic.json profiles/${R}_bench_unet512_bs16_deterministic.json
grep -v amdgpu.ids $O/ops.txt > profiles/${R}_ops_table.txt
cp $O/kernel_stats.csv profiles/${R}_rocprofv3_kernel_stats.csv
cp $O/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $O/pmc_matrix_summary.json profiles/${R}_pmc_matrix_summary.json
for c in FETCH_SIZE WRITE_SIZE; do f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f profiles/${R}_pmc_${c}_counter_collection.csv; done
grep -v amdgpu.ids $O/predict_batch1.txt > profiles/${R}_predict_batch1.txt
f=$(find $O/predict_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f profiles/${R}_predict_kernel_stats.csv
for f in ab_conv_pp_ops ab_conv_pp_model pp_timeline; do [ -f $O/$f.txt ] && grep -v amdgpu.ids $O/$f.txt > profiles/${R}_$f.txt; done
ls -la profiles/${R}_*
