#!/bin/bash
# Round-6 rocprofv3 evidence for profiles/ (run on the GPU box from the repo root; every pass is its own process, counters are collected WITHOUT tracing):
#   1. kernel trace + stats of the driver's bench command                        -> <out>/kernel_stats.csv
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE; counters only)                     -> <out>/pmc_traffic.json (HBM bytes per launch of the dominant kernel family)
#   3. PMC passes over the matrix kernels of the step (conv_h2 / wgrad_h2): MFMA busy, wave-cycle breakdown, VALU / LDS instruction counts
#                                                                                 -> <out>/pmc_matrix_summary.json
#   4. bench lines of the same commit: headline, driver flags, configs[2], strict fp32, the two other graphs, bf16 storage
#   5. per-op table (hipEvent per op, ops serialised)                             -> <out>/ops.txt
#   6. conv_pp A/B + phase timeline                                               -> <out>/ab_conv_pp_*.txt, pp_timeline.txt
# usage: bash tools/collect_r06_profiles.sh gpurun_out/prof_r06
set -u
OUT=${1:-gpurun_out/prof_r06}
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_flags.json 2>> $OUT/bench.err
python bench.py --config 2 --no-cpu-baseline --no-strict-leg > $OUT/bench_cfg2.json 2>> $OUT/bench.err
python bench.py --algo 2 --no-cpu-baseline --steps 20 > $OUT/bench_strict_fp32.json 2>> $OUT/bench.err
python bench.py --arch unetpp --size 256 --batch 32 --no-cpu-baseline --no-strict-leg > $OUT/bench_unetpp.json 2>> $OUT/bench.err
python bench.py --arch classifier --size 224 --batch 256 --no-cpu-baseline --no-strict-leg > $OUT/bench_cls.json 2>> $OUT/bench.err
python bench.py --arch classifier --size 224 --batch 256 --no-cpu-baseline --no-strict-leg --fold16 > $OUT/bench_cls_fold16.json 2>> $OUT/bench.err
python bench.py --no-cpu-baseline --no-strict-leg --deterministic > $OUT/bench_deterministic.json 2>> $OUT/bench.err
python bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2>> $OUT/bench.err
python tools/profile_ops.py > $OUT/ops.txt 2>&1
#   6. the persistent two-half conv schedule (csrc/kernels_conv_pp.hip): op-level and in-model same-box A/B against conv_h2_kernel (context option CONV_PP), and its phase
#      timeline from the measurement build (build/exp/libunet_trace.so = tools/build_variant.sh trace kernels_conv_pp.hip -DPP_TRACE=1, built before the gpurun call)
python tools/gpu/pp_ab.py --ops-only > $OUT/ab_conv_pp_ops.txt 2>&1
bash tools/gpu/pp_model_ab.sh > $OUT/ab_conv_pp_model.txt 2>&1
if [ -f build/exp/libunet_trace.so ]; then
  for m in fwd dgrad; do python tools/pp_timeline.py build/exp/libunet_trace.so 16 $m; done > $OUT/pp_timeline.txt 2>&1
fi
python tools/gpu/predict_trace.py 200 1 > $OUT/predict_batch1.txt 2>&1
R=$(pwd); (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/predict_trace -- python $R/tools/gpu/predict_trace.py 200 1 > $R/$OUT/predict_trace.log 2>&1)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-leg > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python tools/profile_ops.py --reps 1 --warm 3 > $OUT/pmc_$c.log 2>&1
done
python tools/summarize_profiles.py $OUT
bash tools/pmc_kernels.sh $OUT/pmc_matrix "conv_h2_kernel|conv_pp_kernel|wgrad_h2_kernel|wgradT_h2_kernel" \
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" > $OUT/pmc_matrix.txt 2>&1
python - $OUT <<'PY'
import json, sys
out = sys.argv[1]
d = json.load(open(out + "/pmc_matrix/pmc_summary.json"))
res = {}
for k, v in d.items():
    if "GRBM_GUI_ACTIVE" not in v: continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8                    # the counter is summed over the 8 XCDs
    wc = v["SQ_WAVE_CYCLES"]
    nm = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0          # SQ_VALU_MFMA_BUSY_CYCLES per v_mfma_f32_32x32x16_f16 = 32
    res[k] = {"gpu_cycles": round(cyc), "mfma_busy_frac_of_simd_cycles": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 3),
              "wave_parked_frac": round(v["SQ_WAIT_ANY"] / wc, 3), "wave_issue_stall_frac": round(v["SQ_WAIT_INST_ANY"] / wc, 3), "wave_issuing_frac": round(v["SQ_ACTIVE_INST_ANY"] / wc, 3),
              "resident_waves_per_simd": round(wc * 4 / (1024 * cyc), 2), "valu_per_mfma": round(v.get("SQ_INSTS_VALU", 0) / nm, 2), "lds_insts_per_mfma": round(v.get("SQ_INSTS_LDS", 0) / nm, 2),
              "salu_per_mfma": round(v.get("SQ_INSTS_SALU", 0) / nm, 2), "lds_busy_frac": round(v.get("SQ_LDS_IDX_ACTIVE", 0) / (256 * cyc), 3),
              "lds_conflict_frac": round(v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3), "launches_seen": v.get("launches_seen")}
json.dump({"source": "rocprofv3 --pmc (two counter-only passes) -- python tools/profile_ops.py --reps 1 --warm 2; tools/collect_r06_profiles.sh; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, "
           "GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md)", "kernels": res}, open(out + "/pmc_matrix_summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
