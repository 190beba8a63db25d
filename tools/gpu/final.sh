mkdir -p gpurun_out/r4q
timeout 1800 python -m pytest tests/ -q -m gpu > gpurun_out/r4q/gpu_tests.txt 2>&1; tail -4 gpurun_out/r4q/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/collect_r04_profiles.sh gpurun_out/prof_r04 > gpurun_out/prof_r04_log.txt 2>&1; tail -3 gpurun_out/prof_r04_log.txt
timeout 900 bash tools/gpu/comm_ab.sh gpurun_out/r4q 3 2>&1 | tail -6
