O=gpurun_out/r5tl2; mkdir -p $O
L=build/exp/libunet_exp5.so
python tools/h2_timeline.py $L 16 512 512 64 32 dgrad > $O/tl_c9a_dgrad.txt 2>&1
python tools/h2_timeline.py $L 16 512 512 32 32 > $O/tl_c9b_fwd.txt 2>&1
python tools/h2_timeline.py $L 16 256 256 128 64 > $O/tl_c8a_fwd.txt 2>&1
python tools/h2_timeline.py $L 16 256 256 64 32 convT > $O/tl_u9_fwd.txt 2>&1
grep -h "epilogue\|lifetime" $O/*.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/predict_prof -- python $GRAFT_REPO_ROOT/tools/gpu/predict_trace.py 200 1 > $GRAFT_REPO_ROOT/$O/predict.txt 2>&1
cd $GRAFT_REPO_ROOT
find $O/predict_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/predict_kernel_stats.csv
rm -rf $O/predict_prof
