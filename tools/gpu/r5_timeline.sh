O=gpurun_out/r5tl; mkdir -p $O
L=build/exp/libunet_exp5.so
python tools/h2_timeline.py $L 16 512 512 32 32 > $O/tl_c9b_fwd.txt 2>&1
python tools/h2_timeline.py $L 16 512 512 64 32 > $O/tl_c9a_fwd.txt 2>&1
python tools/h2_timeline.py $L 16 512 512 64 32 dgrad > $O/tl_c9a_dgrad.txt 2>&1
python tools/h2_timeline.py $L 16 256 256 128 64 dgrad > $O/tl_c8a_dgrad.txt 2>&1
python tools/h2_timeline.py $L 16 256 256 128 64 > $O/tl_c8a_fwd.txt 2>&1
python tools/h2_timeline.py $L 16 64 64 512 256 > $O/tl_c6a_fwd.txt 2>&1
python tools/h2_timeline.py $L 16 256 256 64 32 convT > $O/tl_u9_fwd.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/predict_prof -- python $GRAFT_REPO_ROOT/tools/gpu/predict_trace.py 200 1 > $GRAFT_REPO_ROOT/$O/predict.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/gpu/predict_trace.py 200 1 > $O/predict_plain.txt 2>&1
find $O/predict_prof -name "*kernel_stats.csv" -exec cp {} $O/predict_kernel_stats.csv \;
rm -rf $O/predict_prof
tail -3 $O/tl_c9b_fwd.txt
