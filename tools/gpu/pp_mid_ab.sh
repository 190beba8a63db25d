PK=one-stop-for-covid-19-infection-and-lung-segmentation-plus-classification_amd
cp $PK/libunet_hip.so /tmp/keep.so
for r in 1 2; do for m in 2 4 6; do cp build/exp/libunet_mid$m.so $PK/libunet_hip.so; echo "mid=$m"; timeout 300 python tools/gpu/pp_ab.py --ops-only 2>&1 | tail -1; done; done
cp /tmp/keep.so $PK/libunet_hip.so
