# same-box A/B of library variants build/exp/libunet_<name>.so on the op-level timings of tools/gpu/pp_ab.py:  bash tools/gpu/pp_var_ab.sh name1 name2 ...
PK=one-stop-for-covid-19-infection-and-lung-segmentation-plus-classification_amd
cp $PK/libunet_hip.so /tmp/keep.so
for r in 1 2; do for m in "$@"; do cp build/exp/libunet_$m.so $PK/libunet_hip.so; echo "variant $m"; timeout 300 python tools/gpu/pp_ab.py --ops-only 2>&1 | tail -2; done; done
cp /tmp/keep.so $PK/libunet_hip.so
