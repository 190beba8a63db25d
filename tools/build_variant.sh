#!/bin/bash
# A variant library for same-box A/B runs: tools/build_variant.sh <name> <file.hip> "<extra flags>"  ->  build/exp/libunet_<name>.so
# (the named source recompiled with the flags, every other object as built by make)
set -e
N=$1; F=$2; X=$3
C=$(ls -d one-stop-*_amd)/csrc
mkdir -p build/exp
make -s -C $C -j8
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function $X -c $C/$F -o build/exp/${F%.hip}_$N.o
OBJS=$(ls $C/*.o | grep -v "/${F%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/exp/${F%.hip}_$N.o -lpthread -o build/exp/libunet_$N.so
echo build/exp/libunet_$N.so
