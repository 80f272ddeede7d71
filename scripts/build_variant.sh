#!/bin/bash
# Builds scripts/_bin/lib_<name>.so: the library with numeric.hip (or the file named in $3) recompiled with extra flags, for
# A/B runs on one GPU box (OKKT_LIB_PATH).  Usage: build_variant.sh <name> "<flags>" [file.hip]
set -e
cd "$(dirname "$0")/../onephase.jl_amd/csrc"
NAME=$1; FLAGS=$2; F=${3:-numeric.hip}
make -s -j8
B=${F%.hip}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 $FLAGS -c $F -o build/${B}_$NAME.o
OBJS=$(for o in amd nd mlnd symbolic api dist dataflow_sched numeric dataflow solve kkt linesearch; do [ "$o" != "$B" ] && echo build/$o.o; done)
mkdir -p ../../scripts/_bin
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scripts/_bin/lib_$NAME.so $OBJS build/${B}_$NAME.o -L/opt/rocm/lib -ldl -lpthread -Wl,-rpath,/opt/rocm/lib
echo built scripts/_bin/lib_$NAME.so
