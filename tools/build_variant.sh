#!/bin/bash
# Build libcgs_<tag>.so next to the product library with extra -D flags for ONE kernel source (A/B measurements on the GPU box:
# CGS_LIB=collaborative-gan-sampling_amd/libcgs_<tag>.so python tools/bx6_bench.py).   bash tools/build_variant.sh <tag> <source.hip> "<flags>"
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/collaborative-gan-sampling_amd/csrc
tag=$1; src=$2; flags=$3
make -C $C -j6 > /dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$C -Wall -Wno-unused-function $flags -c $C/$src -o /tmp/variant_$tag.o
objs=$(ls $C/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs /tmp/variant_$tag.o -o $R/collaborative-gan-sampling_amd/libcgs_$tag.so
echo built libcgs_$tag.so
