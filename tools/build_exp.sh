#!/bin/bash
# Experimental build of the library (-DCGS_EXPERIMENT: environment switches for A/B measurements) next to the product one:
#   bash tools/build_exp.sh  ->  collaborative-gan-sampling_amd/libcgs_exp.so   (use with CGS_LIB=<that path>)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/cgs_exp && mkdir -p /tmp/cgs_exp/pkg/csrc /tmp/cgs_exp/include
cp $R/collaborative-gan-sampling_amd/csrc/*.hip $R/collaborative-gan-sampling_amd/csrc/*.h $R/collaborative-gan-sampling_amd/csrc/Makefile /tmp/cgs_exp/pkg/csrc/
cp $R/include/*.h /tmp/cgs_exp/include/
make -C /tmp/cgs_exp/pkg/csrc -j8 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I/tmp/cgs_exp/include -I. -Wall -Wno-unused-function -DCGS_EXPERIMENT $EXTRA" > /tmp/cgs_exp/build.log 2>&1 || { tail -30 /tmp/cgs_exp/build.log; exit 1; }
cp /tmp/cgs_exp/pkg/libcgs_hip.so $R/collaborative-gan-sampling_amd/libcgs_exp.so
echo built $R/collaborative-gan-sampling_amd/libcgs_exp.so
