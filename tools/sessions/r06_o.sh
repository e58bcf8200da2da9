#!/bin/bash
# round 6, session o: the sums twin's x loads four rows ahead in every VEC kernel (nsu4) against only in the 32-deep ones (product): alternating processes
mkdir -p gpurun_out/r06_o
for rep in 1 2; do
for cfg in "dcgan64 1024 1" "cyclegan256 8 1" "dcgan32 256 8" "dcgan64 64 1" "dcgan32 256 1"; do
  set -- $cfg
  for lib in nsu4 product; do
    if [ $lib = product ]; then unset CGS_LIB; else export CGS_LIB=collaborative-gan-sampling_amd/libcgs_$lib.so; fi
    echo "--- $lib" >> gpurun_out/r06_o/step_ab.txt
    LB_REPS=5 python tools/step_ab.py $1 $2 $3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_o/step_ab.txt
  done
done
done
cat gpurun_out/r06_o/step_ab.txt
