#!/bin/bash
# round 6, session d: single calls of 128-512 images -- per-stage A/B of the launch-shape switches of an experiment build
mkdir -p gpurun_out/r06_d
E=collaborative-gan-sampling_amd/libcgs_exp.so
AB="CGS_SPLITK_MAXBLOCKS=256;CGS_SPLITK_MAXBLOCKS=512;CGS_SPLITK_MAXBLOCKS=1024;CGS_SPLITK_MAXBLOCKS=256,CGS_FORCE_DEEP=0;CGS_SPLITK_MAXBLOCKS=256,CGS_FORCE_DEEP=-,CGS_PIXMAX=0;CGS_SPLITK_MAXBLOCKS=256,CGS_PIXMAX=0,CGS_FORCE_DEEP=0"
for cfg in "dcgan32 256" "dcgan32 128" "dcgan32 512" "dcgan64 256"; do
  set -- $cfg
  CGS_LIB=$E CGS_SPLITK_MAXBLOCKS=1024 LB_AB="$AB,CGS_PIXMAX=-" LB_REPS=3 python tools/stage_bench.py $1 $2 1 > gpurun_out/r06_d/stage_$1_b$2.txt 2>&1
done
cat gpurun_out/r06_d/stage_dcgan32_b256.txt
