#!/bin/bash
# round 6, session za: which norms take the one-launch finalize + apply -- limits on partial rows and tensor bytes (experiment build)
mkdir -p gpurun_out/r06_za
E=collaborative-gan-sampling_amd/libcgs_exp.so
AB="CGS_NORM_FA=0;CGS_NORM_FA=1,CGS_NORM_FA_ROWS=64,CGS_NORM_FA_MB=1;CGS_NORM_FA=1,CGS_NORM_FA_ROWS=64,CGS_NORM_FA_MB=2;CGS_NORM_FA=1,CGS_NORM_FA_ROWS=64,CGS_NORM_FA_MB=4;CGS_NORM_FA=1,CGS_NORM_FA_ROWS=16,CGS_NORM_FA_MB=2"
for cfg in "dcgan32 64 1" "dcgan64 64 1" "mnist 64 1" "dcgan32 256 1"; do
  set -- $cfg
  CGS_LIB=$E LB_AB="$AB" LB_REPS=7 python tools/step_ab.py $1 $2 $3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_za/step_ab.txt
done
cat gpurun_out/r06_za/step_ab.txt
