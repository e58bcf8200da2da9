#!/bin/bash
# round 6, session i: up to which tensor size do the norm-backward sums belong in the producing backward-data epilogue?  (CGS_NSTAT_MAX_MB: 0 = never)
mkdir -p gpurun_out/r06_i
AB="CGS_NSTAT_MAX_MB=0;CGS_NSTAT_MAX_MB=20;CGS_NSTAT_MAX_MB=40;CGS_NSTAT_MAX_MB=80;CGS_NSTAT_MAX_MB=100000"
for cfg in "cyclegan256 8 1" "mnist 64 32" "dcgan32 256 8" "dcgan64 1024 1" "dcgan64 64 1" "dcgan32 64 1" "dcgan32 256 1"; do
  set -- $cfg
  LB_AB="$AB" LB_REPS=5 python tools/step_ab.py $1 $2 $3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_i/step_ab.txt
done
cat gpurun_out/r06_i/step_ab.txt
