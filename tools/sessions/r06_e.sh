#!/bin/bash
# round 6, session e: single calls of 128-512 images -- the split-K factor rule at these sizes (time model of round 5 against "aim at N blocks")
mkdir -p gpurun_out/r06_e
E=collaborative-gan-sampling_amd/libcgs_exp.so
AB="CGS_SPLITK_MODEL=1;CGS_SPLITK_MODEL=0,CGS_SPLITK_TARGET=512;CGS_SPLITK_MODEL=0,CGS_SPLITK_TARGET=768;CGS_SPLITK_MODEL=0,CGS_SPLITK_TARGET=1024;CGS_SPLITK_MODEL=0,CGS_SPLITK_TARGET=1536"
for cfg in "dcgan32 256" "dcgan32 128" "dcgan32 512" "dcgan64 256"; do
  set -- $cfg
  CGS_LIB=$E CGS_SPLITK_MODEL=0 CGS_SPLITK_TARGET=1536 LB_AB="$AB" LB_REPS=5 python tools/step_ab.py $1 $2 1 >> gpurun_out/r06_e/step_ab.txt 2>&1
done
CGS_LIB=$E CGS_PLAN_PRINT=1 LB_ITERS=1 LB_REPS=1 python tools/stage_bench.py dcgan32 256 1 2>&1 | grep "igemm plan" | sort | uniq -c > gpurun_out/r06_e/plans_dcgan32_b256.txt
cat gpurun_out/r06_e/step_ab.txt; cat gpurun_out/r06_e/plans_dcgan32_b256.txt
