# round 5, session g: the launch plan of every contraction stage of every configuration (tiles T against block slots L)
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1
for A in mnist dcgan32 dcgan64 cyclegan256; do
  CGS_PLAN_PRINT=1 LB_ITERS=1 LB_REPS=1 CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so python tools/stage_bench.py $A 2>&1 | grep "igemm plan" | sort | uniq -c | sort -rn > gpurun_out/r05_plan_$A.log
done
