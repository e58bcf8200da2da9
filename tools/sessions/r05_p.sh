# round 5, session p: the reference's batch size (64) on the DCGAN nets, one batch per call (the shaping loop's form): launch plans, per-stage times, kernel table of the whole call
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1 || cat gpurun_out/r05_build_exp.log
for A in dcgan64 dcgan32; do
  CGS_PLAN_PRINT=1 LB_ITERS=1 LB_REPS=1 CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so python tools/stage_bench.py $A 64 1 2>&1 | grep "igemm plan" | sort | uniq -c | sort -rn > gpurun_out/r05_p_plan_${A}_b64.log
  python tools/stage_bench.py $A 64 1 > gpurun_out/r05_p_stage_${A}_b64.log 2>&1
  LB_ITERS=20 python tools/step_ab.py $A 64 1 > gpurun_out/r05_p_step_${A}_b64.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
LB_ITERS=5 LB_REPS=2 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r05_p_prof -o b64 -- python3 $GRAFT_REPO_ROOT/tools/step_ab.py dcgan64 64 1 > $GRAFT_REPO_ROOT/gpurun_out/r05_p_prof.log 2>&1
