# round 5, session r: statistics launches split over K or not, config 5 (PatchGAN convs of 128-244 wide tiles) and the fused small configs, A/B in one process
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1 || cat gpurun_out/r05_build_exp.log
export CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so
LB_AB="CGS_STAT_SPLIT=0;CGS_STAT_SPLIT=1" python tools/stage_bench.py cyclegan256 > gpurun_out/r05_r_stage_cyclegan256.log 2>&1
LB_AB="CGS_STAT_SPLIT=0;CGS_STAT_SPLIT=1" python tools/step_ab.py cyclegan256 > gpurun_out/r05_r_step.log 2>&1
LB_AB="CGS_STAT_SPLIT=0;CGS_STAT_SPLIT=1" python tools/step_ab.py dcgan32 >> gpurun_out/r05_r_step.log 2>&1
LB_AB="CGS_STAT_SPLIT=0;CGS_STAT_SPLIT=1" python tools/step_ab.py mnist >> gpurun_out/r05_r_step.log 2>&1
LB_AB="CGS_STAT_SPLIT=0;CGS_STAT_SPLIT=1" LB_ITERS=20 python tools/step_ab.py dcgan32 64 1 >> gpurun_out/r05_r_step.log 2>&1
LB_AB="CGS_STAT_SPLIT=0;CGS_STAT_SPLIT=1" LB_ITERS=20 python tools/step_ab.py dcgan32 64 4 >> gpurun_out/r05_r_step.log 2>&1
LB_AB="CGS_STAT_SPLIT=0;CGS_STAT_SPLIT=1" LB_ITERS=20 python tools/step_ab.py dcgan64 64 2 >> gpurun_out/r05_r_step.log 2>&1
