# round 5, session l: the tail split on launches that leave norm statistics (the reduce pass writes their partial rows): per-stage and whole-step A/B on mnist
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1 || cat gpurun_out/r05_build_exp.log
CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_TAIL=0,CGS_FORCE_WIDE=0,CGS_TALL=0;CGS_TAIL=1,CGS_FORCE_WIDE=-,CGS_TALL=-" python tools/stage_bench.py mnist > gpurun_out/r05_stage_mnist_stats_tail.log 2>&1
CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_TAIL=0,CGS_FORCE_WIDE=0,CGS_TALL=0;CGS_TAIL=1,CGS_FORCE_WIDE=-,CGS_TALL=-" python tools/step_ab.py mnist > gpurun_out/r05_step_mnist_now.log 2>&1
