# round 5, session f: whole-step A/B of the tail split / wide-tile rule (one call in flight, hipGraph replay, alternating in one process)
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1
CGS_TAIL=1 CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_TAIL=0,CGS_FORCE_WIDE=0;CGS_TAIL=1,CGS_FORCE_WIDE=0;CGS_TAIL=1,CGS_FORCE_WIDE=1" python tools/step_ab.py mnist > gpurun_out/r05_step_ab.log 2>&1
