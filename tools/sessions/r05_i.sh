# round 5, session i: staggered start of the first round's blocks (experiment build, CGS_STAGGER = s_sleep(127) periods per slot)
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1 || cat gpurun_out/r05_build_exp.log
for A in mnist dcgan32 dcgan64; do
  CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_STAGGER=0;CGS_STAGGER=1;CGS_STAGGER=2;CGS_STAGGER=4" python tools/stage_bench.py $A > gpurun_out/r05_stage_${A}_stagger.log 2>&1
done
