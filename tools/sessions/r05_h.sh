# round 5, session h: 16-deep against 32-deep K tiles per stage (experiment build, CGS_FORCE_DEEP), small configurations
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1
for A in mnist dcgan32 cyclegan256; do
  CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_FORCE_DEEP=0;CGS_FORCE_DEEP=1" python tools/stage_bench.py $A > gpurun_out/r05_stage_${A}_deep.log 2>&1
done
