# round 5, session c: tail split A/B per stage, interleaved in one process per configuration (experiment build: CGS_TAIL read per launch)
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1
for A in mnist dcgan32 dcgan64 cyclegan256; do
  CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_TAIL=0;CGS_TAIL=1" python tools/stage_bench.py $A > gpurun_out/r05_stage_${A}_ab.log 2>&1
done
