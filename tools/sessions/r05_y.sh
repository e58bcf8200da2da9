# round 5, session y: split over K the launches of exactly 256 (up to 511) wide tiles too (dcgan32's 4x4 256->512 forward at 8 x 256, config 5's PatchGAN convs), per stage and per step
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1 || cat gpurun_out/r05_build_exp.log
export CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so
export CGS_SPLITK_MAXBLOCKS=512 CGS_SPLITK_TARGET=1024
AB="CGS_SPLITK_MAXBLOCKS=256,CGS_SPLITK_TARGET=512;CGS_SPLITK_MAXBLOCKS=257,CGS_SPLITK_TARGET=512;CGS_SPLITK_MAXBLOCKS=257,CGS_SPLITK_TARGET=1024;CGS_SPLITK_MAXBLOCKS=512,CGS_SPLITK_TARGET=1024"
for A in dcgan32 cyclegan256; do
  LB_AB="$AB" python tools/stage_bench.py $A > gpurun_out/r05_y_stage_$A.log 2>&1
  LB_AB="$AB" python tools/step_ab.py $A > gpurun_out/r05_y_step_$A.log 2>&1
done
