# round 5, session aj: the class flip restricted to image-major two-round launches: A/B over the batch sizes again, then the whole GPU suite on the product build
cd $GRAFT_REPO_ROOT
export CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so
for cfg in "dcgan64 64 1" "dcgan32 64 1" "dcgan64 128 1" "dcgan32 256 1" "dcgan32 128 1" "mnist 64 1"; do
  LB_AB="CGS_CLS_FLIP=0;CGS_CLS_FLIP=1" LB_ITERS=10 python tools/step_ab.py $cfg 2>&1 | grep -v amdgpu >> gpurun_out/r05_aj_step.log
done
unset CGS_LIB
python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/r05_aj_fullsuite.log
