# round 5, session ao: the class flip for unsplit two-round launches too (128 blocks per class): parity, then A/B (0 = off, 2 = split launches only, 1 = both)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py tests/test_gpu_fullsize.py tests/test_gpu_cyclegan.py tests/test_gpu_fuzz.py -q 2>&1 | tail -2 > gpurun_out/r05_ao_tests.log
export CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so
for cfg in "dcgan64 64 1" "dcgan32 64 1" "mnist 64 1" "cyclegan256 8 1" "dcgan64 1024 1"; do
  LB_AB="CGS_CLS_FLIP=2;CGS_CLS_FLIP=1;CGS_CLS_FLIP=2;CGS_CLS_FLIP=1" LB_ITERS=10 python tools/step_ab.py $cfg 2>&1 | grep -v amdgpu >> gpurun_out/r05_ao_step.log
done
LB_AB="CGS_CLS_FLIP=2;CGS_CLS_FLIP=1" python tools/stage_bench.py dcgan64 64 1 2>&1 | grep "fwd\|bwd\|sum of" > gpurun_out/r05_ao_stage.log
