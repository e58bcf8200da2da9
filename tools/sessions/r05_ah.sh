# round 5, session ah: split launches of several parity classes: every other round of 256 blocks takes the classes in reverse order (a CU gets a long and a short K slice)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py tests/test_gpu_fullsize.py tests/test_gpu_cyclegan.py tests/test_gpu_fuzz.py -q 2>&1 | tail -2 > gpurun_out/r05_ah_tests.log
export CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so
for cfg in "dcgan64 64 1" "dcgan32 64 1" "mnist 64 1" "dcgan64 64 2" "cyclegan256 8 1"; do
  LB_AB="CGS_CLS_FLIP=0;CGS_CLS_FLIP=1" LB_ITERS=10 python tools/step_ab.py $cfg 2>&1 | grep -v amdgpu >> gpurun_out/r05_ah_step.log
done
for A in dcgan64 dcgan32; do
  LB_AB="CGS_CLS_FLIP=0;CGS_CLS_FLIP=1" python tools/stage_bench.py $A 64 1 2>&1 | grep "fwd\|bwd\|sum of" >> gpurun_out/r05_ah_stage.log
done
