# round 5, session ai: the class flip at other batch sizes (128, 256 images: other numbers of K slices per round of 256 blocks), parity on the product build
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py tests/test_gpu_fullsize.py tests/test_gpu_cyclegan.py tests/test_gpu_fuzz.py -q 2>&1 | tail -2 > gpurun_out/r05_ai_tests.log
export CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so
for cfg in "dcgan64 128 1" "dcgan64 256 1" "dcgan32 128 1" "dcgan32 256 1" "dcgan64 64 2"; do
  echo "== $cfg" >> gpurun_out/r05_ai_stage.log
  LB_AB="CGS_CLS_FLIP=0;CGS_CLS_FLIP=1" CGS_PLAN_PRINT=0 python tools/stage_bench.py $cfg 2>&1 | grep "Deconv\|D Conv\|^  *bwd\|sum of" | grep "ig<" >> gpurun_out/r05_ai_stage.log
  LB_AB="CGS_CLS_FLIP=0;CGS_CLS_FLIP=1;CGS_CLS_FLIP=0;CGS_CLS_FLIP=1" LB_ITERS=10 python tools/step_ab.py $cfg 2>&1 | grep -v amdgpu >> gpurun_out/r05_ai_step.log
done
