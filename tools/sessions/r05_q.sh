# round 5, session q: split-K for launches that leave norm statistics (the reference's batch size on D's forward convs): parity, then the batch-64 calls again
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py tests/test_gpu_fullsize.py tests/test_gpu_class_fused.py tests/test_gpu_sync_bn.py tests/test_gpu_cyclegan.py tests/test_gpu_shaping.py -q 2>&1 | tail -4 > gpurun_out/r05_q_tests.log
for A in dcgan64 dcgan32 mnist; do
  python tools/stage_bench.py $A 64 1 > gpurun_out/r05_q_stage_${A}_b64.log 2>&1
  LB_ITERS=20 python tools/step_ab.py $A 64 1 > gpurun_out/r05_q_step_${A}_b64.log 2>&1
done
python tools/step_ab.py cyclegan256 > gpurun_out/r05_q_step_cyclegan256.log 2>&1
# split-K target (blocks the split aims at: 512 = two 32-deep blocks per CU) and K-tile depth of split launches, one process per setting (the workspace is sized under the setting)
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1 || cat gpurun_out/r05_build_exp.log
for A in dcgan64 dcgan32 mnist; do
  for M in "CGS_X=0" "CGS_SPLITK_TARGET=1024" "CGS_SPLITK_TARGET=1024 CGS_FORCE_DEEP=0" "CGS_SPLITK_TARGET=768" "CGS_SPLITK_TARGET=256"; do
    echo "== $A $M" >> gpurun_out/r05_q_splitk_target.log
    env $M CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_ITERS=20 python tools/step_ab.py $A 64 1 2>&1 | grep -v amdgpu >> gpurun_out/r05_q_splitk_target.log
    env $M CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so python tools/stage_bench.py $A 64 1 2>&1 | grep "fwd\|bwd\|sum of" >> gpurun_out/r05_q_splitk_target.log
  done
done
