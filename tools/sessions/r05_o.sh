# round 5, session o: logit head + loss seed in one launch, the two row selects in one launch: parity, then step times (same box, before/after is by git stash -> two builds)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py tests/test_gpu_fullsize.py tests/test_gpu_class_fused.py tests/test_gpu_sync_bn.py tests/test_gpu_cyclegan.py -q 2>&1 | tail -4 > gpurun_out/r05_o_tests.log
python tools/step_ab.py mnist > gpurun_out/r05_o_step.log 2>&1
LB_ITERS=20 python tools/step_ab.py mnist 64 1 >> gpurun_out/r05_o_step.log 2>&1
python tools/step_ab.py dcgan64 >> gpurun_out/r05_o_step.log 2>&1
