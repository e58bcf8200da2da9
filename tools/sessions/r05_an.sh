# round 5, session an: the moving-average update of ops.bn in four launches instead of nine: the new test, the generic-path tests, an in-process A/B of generic calls
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops_param_grads.py tests/test_gpu_refine.py tests/test_gpu_class_fused.py -q 2>&1 | tail -2 > gpurun_out/r05_an_tests.log
python tools/generic_bn_ab.py mnist > gpurun_out/r05_an_ab.log 2>&1
python tools/generic_bn_ab.py dcgan32 >> gpurun_out/r05_an_ab.log 2>&1
