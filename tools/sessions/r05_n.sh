# round 5, session n: the one-launch small-group norm (groups of <= 128 rows): parity, then the whole mnist step and the batch-64 class-surface call with and without it
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_cyclegan.py tests/test_gpu_shaping.py tests/test_gpu_refine.py tests/test_gpu_fullsize.py tests/test_gpu_class_fused.py tests/test_gpu_sync_bn.py tests/test_gpu_ops_param_grads.py -q 2>&1 | tail -4 > gpurun_out/r05_n_tests.log
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1 || cat gpurun_out/r05_build_exp.log
CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_NORM_SMALL=0;CGS_NORM_SMALL=1" python tools/step_ab.py mnist > gpurun_out/r05_step_ab_norm_small.log 2>&1
CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_NORM_SMALL=0;CGS_NORM_SMALL=1" LB_ITERS=20 python tools/step_ab.py mnist 64 1 >> gpurun_out/r05_step_ab_norm_small.log 2>&1
