# round 5, session t: the per-sample scalars of the best-row select in the copy launch (tickets): parity + step time; 64-row tiles for launches of <= 64 GEMM rows (experiment build)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py tests/test_gpu_fullsize.py tests/test_gpu_class_fused.py tests/test_gpu_sync_bn.py tests/test_gpu_cyclegan.py -q 2>&1 | tail -4 > gpurun_out/r05_t_tests.log
LB_ITERS=20 python tools/step_ab.py mnist 64 1 > gpurun_out/r05_t_step.log 2>&1
LB_ITERS=20 python tools/step_ab.py dcgan32 64 1 >> gpurun_out/r05_t_step.log 2>&1
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1 || cat gpurun_out/r05_build_exp.log
export CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so
CGS_BM64=1 python -m pytest tests/test_gpu_ops.py -q -k "linear" 2>&1 | tail -3 > gpurun_out/r05_t_bm64_tests.log
CGS_BM64=1 python -m pytest tests/test_gpu_refine.py -q -k "mnist" 2>&1 | tail -3 >> gpurun_out/r05_t_bm64_tests.log
LB_AB="CGS_BM64=0;CGS_BM64=1" python tools/stage_bench.py mnist 64 1 > gpurun_out/r05_t_bm64_stage.log 2>&1
LB_AB="CGS_BM64=0;CGS_BM64=1" LB_ITERS=20 python tools/step_ab.py mnist 64 1 > gpurun_out/r05_t_bm64_step.log 2>&1
