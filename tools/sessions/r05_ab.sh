# round 5, session ab: the class-interleaved order on the split-bf16 kernels (dcgan64): per stage, per step, parity
cd $GRAFT_REPO_ROOT
export CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so
CGS_CLS_INTER=1 python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py -q -x -k "bx6" 2>&1 | tail -3 > gpurun_out/r05_ab_tests.log
CGS_CONTRACTION=bx6 LB_AB="CGS_CLS_INTER=0;CGS_CLS_INTER=1;CGS_CLS_INTER=1024" python tools/stage_bench.py dcgan64 2>&1 | grep -v amdgpu > gpurun_out/r05_ab_stage_dcgan64.log
CGS_CONTRACTION=bx6 LB_AB="CGS_CLS_INTER=0;CGS_CLS_INTER=1;CGS_CLS_INTER=1024" python tools/step_ab.py dcgan64 > gpurun_out/r05_ab_step.log 2>&1
