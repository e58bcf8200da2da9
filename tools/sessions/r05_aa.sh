# round 5, session aa: the parity classes of a tile back to back on one XCD (image-major transposed launches), per stage and per step, f32
cd $GRAFT_REPO_ROOT
export CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so
CGS_CLS_INTER=1 python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py tests/test_gpu_cyclegan.py -q -x 2>&1 | tail -3 > gpurun_out/r05_aa_tests.log
for A in dcgan64 cyclegan256 dcgan32 mnist; do
  LB_AB="CGS_CLS_INTER=0;CGS_CLS_INTER=1;CGS_CLS_INTER=1024" python tools/stage_bench.py $A 2>&1 | grep -v amdgpu | grep "Deconv\|^  *bwd\|modes\|sum of" > gpurun_out/r05_aa_stage_$A.log
done
LB_AB="CGS_CLS_INTER=0;CGS_CLS_INTER=1;CGS_CLS_INTER=1024" python tools/step_ab.py dcgan64 > gpurun_out/r05_aa_step.log 2>&1
LB_AB="CGS_CLS_INTER=0;CGS_CLS_INTER=1;CGS_CLS_INTER=1024" python tools/step_ab.py cyclegan256 >> gpurun_out/r05_aa_step.log 2>&1
