# round 5, session ac: the split-K factor from the time model against the old "aim at 512 blocks" rule: parity on the product build, then one process per setting (the workspace is sized under the setting)
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_refine.py tests/test_gpu_fullsize.py tests/test_gpu_shaping.py tests/test_gpu_cyclegan.py tests/test_gpu_fuzz.py -q 2>&1 | tail -3 > gpurun_out/r05_ac_tests.log
export CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so
for cfg in "dcgan32 64 1" "dcgan64 64 1" "mnist 64 1" "dcgan32 64 4" "cyclegan256 8 1" "dcgan32 256 8" "mnist 64 32"; do
  for M in 0 1 0 1; do
    echo "== $cfg model=$M" >> gpurun_out/r05_ac_step.log
    CGS_SPLITK_MODEL=$M LB_ITERS=10 python tools/step_ab.py $cfg 2>&1 | grep -v amdgpu >> gpurun_out/r05_ac_step.log
  done
done
for A in dcgan32 dcgan64 mnist; do
  for M in 0 1; do
    echo "== $A 64 1 model=$M" >> gpurun_out/r05_ac_stage.log
    CGS_SPLITK_MODEL=$M python tools/stage_bench.py $A 64 1 2>&1 | grep "fwd\|bwd\|sum of" >> gpurun_out/r05_ac_stage.log
  done
done
