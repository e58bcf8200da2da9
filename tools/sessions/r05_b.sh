# round 5, session b: the tail split -- parity of the launches that take it, per-stage timing of every configuration with and without it
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_regressions.py -q -x -k tail 2>&1 | tail -15 > gpurun_out/r05_tail_tests.log
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1
for A in mnist dcgan32 cyclegan256 dcgan64; do
  CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so CGS_TAIL=0 python tools/stage_bench.py $A > gpurun_out/r05_stage_${A}_notail.log 2>&1
  CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so python tools/stage_bench.py $A > gpurun_out/r05_stage_${A}_tail.log 2>&1
done
