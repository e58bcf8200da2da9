# round 5, session d: tail split again with the reduce pass spread over row chunks (parity under the experiment build, then the per-stage A/B)
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1
CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so CGS_TAIL=1 python -m pytest tests/test_gpu_regressions.py -q -k tail 2>&1 | tail -3 > gpurun_out/r05_tail_tests.log
for A in mnist dcgan32 dcgan64 cyclegan256; do
  CGS_TAIL=1 CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_TAIL=0;CGS_TAIL=1" python tools/stage_bench.py $A > gpurun_out/r05_stage_${A}_ab2.log 2>&1
done
