# round 5, session e: pixel-major launches under 1024 wide tiles -- 128x128 tiles in one round + the tail split against today's 128x64 tiles
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1
for A in mnist dcgan32; do
  CGS_FORCE_WIDE=1 CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_FORCE_WIDE=0;CGS_FORCE_WIDE=1" python tools/stage_bench.py $A > gpurun_out/r05_stage_${A}_wide.log 2>&1
done
python -m pytest tests/test_gpu_regressions.py -q -k tail 2>&1 | tail -3 > gpurun_out/r05_tail_tests.log
