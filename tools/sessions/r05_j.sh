# round 5, session j: tall 256 x 64 tiles for launches to 64 channels (experiment build, CGS_TALL): parity with the tile forced everywhere it can run, then the per-stage A/B
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1 || cat gpurun_out/r05_build_exp.log
CGS_TALL=1 CGS_TALL_MIN=1 CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_fuzz_archs.py tests/test_gpu_regressions.py tests/test_gpu_refine.py tests/test_gpu_fullsize.py tests/test_gpu_cyclegan.py -q --deselect "tests/test_gpu_ops.py::test_many_block_grids_use_the_16_deep_variant" 2>&1 | tail -8 > gpurun_out/r05_tall_tests.log
for A in dcgan64 dcgan32 mnist cyclegan256; do
  CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_TALL=0;CGS_TALL=1" python tools/stage_bench.py $A > gpurun_out/r05_stage_${A}_tall.log 2>&1
done
