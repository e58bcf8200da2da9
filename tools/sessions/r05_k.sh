# round 5, session k: tall 256 x 64 tiles -- parity with the tile forced onto every launch it can serve (experiment build), the product
# rule's whole-step effect on mnist (same-process A/B), and the product build's own parity on the configurations it changes
cd $GRAFT_REPO_ROOT
bash tools/build_exp.sh > gpurun_out/r05_build_exp.log 2>&1 || cat gpurun_out/r05_build_exp.log
CGS_TALL=1 CGS_TALL_MIN=1 CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py tests/test_gpu_fuzz_archs.py tests/test_gpu_regressions.py tests/test_gpu_refine.py tests/test_gpu_fullsize.py tests/test_gpu_cyclegan.py -q --deselect "tests/test_gpu_ops.py::test_many_block_grids_use_the_16_deep_variant" 2>&1 | tail -8 > gpurun_out/r05_tall_tests.log
CGS_TAIL=1 CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so LB_AB="CGS_TALL=0;CGS_TALL=1,CGS_TALL_MIN=1536" python tools/step_ab.py mnist > gpurun_out/r05_step_ab_tall.log 2>&1
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_class_fused.py tests/test_gpu_ops.py -q 2>&1 | tail -4 > gpurun_out/r05_tall_product_tests.log
