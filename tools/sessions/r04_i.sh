# round 4, session i: deep activation pipeline (default, V=3) vs V=1 vs loads-inside-the-stream (V=7); parity of the default first
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_i; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py -m gpu -q -x -k "bx6" ) > $O/pytest_ops.log 2>&1; tail -n 2 $O/pytest_ops.log
for v in hip v1 v7; do CGS_LIB=$R/collaborative-gan-sampling_amd/libcgs_$v.so python tools/bx6_bench.py 10 2>&1 | grep -v amdgpu.ids; done | tee $O/bench.log
CGS_LIB=$R/collaborative-gan-sampling_amd/libcgs_v7.so timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "bx6 and (conv2d_fwd or deconv2d_bwd)" 2>&1 | tail -n 2
