# round 4, session b: first run of the split-bf16 implicit GEMM: operator parity, then per-layer timing in both modes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_b; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "conv2d_fwd or conv2d_bwd or deconv2d or bwd_data_epilogues or fused" ) > $O/pytest_ops.log 2>&1; tail -n 15 $O/pytest_ops.log
for m in f32 bx6; do echo "== $m"; CGS_CONTRACTION=$m timeout 300 python tools/layer_bench.py dcgan64 1024 2>&1 | tail -n 9; done | tee $O/layers.log
