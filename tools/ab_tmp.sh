python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -4
export LB_ITERS=30
for V in 0 1 0 1; do echo "== CGS_DIRECT_EPI=$V"; env CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so CGS_DIRECT_EPI=$V python tools/layer_bench.py dcgan64 1024 2>&1 | grep -E "conv|sum"; done
