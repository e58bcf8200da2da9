export LB_ITERS=30
for V in CGS_X=1 CGS_SKIP_EPI=1; do echo "== $V"; env CGS_LIB=$PWD/collaborative-gan-sampling_amd/libcgs_exp.so $V python tools/layer_bench.py dcgan64 1024 2>&1 | grep -E "conv|sum"; done
