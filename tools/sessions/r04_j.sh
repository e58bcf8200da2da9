# round 4, session j: the one-wave 128x64 block for 64-channel tiles (default) against the two-wave 256x64 block (solo0)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_j; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fuzz.py -m gpu -q -x -k "bx6" ) > $O/pytest_ops.log 2>&1; tail -n 2 $O/pytest_ops.log
for v in hip solo0; do CGS_LIB=$R/collaborative-gan-sampling_amd/libcgs_$v.so python tools/bx6_bench.py 10 2>&1 | grep -v amdgpu.ids | grep "64\b\|SUM\|N = 64\|->64\|64->128 bwd"; done | tee $O/bench.log
python bench.py --contraction bx6 --no-cpu-baseline --no-other-configs 2>>$O/stderr.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bx6', d['value'], d['ms_per_step'], d['roofline']['step_executed_frac'])" | tee -a $O/bench.log
